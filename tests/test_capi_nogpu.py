"""CPU-only: the C-ABI library loads, exports every declared symbol, validates
arguments, and fails loudly (no CPU fallback) when there is no HIP device."""
import ctypes
import os
import re

import numpy as np
import pytest

from rcppsparse_amd import capi, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "rcppsparse_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rsp_[a-z_0-9]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert _declared_symbols() == sorted(capi.EXPORTED_SYMBOLS)


def test_library_exports_every_declared_symbol():
    L = capi.load()
    for sym in _declared_symbols():
        assert hasattr(L, sym), sym
    assert "gfx950" in capi.version()


def _no_gpu():
    return capi.device_count() == 0


def test_device_count_is_an_int():
    assert capi.device_count() >= 0


def test_no_cpu_fallback_without_device():
    if not _no_gpu():
        pytest.skip("a GPU is present")
    m = synth.rsparsematrix(10, 10, density=0.1, seed=1)
    with pytest.raises(capi.RspError) as e:
        capi.column_sums_host(m["x"], m["p"])
    assert e.value.code == capi.RSP_ERR_NO_DEVICE
    with pytest.raises(capi.RspError):
        capi.DeviceCSC(m["x"], m["p"], m["Dim"])


def test_bad_offsets_rejected_before_touching_a_device():
    x = np.ones(4)
    for p in ([1, 2, 3, 4], [0, 3, 2, 4], [0, 1, 2, 3]):
        with pytest.raises(capi.RspError) as e:
            capi.column_sums_host(x, np.array(p, dtype=np.int32))
        assert e.value.code == capi.RSP_ERR_BAD_ARG


def test_partition_columns_balanced_and_exact():
    rng = np.random.default_rng(0)
    for ncol, nparts in [(1, 1), (5, 8), (1000, 2), (1000, 4), (1000, 8), (12345, 7)]:
        counts = rng.integers(0, 50, size=ncol)
        p = synth.offsets_from_counts(counts)
        b = capi.partition_columns(p, nparts)
        assert b[0] == 0 and b[-1] == ncol and np.all(np.diff(b) >= 0)
        nnz = int(p[-1])
        for k in range(1, nparts):
            target = k * nnz // nparts
            want = int(np.searchsorted(p, target, side="left"))
            assert b[k] == max(want, b[k - 1])
        # shards reassemble: rebased offsets describe the same columns
        for k in range(nparts):
            c0, c1 = int(b[k]), int(b[k + 1])
            pl = capi.rebase_offsets(p, c0, c1)
            assert pl[0] == 0 and np.array_equal(np.diff(pl), np.diff(p[c0:c1 + 1]))


def test_partition_zipf_never_splits_a_column():
    counts = synth.zipf_counts(2000, 200000, seed=3, nrow=30000)
    p = synth.offsets_from_counts(counts)
    b = capi.partition_columns(p, 8)
    shard_nnz = np.diff(p[b])
    assert shard_nnz.sum() == p[-1]
    assert shard_nnz.max() <= p[-1] / 8 + counts.max()


def test_partition_properties_hypothesis():
    """Any column-length profile, any number of parts: contiguous cover, no column split, every
    part within one column of the ideal share, and summing the shards with the oracle gives the
    whole-matrix result bit for bit (columns are independent)."""
    from hypothesis import given, settings, strategies as st
    import oracle

    lengths = st.lists(st.one_of(st.integers(0, 6), st.integers(0, 400), st.just(0), st.integers(2000, 9000)),
                       min_size=1, max_size=300)

    @settings(max_examples=120, deadline=None)
    @given(lengths, st.integers(1, 16), st.integers(0, 2**31 - 1))
    def check(counts, nparts, seed):
        counts = np.asarray(counts, dtype=np.int64)
        p = synth.offsets_from_counts(counts)
        b = capi.partition_columns(p, nparts)
        assert b[0] == 0 and b[-1] == counts.size and np.all(np.diff(b) >= 0)
        shard_nnz = np.diff(p[b].astype(np.int64))
        assert shard_nnz.sum() == p[-1]
        assert shard_nnz.max() <= -(-int(p[-1]) // nparts) + int(counts.max())
        x = synth.gen_values(int(p[-1]), seed=seed % 1000, kind=0)
        whole = oracle.column_sums(x, p)
        parts = []
        for k in range(nparts):
            c0, c1 = int(b[k]), int(b[k + 1])
            pl = capi.rebase_offsets(p, c0, c1)
            parts.append(oracle.column_sums(x[p[c0]:p[c1]], pl))
        assert np.concatenate(parts).tobytes() == whole.tobytes()

    check()


def test_argument_errors_do_not_need_a_device():
    L = capi.load()
    import ctypes
    # negative sizes / null pointers are rejected with BAD_ARG and a message, not a crash
    out = np.zeros(4)
    p = np.zeros(5, dtype=np.int32)
    rc = L.rsp_column_sums_host(None, p.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), 4, -1,
                                out.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), 0)
    assert rc == capi.RSP_ERR_BAD_ARG and b"nnz" in L.rsp_last_error()
    rc = L.rsp_column_sums_host(None, p.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), -3, 0,
                                out.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), 0)
    assert rc == capi.RSP_ERR_BAD_ARG
    assert L.rsp_debug_set(b"chunk_rows", -1) == capi.RSP_ERR_BAD_ARG and L.rsp_debug_set(b"experiment", -1) == capi.RSP_ERR_BAD_ARG
    assert L.rsp_debug_set(b"no_such_knob", 1) == capi.RSP_ERR_BAD_ARG and b"no_such_knob" in L.rsp_last_error()
    assert L.rsp_debug_set(None, 1) == capi.RSP_ERR_BAD_ARG
    assert L.rsp_partition_columns(None, 3, 2, None) == capi.RSP_ERR_BAD_ARG
    assert L.rsp_csc_free(None) == capi.RSP_OK          # freeing nothing is fine
    assert L.rsp_comm_destroy(None) == capi.RSP_OK
    # 2^31 nonzeros cannot be addressed by the 32-bit p[] (RcppSparse.h:30)
    rc = L.rsp_column_sums_device(None, None, 1, 2**31, None, None, 0, None)
    assert rc == capi.RSP_ERR_BAD_ARG


def test_workspace_size_is_a_pure_function_of_nnz():
    a = capi.workspace_bytes(10, 10**9)
    assert a == capi.workspace_bytes(10**6, 10**9) and a >= 24 * (10**9 // (256 * 128))
    assert capi.workspace_bytes(5, 0) > 0


def _chunk_starts(plan, nnz):
    w = np.arange(plan["nchunks"], dtype=np.int64)
    edge = plan["nbody"] * plan["body_elems"]
    return np.where(w < plan["nbody"], w * plan["body_elems"], edge + (w - plan["nbody"]) * plan["tail_elems"])


@pytest.mark.parametrize("nnz", [1, 127, 128, 129, 2048, 10**5, 10**7, 125_000_000, 10**9, 2**31 - 1])
def test_chunk_plan_tiles_x_exactly(nnz):
    """rsp_plan_describe: whatever the knobs say, the chunks are whole 128-element rows, start where the
    previous one ends, cover [0, nnz) with no empty chunk, never exceed 1 GiB of x, and the workspace is
    32 bytes per chunk."""
    try:
        for tuning, taper in [(0, (-1, -1)), (0, (0, 0)), (0, (300, 16)), (0, (1000, 8)), (1, (-1, -1)),
                              (37, (-1, -1)), (2**22, (-1, -1)), (2**30, (500, 1))]:
            capi.set_tuning(tuning)
            capi.set_taper(*taper)
            plan = capi.plan_describe(nnz)
            assert plan["body_elems"] % 128 == 0 and plan["tail_elems"] % 128 == 0
            assert 128 <= plan["tail_elems"] <= plan["body_elems"] <= 2**27
            assert 0 <= plan["nbody"] <= plan["nchunks"] and plan["nchunks"] >= 1
            starts = _chunk_starts(plan, nnz)
            assert starts[0] == 0 and np.all(np.diff(starts) > 0) and starts[-1] < nnz
            last_len = plan["body_elems"] if plan["nchunks"] == plan["nbody"] else plan["tail_elems"]
            assert starts[-1] + last_len >= nnz                       # the last chunk reaches the end ...
            if plan["nchunks"] > plan["nbody"] > 0:                   # ... and the tail begins where the body ends
                assert starts[plan["nbody"]] == plan["nbody"] * plan["body_elems"]
            assert capi.workspace_bytes(10, nnz) == -(-32 * plan["nchunks"] // 256) * 256
    finally:
        capi.set_tuning(0)
        capi.set_taper(-1, -1)


def test_automatic_chunk_plan_policy():
    """The automatic policy: short calls (one round of wavefronts) get 20-row chunks and no taper; calls of
    more than two rounds get 256-row chunks with the last tenth of x in 64-row chunks; an explicit
    rsp_debug_set("chunk_rows", n) switches the taper off."""
    c2 = capi.plan_describe(10**7)
    assert c2["body_elems"] == 20 * 128 and c2["nchunks"] == c2["nbody"] == -(-78125 // 20)
    shard = capi.plan_describe(125_000_000)
    assert shard["nchunks"] == shard["nbody"] and 6144 < shard["nchunks"] <= 12288
    c3 = capi.plan_describe(10**9)
    assert c3["body_elems"] == 256 * 128 and c3["tail_elems"] == 64 * 128
    tail_rows = (c3["nchunks"] - c3["nbody"]) * 64
    total_rows = -(-10**9 // 128)
    assert 0.09 * total_rows < tail_rows < 0.11 * total_rows
    try:
        capi.set_tuning(256)
        fixed = capi.plan_describe(10**9)
        assert fixed["nchunks"] == fixed["nbody"] == -(-total_rows // 256)
    finally:
        capi.set_tuning(0)


def test_plan_entries_validate_arguments_and_do_not_answer_without_a_device():
    """rsp_column_sums_plan_*: a bad p[] is rejected on the host before any device is touched; without a device
    the inspector says so (it uploads its records) instead of handing out a plan nothing could execute."""
    good = np.array([0, 2, 2, 5], dtype=np.int32)
    for p in ([1, 2, 3, 4], [0, 3, 2, 4], [0, 1, 2, 3]):
        with pytest.raises(capi.RspError) as e:
            capi.ColumnSumsPlan(np.array(p, dtype=np.int32), nnz=4)
        assert e.value.code == capi.RSP_ERR_BAD_ARG
    L = capi.load()
    assert L.rsp_column_sums_plan_destroy(None) == capi.RSP_OK           # destroying nothing is fine
    assert L.rsp_column_sums_plan_info(None, None, None) == capi.RSP_ERR_BAD_ARG
    assert L.rsp_column_sums_planned_device(None, None, None, 0, 0, 0, None, None, 0, None) == capi.RSP_ERR_BAD_ARG
    if _no_gpu():
        with pytest.raises(capi.RspError) as e:
            capi.ColumnSumsPlan(good)
        assert e.value.code == capi.RSP_ERR_NO_DEVICE
    capi.set_lean(False)
    capi.set_lean(True)


def test_row_reduce_workspace_arithmetic_and_argument_checks():
    """rsp_comm_reduce_rows_workspace_bytes: nranks incoming pieces + the reduced slice, slices of whole 16-byte
    pairs; the entries reject null communicators / buffers instead of dereferencing them."""
    for nranks, nrow in [(1, 1), (1, 7), (2, 7), (3, 10), (8, 10_000_000), (8, 10_000_001), (5, 0)]:
        per = -(-nrow // nranks)
        per += per & 1
        want = ((nranks + 1) * per * 8 + 255) // 256 * 256
        assert capi.reduce_rows_workspace_bytes(nranks, nrow) == want, (nranks, nrow)
    assert capi.reduce_rows_workspace_bytes(0, 5) == 0 and capi.reduce_rows_workspace_bytes(2, -1) == 0
    L = capi.load()
    assert L.rsp_comm_reduce_rows(None, None, 5, 0, None, None, 0, 0, None) == capi.RSP_ERR_BAD_ARG
    assert L.rsp_add_partials_device(None, 0, 4, 4, 0, None, None) == capi.RSP_ERR_BAD_ARG      # no parts
    assert L.rsp_add_partials_device(None, 2, 3, 4, 0, None, None) == capi.RSP_ERR_BAD_ARG      # stride < n
    assert L.rsp_add_partials_device(None, 2, 4, 4, 0, None, None) == capi.RSP_ERR_BAD_ARG      # null buffers
    assert L.rsp_add_partials_device(None, 2, 0, 0, 0, None, None) == capi.RSP_OK               # nothing to add
    assert L.rsp_mcsc_row_sums(None, None) == capi.RSP_ERR_BAD_ARG
    assert L.rsp_mcsc_column_means(None, None) == capi.RSP_ERR_BAD_ARG


def test_row_restricted_form_selection_is_host_logic():
    """rsp_column_sums_in_rows_form: which form a call takes follows from its sizes and workspace alone (the slice-major
    form's device-side guard aside) -- no device needed.  Bitmap <= 16 KB: L1; <= 128 KB: LDS; above: slices when the
    columns are long enough, there are enough of them, a column group outweighs its bitmap copies and the workspace has
    room for the guard's flag; L2 otherwise."""
    capi.load()
    big = dict(nrow=10_000_000, ncol=1_000_000, nnz=1_000_000_000)
    assert capi.in_rows_form(131_072, 10, 100) == "L1" and capi.in_rows_form(131_073, 10, 100) == "LDS"
    assert capi.in_rows_form(1_048_576, 10, 100) == "LDS" and capi.in_rows_form(1_048_577, 10, 100) == "L2"
    assert capi.in_rows_form(**big) == "slices"
    assert capi.in_rows_form(**big, workspace_bytes=capi.workspace_bytes(big["ncol"], big["nnz"])) == "L2"
    assert capi.in_rows_form(**big, workspace_bytes=capi.in_rows_workspace_bytes(**big) - 1) == "L2"
    assert capi.in_rows_form(10_000_000, 4_000_000, 1_000_000_000) == "L2"      # 25 entries per column and slice
    assert capi.in_rows_form(30_000_000, 1_000_000, 1_000_000_000) == "slices"   # 34
    assert capi.in_rows_form(10_000_000, 13_311, 1_000_000_000) == "L2"         # one column too few (round 4: 13312, was 16384)
    assert capi.in_rows_form(10_000_000, 13_312, 1_000_000_000) == "slices"
    assert capi.in_rows_form(10_000_000, 16_384, 1_000_000_000) == "slices"
    assert capi.in_rows_form(2**31 - 1, 16_384, 2**31 - 1) == "L2"              # 2048 bitmap copies per 64 columns
    capi.set_row_slices(0)
    try:
        assert capi.in_rows_form(**big) == "L2"
        capi.set_row_slices(2)
        assert capi.in_rows_form(1_048_577, 3, 100) == "slices" and capi.in_rows_form(1_048_576, 3, 100) == "LDS"
    finally:
        capi.set_row_slices(1)
    with pytest.raises(ValueError):
        capi.in_rows_form(-1, 1, 1)


def test_host_barrier_between_processes(tmp_path):
    """rsp_host_barrier_* (the fence of the direct-write gather): three processes cross the same barrier 3000 times;
    between two crossings each adds its rank's increment to a shared file-backed counter array, and after every
    crossing every process sees all the increments of the round before.  A barrier one rank never reaches times out
    with an error instead of hanging."""
    import subprocess
    import sys
    name = f"/rsp_nogpu_{os.getpid()}"
    counters = tmp_path / "counters.bin"
    counters.write_bytes(bytes(3 * 8))
    code = f"""
import sys, mmap, struct
sys.path.insert(0, {ROOT!r})
from rcppsparse_amd import capi
capi.load(build=False)
rank = int(sys.argv[1])
b = capi.HostBarrier({name!r}, 3, rank)
f = open({str(counters)!r}, "r+b"); m = mmap.mmap(f.fileno(), 24)
for k in range(3000):
    struct.pack_into("q", m, 8 * rank, k + 1)
    b.wait(30.0)
    seen = struct.unpack_from("3q", m, 0)
    assert min(seen) >= k + 1, (rank, k, seen)
    b.wait(30.0)
b.close()
print("ok", rank)
"""
    procs = [subprocess.Popen([sys.executable, "-c", code, str(r)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for r in range(3)]
    for r, pr in enumerate(procs):
        out, err = pr.communicate(timeout=180)
        assert pr.returncode == 0 and f"ok {r}" in out, err[-1500:]
    # a missing rank: the wait comes back with an error after its timeout
    b = capi.HostBarrier(name + "_lonely", 2, 0)
    with pytest.raises(capi.RspError) as e:
        b.wait(0.3)
    assert "timed out" in str(e.value)
    b.close()


def test_round6_entries_check_their_arguments_without_a_device():
    """The entries round 6 added -- the single-process multi-GPU handle's knobs, rsp_mcsc_wrap_device, the plan-free entry's
    settle, the shared host vector, the RCCL / peer queries -- refuse bad arguments (or answer) before any device is
    touched, and the read-only counters of the plan cache are there."""
    L = capi.load()
    null = ctypes.c_void_p()
    info = np.zeros(4, dtype=np.int32)
    assert L.rsp_mcsc_set_gather(null, 0) == capi.RSP_ERR_BAD_ARG
    assert L.rsp_mcsc_set_launch(null, 0) == capi.RSP_ERR_BAD_ARG
    assert L.rsp_mcsc_config(null, info.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))) == capi.RSP_ERR_BAD_ARG
    assert not L.rsp_mcsc_result_buffer(null)                                # NULL for a null handle
    us = np.zeros(8)
    assert L.rsp_mcsc_last_call_stamps(null, us.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), 8) == capi.RSP_ERR_BAD_ARG
    ms = ctypes.c_float(0)
    assert L.rsp_mcsc_shard_kernel_ms(null, 0, 3, ctypes.byref(ms)) == capi.RSP_ERR_BAD_ARG
    h = ctypes.c_void_p()
    assert L.rsp_mcsc_wrap_device(0, None, None, None, None, None, None, 5, ctypes.byref(h)) == capi.RSP_ERR_BAD_ARG
    assert not h.value
    assert L.rsp_column_sums_device_settle(None, 10, 100, None) == -1       # no offsets: nothing to settle
    assert L.rsp_column_sums_device_settle(ctypes.c_void_p(4096), 0, 0, None) == -1
    hp = ctypes.c_void_p()
    assert L.rsp_shared_host_open(b"no-leading-slash", 64, 1, ctypes.byref(hp)) == capi.RSP_ERR_BAD_ARG
    assert L.rsp_shared_host_open(b"/rsp_test_zero", 0, 1, ctypes.byref(hp)) == capi.RSP_ERR_BAD_ARG
    assert L.rsp_shared_host_close(None, 64, None) == capi.RSP_OK            # closing nothing is fine
    assert L.rsp_copy_to_host_async(None, None, -1, None) == capi.RSP_ERR_BAD_ARG
    assert L.rsp_copy_to_host_async(None, None, 0, None) == capi.RSP_OK
    can = ctypes.c_int(-1)
    assert L.rsp_device_can_access_peer(3, 3, ctypes.byref(can)) == capi.RSP_OK and can.value == 1   # a device reaches itself
    assert L.rsp_device_can_access_peer(0, 1, None) == capi.RSP_ERR_BAD_ARG
    inf = capi.rccl_info()                                                   # which RCCL this process runs: answered without a device
    assert inf["version"] > 20000 and "rccl" in inf["library"]
    for key in ("auto_plans_made", "auto_plans_freed", "auto_plans_retired", "auto_plans_recycled", "fold_fixup"):
        assert capi.debug_get(key) >= 0
    assert capi.debug_get("fold_fixup") == 0 and capi.debug_get("auto_plan") == 1      # the defaults
    with pytest.raises(capi.RspError):
        capi.debug_set("auto_plans_made", 3)                                 # read-only
