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
    assert L.rsp_set_tuning(-1) == capi.RSP_ERR_BAD_ARG and L.rsp_set_experiment(-1) == capi.RSP_ERR_BAD_ARG
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
