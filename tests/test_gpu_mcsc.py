"""The single-process multi-GPU handle (rsp_mcsc_*, csrc/multigpu.cpp): the path an R session reaches.

SURVEY.md 8e: contiguous nnz-balanced column ranges (reference src/example.cpp:28 -- columns are independent),
one gather of disjoint output slices.  Round 6 rebuilt the call so that it creates nothing (no thread, stream or
allocation per call); these tests hold every launch x gather combination to the BITS of the per-shard device calls
and of one another, on this box's one device (several shards share it; the RCCL gather, which needs a device per
shard, runs with one shard).
"""
import os
import threading

import numpy as np
import pytest

import oracle
from rcppsparse_amd import capi, synth

pytestmark = pytest.mark.gpu
RTOL = 1e-12


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    capi.load()
    return torch


def mixed_matrix(ncol=20_000, nnz=2_500_001, nrow=400_000, seed=12):
    counts = synth.zipf_counts(ncol, nnz, seed=seed, nrow=nrow)
    counts[::17] = 0
    p = synth.offsets_from_counts(counts)
    x = synth.gen_values(int(p[-1]), seed=seed, kind=0)
    return x, p, nrow, ncol


def per_shard_device_sums(torch, x, p, G):
    """What the handle must reproduce bit for bit: every shard through its own resident single-device handle."""
    bounds = capi.partition_columns(p, G)
    out = np.zeros(len(p) - 1)
    for k in range(G):
        c0, c1 = int(bounds[k]), int(bounds[k + 1])
        if c1 == c0:
            continue
        h = capi.DeviceCSC(x[p[c0]:p[c1]], capi.rebase_offsets(p, c0, c1), (1, c1 - c0))
        out[c0:c1] = h.column_sums()
        h.close()
    return out


@pytest.mark.parametrize("G", [1, 2, 3, 8])
def test_every_launch_and_gather_gives_the_per_shard_bits(torch_cuda, G):
    x, p, nrow, ncol = mixed_matrix()
    want = per_shard_device_sums(torch_cuda, x, p, G)
    scale = oracle.column_abs_sums(x, p)
    assert np.all(np.abs(want - oracle.column_sums(x, p)) <= RTOL * scale)
    h = capi.MultiDeviceCSC(x, p, (nrow, ncol), devices=[0] * G)
    try:
        cfg = h.config()
        # (a device per shard -- here: the single shard -- gets the copy kernel, shards sharing a device the copy command)
        assert cfg["gather"] == ("blit" if G == 1 else "d2h") and cfg["launch"] == ("workers" if G >= 3 else "serial")
        seen = {}
        for launch in ("serial", "workers"):
            for gather in ("d2h", "blit", "stores"):
                h.set_launch(launch)
                h.set_gather(gather)
                got = h.column_sums()
                assert got.tobytes() == want.tobytes(), (launch, gather)
                again = h.column_sums()
                assert again.tobytes() == got.tobytes()
                means = h.column_means()
                assert means.tobytes() == (want / nrow).tobytes(), (launch, gather)
                seen[(launch, gather)] = h.last_call_stamps()
        # workers exist once the mode has been used (G - 1 of them), and only then
        assert h.config()["workers"] == (G - 1 if G > 1 else 0)
        # no host copy at all: the caller reads the page-locked vector
        buf = h.result_buffer()
        out = h.column_sums(out=buf)
        assert out.tobytes() == want.tobytes()
        st = seen[("workers", "d2h")]
        assert st["call_us"] > 0 and len(st["done_us"]) == G
        assert all(b <= e <= d <= c for b, e, d, c in zip(st["begin_us"], st["enqueued_us"], st["done_us"], st["copied_us"]))
    finally:
        h.close()


def test_rccl_gather_over_comm_init_all_one_device(torch_cuda):
    """SURVEY 8e's collective in its single-process form: ncclCommInitAll over the handle's devices (here one), the own
    slice in place, one D2H of the whole vector.  More than one shard on one device must be refused with a message
    (RCCL cannot put two ranks of a communicator on one device), and the handle keeps working in its previous mode."""
    x, p, nrow, ncol = mixed_matrix(ncol=5000, nnz=600_000, nrow=100_000, seed=5)
    h = capi.MultiDeviceCSC(x, p, (nrow, ncol), devices=[0])
    try:
        want = h.column_sums()
        h.set_gather("rccl")
        assert h.config() == {"gather": "rccl", "launch": "serial", "workers": 0, "comms": 1}
        got = h.column_sums()
        assert got.tobytes() == want.tobytes()
        assert h.column_means().tobytes() == (want / nrow).tobytes()
        h.set_gather("blit")
        assert h.column_sums().tobytes() == want.tobytes()
    finally:
        h.close()
    h2 = capi.MultiDeviceCSC(x, p, (nrow, ncol), devices=[0, 0])
    try:
        with pytest.raises(capi.RspError) as e:
            h2.set_gather("rccl")
        assert e.value.code == capi.RSP_ERR_BAD_ARG and "one DEVICE per shard" in str(e.value)
        assert h2.config()["gather"] == "d2h"
        assert h2.column_sums().tobytes() == per_shard_device_sums(torch_cuda, x, p, 2).tobytes()
    finally:
        h2.close()


def test_wrap_device_shards_resident_in_hbm(torch_cuda):
    """rsp_mcsc_wrap_device: the shards' x / p already live in HBM (what bench.py's single-process figure uses); nothing
    is copied, the caller's tensors survive the handle, every shard is planned on the device."""
    torch = torch_cuda
    ncol, nnz, nrow, G = 64_000, 64_000_00, 1_000_000, 4
    p = synth.offsets_from_counts(synth.uniform_counts(ncol, nnz, seed=3, nrow=nrow))
    bounds = capi.partition_columns(p, G)
    xs, ps = [], []
    for k in range(G):
        c0, c1 = int(bounds[k]), int(bounds[k + 1])
        xt = torch.empty(int(p[c1] - p[c0]), dtype=torch.float64, device="cuda")
        capi.gen_values_device(xt, seed=3, first_idx=int(p[c0]), kind=0)
        xs.append(xt)
        ps.append(torch.from_numpy(capi.rebase_offsets(p, c0, c1)).cuda())
    torch.cuda.synchronize()
    h = capi.MultiDeviceCSC.wrap_device(xs, ps, nrow)
    try:
        assert h.dims() == (nrow, ncol, G)
        info = [h.shard_info(k) for k in range(G)]
        assert [s["c0"] for s in info] + [ncol] == [int(b) for b in bounds]
        got = h.column_sums()
        x = synth.gen_values(nnz, seed=3, kind=0)
        ref, scale = oracle.column_sums(x, p), oracle.column_abs_sums(x, p)
        assert np.all(np.abs(got - ref) <= RTOL * scale)
        assert h.column_sums().tobytes() == got.tobytes()
        assert h.shard_kernel_ms(1, reps=5) > 0
    finally:
        h.close()
    # the caller's memory is untouched and still the caller's
    assert float(xs[0][0]) == float(x[0]) and int(ps[-1][-1]) == int(p[-1] - p[int(bounds[-2])])
    with pytest.raises(capi.RspError):
        capi.MultiDeviceCSC.wrap_device([xs[0][1:]], [ps[0]], nrow)     # x not 16-byte aligned / sizes do not fit


def test_empty_shards_and_empty_matrix(torch_cuda):
    """More shards than columns with entries: empty ranges do nothing, in every mode; a matrix without columns returns."""
    p = np.array([0, 0, 3, 3, 3], dtype=np.int32)
    x = np.array([1.0, 2.0, 4.0])
    for launch in ("serial", "workers"):
        h = capi.MultiDeviceCSC(x, p, (5, 4), devices=[0] * 6)
        h.set_launch(launch)
        for gather in ("d2h", "blit", "stores"):
            h.set_gather(gather)
            assert h.column_sums().tolist() == [0.0, 7.0, 0.0, 0.0]
        h.close()
    h = capi.MultiDeviceCSC(np.array([], dtype=np.float64), np.zeros(1, dtype=np.int32), (5, 0), devices=[0, 0])
    assert h.column_sums().size == 0
    h.close()


def test_two_handles_from_two_threads_and_many_calls(torch_cuda):
    """Different handles may be used from different threads at once (include/rcppsparse_hip.h); 300 calls each through
    the parked workers, every result the same bits."""
    x, p, nrow, ncol = mixed_matrix(ncol=9000, nnz=900_000, nrow=200_000, seed=8)
    want = per_shard_device_sums(torch_cuda, x, p, 4)
    errors = []
    nthreads = lambda: len(os.listdir(f"/proc/{os.getpid()}/task"))   # noqa: E731
    h0 = capi.MultiDeviceCSC(x, p, (nrow, ncol), devices=[0] * 4)    # (whatever the runtime starts for itself exists now)
    h0.set_launch("workers")
    h0.column_sums()
    with_one_handle = nthreads()
    h0.close()
    before = nthreads()
    assert before == with_one_handle - 3                              # the handle's three parked workers are gone

    def body():
        try:
            h = capi.MultiDeviceCSC(x, p, (nrow, ncol), devices=[0] * 4)
            h.set_launch("workers")
            for _ in range(300):
                if h.column_sums().tobytes() != want.tobytes():
                    errors.append("mismatch")
                    break
            h.close()
        except Exception as e:   # noqa: BLE001
            errors.append(repr(e))

    ts = [threading.Thread(target=body) for _ in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors
    # the threads of both handles are gone again (nothing of a closed handle stays behind)
    assert nthreads() <= before


def test_a_forked_child_gets_an_error_not_a_hang(torch_cuda):
    """R code forks (parallel::mclapply).  A child inherits the handle's pointer but neither its parked worker threads nor a
    usable GPU context: a call there must come back with a message at once -- before round 6's fix it would have waited for
    threads that do not exist -- and releasing the handle in the child must leave the parent's alone."""
    x, p, nrow, ncol = mixed_matrix(ncol=3000, nnz=300_000, nrow=50_000, seed=21)
    h = capi.MultiDeviceCSC(x, p, (nrow, ncol), devices=[0] * 3)
    h.set_launch("workers")
    want = h.column_sums()
    r, w = os.pipe()
    pid = os.fork()
    if pid == 0:                                             # the child: no GPU call of its own before this one
        os.close(r)
        msg = b"no error"
        try:
            h.column_sums()
        except capi.RspError as e:
            msg = str(e).encode()[:200]
        except BaseException as e:   # noqa: BLE001
            msg = ("other: " + repr(e)).encode()[:200]
        try:
            h.close()
            os.write(w, msg)
        finally:
            os._exit(0)
    os.close(w)
    import select
    ready, _, _ = select.select([r], [], [], 30)
    assert ready, "the forked child did not answer within 30 s (it hangs)"
    said = os.read(r, 400).decode()
    os.close(r)
    os.waitpid(pid, 0)
    assert "does not survive a fork" in said, said
    assert h.column_sums().tobytes() == want.tobytes()       # the parent's handle is untouched
    h.close()


def test_a_forked_child_is_a_machine_without_a_device(torch_cuda):
    """The HIP runtime does not survive a fork.  A child of a process that has used the GPU through this library sees no
    device at the C ABI (rsp_device_count = 0, the host entries RSP_ERR_NO_DEVICE with a message, without touching the runtime)
    -- which is what lets the Rcpp layer above the ABI answer on the host in an mclapply child, as on a machine without a GPU.
    The parent is unaffected."""
    assert capi.device_count() >= 1
    x, p = np.array([1.0, 2.0, 4.0]), np.array([0, 1, 3], dtype=np.int32)
    assert capi.column_sums_host(x, p).tolist() == [1.0, 6.0]
    carried = capi.DeviceCSC(x, p, (5, 2))                   # a single-device handle made BEFORE the fork
    r, w = os.pipe()
    pid = os.fork()
    if pid == 0:
        os.close(r)
        said = []
        try:
            said.append(f"count={capi.device_count()}")
            for fn in (lambda: capi.column_sums_host(x, p), lambda: capi.column_sums_host_multi(x, p, devices=[0]),
                       lambda: capi.DeviceCSC(x, p, (5, 2)), lambda: capi.MultiDeviceCSC(x, p, (5, 2), devices=[0]),
                       lambda: carried.column_sums()):
                try:
                    fn()
                    said.append("answered")
                except capi.RspError as e:
                    said.append(f"code={e.code} fork={'fork' in str(e)}")
        except BaseException as e:   # noqa: BLE001
            said.append("other: " + repr(e)[:80])
        try:
            os.write(w, ";".join(said).encode())
        finally:
            os._exit(0)
    os.close(w)
    import select
    ready, _, _ = select.select([r], [], [], 30)
    assert ready, "the forked child did not answer within 30 s"
    said = os.read(r, 1000).decode()
    os.close(r)
    os.waitpid(pid, 0)
    want = "count=0;" + ";".join([f"code={capi.RSP_ERR_NO_DEVICE} fork=True"] * 4) + f";code={capi.RSP_ERR_HIP} fork=False"
    assert said == want, said                                 # (the carried handle: an error from its device guard, no hang)
    assert capi.device_count() >= 1 and capi.column_sums_host(x, p).tolist() == [1.0, 6.0]
    assert carried.column_sums().tolist() == [1.0, 6.0]      # the parent's handle is untouched
    carried.close()
