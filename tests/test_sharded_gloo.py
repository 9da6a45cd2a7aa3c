"""N > 1 path on CPU: world_size-2 (and 4) gloo runs of the column-range sharded
driver (rcppsparse_amd/sharded.py).  The per-shard compute is injected: here the
oracle stands in for the HIP kernel (there is no GPU in this container), the
gather goes over gloo instead of RCCL; partitioning, rebasing, slice layout and
reassembly are the code the GPU ranks run."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle
from rcppsparse_amd import sharded, synth


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _matrix(shape):
    if shape == "uniform":
        counts = synth.uniform_counts(3000, 120_000, seed=11, nrow=None)
    elif shape == "zipf":
        counts = synth.zipf_counts(3000, 120_000, seed=11, nrow=20_000)
    else:   # leading/trailing empties and one giant column
        counts = np.concatenate([np.zeros(50), [90_000], np.full(100, 3), np.zeros(75)]).astype(np.int64)
    p = synth.offsets_from_counts(counts)
    return p, synth.gen_values(int(p[-1]), seed=12, kind=0)


def _worker(rank, world, port, shape, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        p, x = _matrix(shape)
        shard = sharded.make_shard(p, rank, world)
        counts, _ = sharded.gather_layout(shard.bounds)

        def compute(sh):   # stand-in for rsp_column_sums_device on this rank's HBM shard
            return torch.from_numpy(oracle.column_sums(x[sh.x0:sh.x1], sh.p_local))

        recv = torch.empty(len(p) - 1, dtype=torch.float64) if rank == 0 else None
        driver = sharded.ShardedColumnSums(shard, compute, sharded.GlooGather(counts))
        local = driver.step(recv)
        assert local.numel() == shard.ncol
        if rank == 0:
            ref = oracle.column_sums(x, p)
            q.put(("ok", bool(recv.numpy().tobytes() == ref.tobytes()), sharded.imbalance(p, shard.bounds)))
    except Exception as e:   # pragma: no cover
        if rank == 0:
            q.put(("err", repr(e), 0.0))
        raise
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
@pytest.mark.parametrize("shape", ["uniform", "zipf", "giant"])
def test_sharded_columnsums_over_gloo(world, shape):
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    mp.spawn(_worker, args=(world, _free_port(), shape, q), nprocs=world, join=True)
    status, same, imb = q.get()
    assert status == "ok" and same
    if shape == "uniform":
        assert imb < 1.05


def _eight_worker(rank, world, port, partition, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        p, x = _matrix("zipf")
        shard = sharded.make_shard(p, rank, world, balance=partition)
        counts, displs = sharded.gather_layout(shard.bounds)

        def compute(sh):   # stand-in for rsp_column_sums_device on this rank's HBM shard
            return torch.from_numpy(oracle.column_sums(x[sh.x0:sh.x1], sh.p_local))

        recv = torch.empty(len(p) - 1, dtype=torch.float64) if rank == 0 else None
        # the exchange of bench.py --rendezvous gloo (host copies over gloo), the driver of `value`
        driver = sharded.ShardedColumnSums(shard, compute, sharded.HostStagedGather(dist, rank, world, counts, displs, 0))
        for _ in range(3):                       # back-to-back calls reuse the staging buffers
            driver.step(recv)
        everyone = [None] * world
        dist.all_gather_object(everyone, (shard.c0, shard.c1, shard.x0, shard.x1))
        if rank == 0:
            ref = oracle.column_sums(x, p)
            q.put((bool(recv.numpy().tobytes() == ref.tobytes()), everyone, [int(c) for c in counts], [int(d) for d in displs],
                   sharded.imbalance(p, shard.bounds)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("partition", ["nnz", "cols"])
def test_eight_ranks_over_gloo_both_partitions(partition):
    """World size 8 -- the driver's node -- on CPU: eight processes, the 8-entry counts / displacements, the nnz-balanced
    cut of SURVEY.md 8e (bounds[k] = lower_bound(p, k * nnz / 8)) and the naive equal-column-count comparator, on a Zipf
    matrix; the gathered result has the oracle's bits and every rank reports the range the formula gives it."""
    world = 8
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    mp.spawn(_eight_worker, args=(world, _free_port(), partition, q), nprocs=world, join=True)
    same, everyone, counts, displs, imb = q.get()
    assert same
    p, _ = _matrix("zipf")
    ncol, nnz = len(p) - 1, int(p[-1])
    if partition == "nnz":
        want = [0] + [int(np.searchsorted(p, (k * nnz) // world, side="left")) for k in range(1, world)] + [ncol]
        assert imb <= 1.0 + int(np.diff(p).max()) / (nnz / world)      # no column is split: at most one column over the mean
    else:
        want = [(k * ncol) // world for k in range(world + 1)]
    assert [e[0] for e in everyone] + [ncol] == want
    assert all(a[1] == b[0] and a[3] == b[2] for a, b in zip(everyone, everyone[1:]))        # the ranges tile columns and x
    assert [e[2] for e in everyone] == [int(p[c]) for c in want[:-1]]
    assert counts == [b - a for a, b in zip(want, want[1:])] and displs == want[:-1] and sum(counts) == ncol


def test_shards_tile_the_matrix_exactly():
    p, _ = _matrix("zipf")
    for world in (1, 2, 3, 8):
        shards = [sharded.make_shard(p, r, world) for r in range(world)]
        assert shards[0].c0 == 0 and shards[-1].c1 == len(p) - 1
        assert shards[0].x0 == 0 and shards[-1].x1 == p[-1]
        for a, b in zip(shards, shards[1:]):
            assert a.c1 == b.c0 and a.x1 == b.x0
        counts, displs = sharded.gather_layout(shards[0].bounds)
        assert counts.sum() == len(p) - 1 and np.array_equal(displs, np.cumsum(counts) - counts)


def _torch_gather_worker(rank, world, port, which, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        counts = [5, 0, 7][:world] if world == 3 else [4, 9]
        displs = np.cumsum([0] + counts[:-1])
        send = torch.full((counts[rank],), float(rank + 1), dtype=torch.float64)
        recv = torch.zeros(sum(counts), dtype=torch.float64) if rank == 0 else None
        # the same driver bench.py runs, with the fallback gather as its exchange step
        seen = []
        gather_cls = {"torch": sharded.TorchGather, "staged": sharded.HostStagedGather}[which]
        driver = sharded.ShardedColumnSums(None, lambda _shard: send,
                                           gather_cls(dist, rank, world, counts, displs, 0))
        assert driver.step(recv, on_computed=lambda: seen.append(1)) is send and seen == [1]
        if rank == 0:
            want = np.concatenate([np.full(c, r + 1.0) for r, c in enumerate(counts)])
            q.put(bool(np.array_equal(recv.numpy(), want)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("which", ["torch", "staged"])
@pytest.mark.parametrize("world", [2, 3])
def test_bench_fallback_gatherv_layout(world, which):
    """sharded.TorchGather (bench.py uses it only if the C-ABI communicator cannot be created) and
    sharded.HostStagedGather (bench.py --rendezvous gloo: ranks sharing a device) fill the same
    counts/displacements layout as the RCCL gatherv, including an empty slice."""
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    mp.spawn(_torch_gather_worker, args=(world, _free_port(), which, q), nprocs=world, join=True)
    assert q.get()


# ------------------------------------------------------------------ rowSums over column-range shards
def _row_matrix():
    m = synth.rsparsematrix(700, 90, density=0.08, seed=21)
    return m["x"], m["i"], m["p"], 700


def _rowsums_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        x, i, p, nrow = _row_matrix()
        shard = sharded.make_shard(p, rank, world)

        def compute(sh):   # stand-in for rsp_row_sums_device on this rank's x / i slices
            return torch.from_numpy(oracle.row_sums(x[sh.x0:sh.x1], i[sh.x0:sh.x1], sh.p_local, nrow))

        result = torch.empty(nrow, dtype=torch.float64) if rank == 0 else None
        driver = sharded.ShardedRowSums(shard, compute, sharded.GlooReduceRows(dist, rank, world))
        driver.step(result)
        means = torch.empty(nrow, dtype=torch.float64) if rank == 0 else None
        driver.step(means, ncol_for_means=len(p) - 1)
        if rank == 0:
            q.put((result.numpy().copy(), means.numpy().copy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_rowsums_reduce_in_rank_order_over_gloo(world):
    """rowSums of a column-range sharded matrix (reference RcppSparse.h:138-144): every rank sums the rows of
    ITS columns, the partial vectors are added in rank order on the root.  That is a blocking of the
    reference's own column-major scatter order, so: bit-identical to adding the shards' oracle results in
    order, within 1e-12 * sum|x| per row of the oracle on the whole matrix, empty rows exactly +0.0."""
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    mp.spawn(_rowsums_worker, args=(world, _free_port(), q), nprocs=world, join=True)
    got, means = q.get()
    x, i, p, nrow = _row_matrix()
    blocked = None
    for r in range(world):
        sh = sharded.make_shard(p, r, world)
        part = oracle.row_sums(x[sh.x0:sh.x1], i[sh.x0:sh.x1], sh.p_local, nrow)
        blocked = part if blocked is None else blocked + part
    assert got.tobytes() == (blocked + 0.0).tobytes()
    ref = oracle.row_sums(x, i, p, nrow)
    scale = np.bincount(i, weights=np.abs(x), minlength=nrow)
    assert np.all(np.abs(got - ref) <= 1e-12 * scale)
    empty = np.bincount(i, minlength=nrow) == 0
    assert np.all(got[empty] == 0.0) and not np.any(np.signbit(got[empty]))
    assert means.tobytes() == (got / (len(p) - 1)).tobytes()        # RcppSparse.h:153-154
