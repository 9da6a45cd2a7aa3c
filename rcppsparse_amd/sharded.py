"""Column-range data parallelism for columnSums: one process per GPU.

Columns are independent units of the reference loop (src/example.cpp:28), so the
path shards by contiguous, nnz-balanced column ranges (``rsp_partition_columns``)
with exactly one exchange step: a gatherv of the disjoint per-shard output
slices to rank 0 (RCCL over xGMI on GPUs; grouped ncclSend/ncclRecv in
``rsp_comm_gatherv``).  No column is ever split across GPUs, so there is no
reduction between ranks and the result is independent of the rank count up to
the per-shard chunking (each shard is summed by the same single-GPU kernel).

The compute step and the gather are injected, so the same driver runs
  * on GPUs:  compute = C-ABI ``rsp_column_sums_device``, gather = RCCL;
  * in the world_size-2 ``gloo`` CPU tests: compute = whatever the test passes
    in, gather = ``torch.distributed.gather`` over gloo.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from . import capi


@dataclass
class Shard:
    rank: int
    world: int
    c0: int            # first owned column
    c1: int            # one past the last owned column
    x0: int            # first owned element of x  (= p[c0])
    x1: int            # one past the last owned element (= p[c1])
    p_local: np.ndarray   # rebased offsets, int32, length c1 - c0 + 1
    bounds: np.ndarray    # all column bounds, int32, length world + 1

    @property
    def ncol(self) -> int:
        return self.c1 - self.c0

    @property
    def nnz(self) -> int:
        return self.x1 - self.x0


def make_shard(p: np.ndarray, rank: int, world: int, balance: str = "nnz") -> Shard:
    """Contiguous column range of `rank` (pure integer, host): nnz-balanced
    (rsp_partition_columns) or, as the naive comparator, equal column counts."""
    p = np.ascontiguousarray(p, dtype=np.int32)
    if balance == "nnz":
        bounds = capi.partition_columns(p, world)
    elif balance == "cols":
        ncol = len(p) - 1
        bounds = np.array([(k * ncol) // world for k in range(world + 1)], dtype=np.int32)
    else:
        raise ValueError(balance)
    c0, c1 = int(bounds[rank]), int(bounds[rank + 1])
    return Shard(rank, world, c0, c1, int(p[c0]), int(p[c1]), capi.rebase_offsets(p, c0, c1), bounds)


def gather_layout(bounds: np.ndarray):
    """counts / displacements (in doubles) of every rank's output slice."""
    counts = np.diff(bounds).astype(np.int64)
    displs = bounds[:-1].astype(np.int64)
    return counts, displs


def imbalance(p: np.ndarray, bounds: np.ndarray) -> float:
    """max / mean nnz per shard (1.0 = perfect)."""
    per = np.diff(np.asarray(p, dtype=np.int64)[bounds])
    return float(per.max() / max(per.mean(), 1e-300))


class GlooGather:
    """Host-side stand-in for the RCCL gatherv (CPU tests): torch.distributed.gather."""

    def __init__(self, counts, root: int = 0):
        import torch.distributed as dist
        self.dist, self.counts, self.root = dist, [int(c) for c in counts], root

    def __call__(self, send, recv):
        import torch
        dist = self.dist
        rank, world = dist.get_rank(), dist.get_world_size()
        width = max(self.counts) if self.counts else 0
        padded = torch.zeros(max(width, 1), dtype=torch.float64)
        padded[:send.numel()] = send
        bufs = [torch.zeros_like(padded) for _ in range(world)] if rank == self.root else None
        dist.gather(padded, bufs, dst=self.root)
        if rank == self.root:
            off = 0
            for r in range(world):
                recv[off:off + self.counts[r]] = bufs[r][:self.counts[r]]
                off += self.counts[r]


class RcclGather:
    """RCCL gatherv through the C ABI (rsp_comm_gatherv) on the current torch stream."""

    def __init__(self, comm: capi.Comm, counts, displs, root: int = 0):
        self.comm, self.counts, self.displs, self.root = comm, counts, displs, root

    def __call__(self, send, recv):
        self.comm.gatherv(send, recv, self.counts, self.displs, self.root)


class ShardedColumnSums:
    """columnSums of one shard + gather of all shards' slices to rank 0.

    compute(shard) -> per-shard sums (tensor of shard.ncol doubles)
    gather(send, recv) -> fills recv (rank 0 only) from every rank's send
    """

    def __init__(self, shard: Shard, compute, gather):
        self.shard, self.compute, self.gather = shard, compute, gather

    def step(self, recv):
        local = self.compute(self.shard)
        if self.shard.world > 1:
            self.gather(local, recv)
        return local
