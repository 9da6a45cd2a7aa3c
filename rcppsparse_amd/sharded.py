"""Column-range data parallelism for columnSums: one process per GPU.

Columns are independent units of the reference loop (src/example.cpp:28), so the
path shards by contiguous, nnz-balanced column ranges (``rsp_partition_columns``)
with exactly one exchange step: a gatherv of the disjoint per-shard output
slices to rank 0 (RCCL over xGMI on GPUs; grouped ncclSend/ncclRecv in
``rsp_comm_gatherv``).  No column is ever split across GPUs, so there is no
reduction between ranks and the result is independent of the rank count up to
the per-shard chunking (each shard is summed by the same single-GPU kernel).

The compute step and the gather are injected, so the same driver runs
  * on GPUs (``bench.py``):  compute = C-ABI ``rsp_column_sums_device``,
    gather = ``RcclGather`` (or ``TorchGather`` if the C-ABI communicator
    cannot be created; ``HostStagedGather`` in the rehearsal mode where
    several ranks share one GPU);
  * in the world_size-2 ``gloo`` CPU tests: compute = whatever the test passes
    in, gather = ``GlooGather``.

Two drivers:
  * ``ShardedColumnSums``  -- one call = compute, then gather, in order on one
    stream: the synchronous-call semantics of reference src/example.cpp:26-32.
    This is the protocol ``bench.py`` reports as ``value`` at every N.
  * ``PipelinedColumnSums`` -- GPU only: launches alternate over several compute
    streams and the gather of call k runs on its own stream beside the kernel of
    call k+1.  Reported by ``bench.py`` as a separate, labelled figure.
"""
from __future__ import annotations

import contextlib
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


# ------------------------------------------------------------------ gather backends
class GlooGather:
    """Host-side stand-in for the RCCL gatherv (CPU tests): torch.distributed.gather."""
    name = "torch.distributed.gather (gloo, CPU tests)"

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
    """RCCL gatherv through the C ABI (rsp_comm_gatherv), enqueued on `stream` (None = the
    current torch stream at call time).  The ctypes arguments are converted once per send
    buffer, so a call in a hot loop is a single foreign call."""
    name = "rsp_comm_gatherv (C ABI, RCCL)"

    def __init__(self, comm: capi.Comm, counts, displs, root: int = 0, stream=None):
        self.comm, self.counts, self.displs, self.root, self.stream = comm, counts, displs, root, stream
        self._prepared = {}

    def __call__(self, send, recv):
        if self.stream is None:
            self.comm.gatherv(send, recv, self.counts, self.displs, self.root)
            return
        key = (send.data_ptr(), 0 if recv is None else recv.data_ptr())
        run = self._prepared.get(key)
        if run is None:
            run = self.comm.prepared_gatherv(send, recv, self.counts, self.displs, self.root, stream=self.stream)
            self._prepared[key] = run
        run()


class TorchGather:
    """The same gatherv (grouped RCCL send/recv) issued through torch.distributed's own
    communicator; bench.py uses it only if the C-ABI communicator cannot be created, and says
    so in ``config.gather``.  Works over gloo too (CPU test of the layout)."""
    name = "torch.distributed batch_isend_irecv (fallback: the C-ABI communicator failed)"

    def __init__(self, dist, rank, world, counts, displs, root: int = 0, stream=None):
        self.dist, self.rank, self.world = dist, rank, world
        self.counts, self.displs, self.root, self.stream = counts, displs, root, stream

    def __call__(self, send, recv):
        import contextlib
        import torch
        dist, root, counts, displs = self.dist, self.root, self.counts, self.displs
        ctx = torch.cuda.stream(self.stream) if self.stream is not None else contextlib.nullcontext()
        with ctx:
            if self.rank == root:
                ops = [dist.P2POp(dist.irecv, recv[int(displs[r]):int(displs[r] + counts[r])], r)
                       for r in range(self.world) if r != root and counts[r] > 0]
                recv[int(displs[root]):int(displs[root] + counts[root])].copy_(send, non_blocking=True)
            else:
                ops = [dist.P2POp(dist.isend, send, root)] if send.numel() > 0 else []
            if ops:
                for req in dist.batch_isend_irecv(ops):
                    req.wait()


class HostStagedGather:
    """Rehearsal exchange for ranks that SHARE a device (``bench.py --rendezvous gloo``: the N > 1
    control flow of the bench with real HIP compute on a box with fewer GPUs than ranks).  RCCL
    refuses two ranks on one device, so every slice is copied to page-locked host memory, sent to
    the root over gloo and copied into the root's device result at its displacement.  Same
    counts / displacements as the RCCL gatherv, and like it the call returns with the exchange
    ordered after the kernels on `stream`; here it also waits for it (the host carries the data).
    Never the exchange of a measured multi-GPU figure; ``config.gather`` names it."""
    name = "gloo send/recv of host copies (rehearsal: ranks share a device; not RCCL)"

    def __init__(self, dist, rank, world, counts, displs, root: int = 0, stream=None):
        import torch
        self.torch, self.dist, self.rank, self.world, self.root = torch, dist, rank, world, root
        self.counts, self.displs, self.stream = [int(c) for c in counts], [int(d) for d in displs], stream
        pin = torch.cuda.is_available()
        if rank == root:
            self.host = {r: torch.empty(self.counts[r], dtype=torch.float64, pin_memory=pin)
                         for r in range(world) if r != root and self.counts[r] > 0}
        else:
            self.host = torch.empty(self.counts[rank], dtype=torch.float64, pin_memory=pin)

    def __call__(self, send, recv):
        torch, dist, root = self.torch, self.dist, self.root
        on_device = send.is_cuda
        stream = (self.stream or torch.cuda.current_stream()) if on_device else None
        if self.rank != root:
            if send.numel() == 0:
                return
            if on_device:
                with torch.cuda.stream(stream):
                    self.host.copy_(send, non_blocking=True)
                stream.synchronize()              # the kernels before it on the stream have finished
            else:
                self.host.copy_(send)
            dist.send(self.host, root)
            return
        reqs = [(r, dist.irecv(buf, src=r)) for r, buf in self.host.items()]
        d0, c0 = self.displs[root], self.counts[root]
        own = recv[d0:d0 + c0]
        ctx = torch.cuda.stream(stream) if on_device else contextlib.nullcontext()
        with ctx:
            if c0 > 0 and own.data_ptr() != send.data_ptr():
                own.copy_(send, non_blocking=True)
            for r, req in reqs:
                req.wait()
                recv[self.displs[r]:self.displs[r] + self.counts[r]].copy_(self.host[r], non_blocking=True)
        if on_device:
            stream.synchronize()                  # the staging buffers are reused by the next call


# ------------------------------------------------------------------------- drivers
class ShardedColumnSums:
    """columnSums of one shard, then the gather of all shards' slices to rank 0, in order.

    compute(shard) -> per-shard sums (tensor of shard.ncol doubles)
    gather(send, recv) -> fills recv (rank 0 only) from every rank's send; None = no
                          exchange step (a single shard that is not asked to rehearse it)
    """

    def __init__(self, shard: Shard, compute, gather):
        self.shard, self.compute, self.gather = shard, compute, gather

    def step(self, recv, on_computed=None):
        local = self.compute(self.shard)
        if on_computed is not None:
            on_computed()
        if self.gather is not None:
            self.gather(local, recv)
        return local


# ------------------------------------------------------------------ rowSums over column-range shards
class GlooReduceRows:
    """Host-side stand-in for rsp_comm_reduce_rows (CPU tests; the rehearsal with ranks sharing a GPU):
    every rank's partial vector goes to the root, which adds them in RANK order -- the same sum, term
    for term, as the device form's slice-wise add (rows_add_partials_kernel)."""
    name = "gloo gather of the partial vectors, added in rank order on the root"

    def __init__(self, dist, rank, world, root: int = 0):
        self.dist, self.rank, self.world, self.root = dist, rank, world, root

    def __call__(self, partial, result, ncol_for_means: int = 0):
        import torch
        host = partial.detach().to("cpu").contiguous()
        bufs = [torch.empty_like(host) for _ in range(self.world)] if self.rank == self.root else None
        self.dist.gather(host, bufs, dst=self.root)
        if self.rank != self.root:
            return None
        total = bufs[0].numpy().copy()
        for k in range(1, self.world):
            total = total + bufs[k].numpy()          # one rounding per add, rank order
        total = total + 0.0
        if ncol_for_means:
            total = total / float(ncol_for_means)    # RcppSparse.h:153-154
        result.copy_(torch.from_numpy(total))
        return result


class RcclReduceRows:
    """rsp_comm_reduce_rows through the C ABI: slices all-to-all over xGMI, rank-ordered add, gatherv."""
    name = "rsp_comm_reduce_rows (C ABI, RCCL)"

    def __init__(self, comm: capi.Comm, nrow: int, device, root: int = 0, stream=None):
        import torch
        self.comm, self.root, self.stream = comm, root, stream
        self.ws = torch.empty(capi.reduce_rows_workspace_bytes(comm.nranks, nrow), dtype=torch.uint8, device=device)

    def __call__(self, partial, result, ncol_for_means: int = 0):
        return self.comm.reduce_rows(partial, result, self.root, self.ws, ncol_for_means, stream=self.stream)


class ShardedRowSums:
    """Matrix::rowSums / rowMeans (reference RcppSparse.h:138-156) of a column-range sharded matrix.

    Unlike columnSums, a shard does not own output elements: every shard's columns touch every row, so
    the exchange step is a REDUCE of f64[nrow] (80 MB at nrow = 1e7), not a gather of slices.
    compute(shard) -> partial row sums of the shard's columns (nrow doubles; rsp_row_sums_device on the
                      shard's x / i slices);
    reduce(partial, result, ncol_for_means) -> on the root, the partials added in rank order (= column
                      order, the order the reference's scatter loop meets the entries in).
    The result does not depend on the topology or on how many ranks there are beyond the blocking of
    that one sum, and is bit-identical from run to run."""

    def __init__(self, shard: Shard, compute, reduce):
        self.shard, self.compute, self.reduce = shard, compute, reduce

    def step(self, result, ncol_for_means: int = 0, on_computed=None):
        partial = self.compute(self.shard)
        if on_computed is not None:
            on_computed()
        return self.reduce(partial, result, ncol_for_means)


class PipelinedColumnSums:
    """GPU only.  Call n runs its kernel on compute stream n % S into output buffer n % B and its
    gather on the communication stream, so the gather of call n overlaps the kernel of call n+1
    and, with S > 1, call n+1 fills the chip while call n drains.  Every stream needs its own
    carries workspace (include/rcppsparse_hip.h); a buffer is reused only after its previous
    gather has drained.

    launches[q][k]() enqueues the column sums on compute stream q into outs[k];
    gathers[k]() enqueues the gather of outs[k] on comm_stream (None entries = no exchange).
    """

    def __init__(self, torch, launches, gathers, compute_streams, comm_stream, nbuf: int):
        self.launches, self.gathers = launches, gathers
        self.compute_streams, self.comm_stream, self.nbuf = compute_streams, comm_stream, nbuf
        self.n = 0
        # events are created up front: the hot loop only records and waits
        self.kernel_done = [torch.cuda.Event() for _ in range(nbuf)]
        self.buffer_free = [torch.cuda.Event() for _ in range(nbuf)]
        self.freed_on = [None] * nbuf      # stream that recorded buffer_free[k] (None = never used)

    def step(self):
        n = self.n
        k, q = n % self.nbuf, n % len(self.compute_streams)
        sc = self.compute_streams[q]
        self.n = n + 1
        # outs[k] may be overwritten once whatever read it last (its gather, or with no exchange the
        # kernel that wrote it, if that ran on another stream) has finished
        if self.freed_on[k] is not None and self.freed_on[k] is not sc and not self.buffer_free[k].query():
            sc.wait_event(self.buffer_free[k])
        self.launches[q][k](n)
        if self.gathers[k] is not None:
            self.kernel_done[k].record(sc)
            self.comm_stream.wait_event(self.kernel_done[k])
            self.gathers[k]()
            self.buffer_free[k].record(self.comm_stream)
            self.freed_on[k] = self.comm_stream
        else:
            self.buffer_free[k].record(sc)
            self.freed_on[k] = sc
