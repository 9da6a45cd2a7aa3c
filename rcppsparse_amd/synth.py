"""Synthetic dgCMatrix inputs shaped like ``Matrix::rsparsematrix`` output.

BASELINE.json configs:
  C1  10 x 10, density 0.1                 (README.md:33-38 of the reference)
  C2  1e6 x 1e6, nnz 1e7, uniform
  C3  1e7 x 1e6, nnz 1e9, uniform          (the headline config)
  C5  1e7 x 1e6, nnz 1e9, Zipf nnz/column

"uniform" = nnz positions uniform over nrow*ncol, so per-column counts are
multinomial (about Poisson(nnz/ncol)); rows inside a column distinct and
ascending.  Column offsets ``p`` are built on the host with a seeded numpy
generator (4 MB at ncol = 1e6); values come from a counter-based integer hash
(``gen_values`` here, ``rsp_gen_values_device`` on the GPU, ``oracle_gen_values``
in the oracle: all three bit-identical), so an 8 GB ``x`` never has to cross
PCIe.  Row indices ``i`` are only materialised for small matrices: the hot path
never reads them (reference RcppSparse.h:227 ``row()`` is not called by
src/example.cpp:28-30).
"""
from __future__ import annotations

import numpy as np

_M1 = np.uint64(0x9E3779B97F4A7C15)
_M2 = np.uint64(0xBF58476D1CE4E5B9)
_M3 = np.uint64(0x94D049BB133111EB)
_K = np.uint64(0xD1342543DE82EF95)


def _mix64(z: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = z + _M1
        z = (z ^ (z >> np.uint64(30))) * _M2
        z = (z ^ (z >> np.uint64(27))) * _M3
        return z ^ (z >> np.uint64(31))


def gen_values(n: int, seed: int, first_idx: int = 0, kind: int = 0) -> np.ndarray:
    """x[k] = value(seed, first_idx + k); numpy twin of the device generator.

    kind 0: signed, two decimals, in [-5.10, 5.10] (bell-shaped: centred sum of
            four 8-bit uniforms / 100) -- a stand-in for rsparsematrix's default
            rounded-normal values;
    kind 1: U(0, 1) with 53 random bits, all positive.
    """
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64) + np.uint64(first_idx)
        h = _mix64(np.uint64(seed) * _K + idx)
    if kind == 1:
        return (h >> np.uint64(11)).astype(np.float64) * 2.0 ** -53
    b = np.uint64(255)
    s = ((h & b) + ((h >> np.uint64(8)) & b) + ((h >> np.uint64(16)) & b)
         + ((h >> np.uint64(24)) & b)).astype(np.int64)
    return (s - 510).astype(np.float64) / 100.0


def uniform_counts(ncol: int, nnz: int, seed: int, nrow: int | None = None) -> np.ndarray:
    """Per-column nnz for uniformly scattered positions (multinomial, clipped to nrow)."""
    rng = np.random.default_rng(seed)
    counts = rng.multinomial(nnz, np.full(ncol, 1.0 / ncol)).astype(np.int64)
    return _clip_redistribute(counts, nrow, rng)


def zipf_counts(ncol: int, nnz: int, seed: int, nrow: int, s: float = 1.0,
                order: str = "shuffled") -> np.ndarray:
    """Power-law column degrees: count(rank r) ~ r^-s, clipped to nrow, excess
    redistributed; order 'shuffled' (seeded permutation) or 'descending' (worst case)."""
    rng = np.random.default_rng(seed)
    w = np.arange(1, ncol + 1, dtype=np.float64) ** (-s)
    counts = np.floor(w / w.sum() * nnz).astype(np.int64)
    counts = _clip_redistribute(counts, nrow, rng, total=nnz)
    if order == "shuffled":
        counts = counts[rng.permutation(ncol)]
    elif order != "descending":
        raise ValueError(order)
    return counts


def _clip_redistribute(counts, nrow, rng, total=None):
    total = int(counts.sum()) if total is None else int(total)
    if nrow is not None:
        counts = np.minimum(counts, nrow)
    deficit = total - int(counts.sum())
    guard = 0
    while deficit > 0 and guard < 64:
        room = (nrow - counts) if nrow is not None else np.full_like(counts, deficit)
        open_idx = np.flatnonzero(room > 0)
        if open_idx.size == 0:
            raise ValueError("nnz exceeds nrow*ncol")
        add = np.zeros_like(counts)
        share, rem = divmod(deficit, open_idx.size)
        add[open_idx] = share
        if rem:
            add[rng.choice(open_idx, size=rem, replace=False)] += 1
        add = np.minimum(add, room)
        counts = counts + add
        deficit = total - int(counts.sum())
        guard += 1
    return counts


def offsets_from_counts(counts) -> np.ndarray:
    p = np.zeros(len(counts) + 1, dtype=np.int64)
    np.cumsum(counts, out=p[1:])
    if p[-1] > np.iinfo(np.int32).max:
        raise ValueError("nnz exceeds int32 (p[] is 32-bit, RcppSparse.h:30)")
    return p.astype(np.int32)


def row_indices(p: np.ndarray, nrow: int, seed: int) -> np.ndarray:
    """Valid CSC row indices (distinct, ascending inside each column). Small inputs only."""
    rng = np.random.default_rng(seed + 1)
    i = np.empty(int(p[-1]), dtype=np.int32)
    for c in range(len(p) - 1):
        k = int(p[c + 1] - p[c])
        if k:
            i[p[c]:p[c + 1]] = np.sort(rng.choice(nrow, size=k, replace=False))
    return i


def rsparsematrix(nrow: int, ncol: int, density: float | None = None, nnz: int | None = None,
                  seed: int = 42, kind: int = 0, with_i: bool = True):
    """Host-side synthetic dgCMatrix slots: dict(x, i, p, Dim).  Mirrors the call
    shape of Matrix::rsparsematrix(nrow, ncol, density) used in the reference docs."""
    if nnz is None:
        nnz = int(round(density * nrow * ncol))
    p = offsets_from_counts(uniform_counts(ncol, nnz, seed, nrow))
    x = gen_values(nnz, seed, 0, kind)
    i = row_indices(p, nrow, seed) if with_i else None
    return {"x": x, "i": i, "p": p, "Dim": np.array([nrow, ncol], dtype=np.int32)}
