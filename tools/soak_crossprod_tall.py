#!/usr/bin/env python3
"""Soak of crossprod's tall (f64 MFMA) form on one GPU: random matrices of 1..256 columns whose average
column length puts them on that form, with columns of very different lengths (some empty), rows spread over
the whole matrix, clustered into a few 64-row panels, shared by all columns, or (from 16 columns on) striped -- every
panel holding entries of a few column tiles only; against the oracle's merges
within 1e-12 * sum|x1 x2| per entry, bit-stable, symmetric; a quarter of the cases hold infinities / NaNs (round 5):
the exact kernels that stand by behind the tall form's flag answer, equal to the oracle entry for entry.

    python3 tools/soak_crossprod_tall.py [seconds] [seed] [mincol]
    python3 tools/soak_crossprod_tall.py 0 seed mincol case [dump.npz]        (replay one case of a seed)

mincol (default 1): the smallest number of columns drawn; 193 keeps every case on the panel-table kernel of
16 column tiles, 257 on its 24- and 32-tile forms (257..512 columns, 16-row panels).  The cost model is bypassed (RSP_CROSSPROD_TALL_ALWAYS=1): every case takes the tall form.
"""
import sys, os, time
os.environ["RSP_CROSSPROD_TALL_ALWAYS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import oracle
from rcppsparse_amd import capi


def striped(rng, ncol):
    """kind 4 (round 5, after the has[] defect): panel by panel of 16 / 32 rows, entries in the columns of zero to three
    column tiles drawn at random, runs of empty panels between them, and the matrix's first entry somewhere in the
    middle of the rows -- every workgroup of a panel range keeps meeting panels that hold nothing of its own tiles."""
    ph = 16 if ncol > 256 else (32 if ncol > 96 else 64)            # (the panel heights of the three kernels)
    ntile = (ncol + 15) // 16
    fill = float(rng.uniform(0.5, 0.95))
    kmax = min(3, ntile)
    per_panel = 0.5 * kmax / ntile * ph * fill                         # entries per column and panel, on average
    npanels = int(4300 / per_panel * 1.7 * rng.uniform(1.0, 1.5)) + 8     # (about 38 % of the panels fall into gaps)
    nrow = npanels * ph - int(rng.integers(0, ph))                     # (the last panel partial)
    rows_of = [[] for _ in range(ncol)]
    P = 0
    while P < npanels:
        if rng.random() < 0.02:
            P += int(rng.integers(1, 60))                             # a gap
            continue
        for t in rng.choice(ntile, size=int(rng.integers(0, kmax + 1)), replace=False):
            for c in range(16 * t, min(16 * t + 16, ncol)):
                r = P * ph + np.flatnonzero(rng.random(ph) < fill)
                rows_of[c].append(r[r < nrow])
        P += 1
    # column 0 starts late: its first row (entry 0 of the matrix) lies well inside the rows
    cut = int(nrow * rng.uniform(0.2, 0.8))
    cols = [np.concatenate(r) if r else np.zeros(0, np.int64) for r in rows_of]
    cols[0] = cols[0][cols[0] >= cut]
    lens = np.array([len(c) for c in cols], dtype=np.int64)
    short = 4097 * ncol - int(lens.sum())
    if short > 0:                                                      # keep the average on the tall side: one dense stretch in the longest column
        c = int(np.argmax(lens))
        extra = np.setdiff1d(np.arange(nrow), cols[c])[:short]
        cols[c] = np.sort(np.concatenate([cols[c], extra]))
        lens[c] = len(cols[c])
    i = np.concatenate(cols).astype(np.int32)
    p = np.concatenate(([0], np.cumsum(lens))).astype(np.int32)
    x = rng.standard_normal(i.size) * np.exp(rng.uniform(-20, 20))
    nonfinite = rng.random() < 0.15
    if nonfinite:
        x[rng.integers(0, i.size, int(rng.integers(1, 4)))] = rng.choice([np.inf, -np.inf, np.nan])
    return x, i, p, nrow, ncol, 4, nonfinite


def make(rng, mincol=1):
    """one random case: everything drawn here, nothing computed (so that a case can be replayed by its number)"""
    ncol = int(rng.choice([1, 2, 15, 16, 17, 31, 33, 48, 64, 65, 80, 96, 97, 128, 129, 160, 192, 193, 256, int(rng.integers(1, 257))]))
    if ncol < mincol:
        top = 512 if mincol > 256 else 256
        ncol = int(rng.choice([mincol, top, int(rng.integers(mincol, top + 1))]))
    mean_len = int(rng.integers(4096, 40000)) if ncol > 40 else int(rng.integers(4096, 90000))
    if ncol > 256:
        mean_len = int(rng.integers(4096, 6000))                      # (the oracle's merges: ncol^2 x length)
    kind = int(rng.integers(0, 5)) if ncol >= 16 else int(rng.integers(0, 4))
    if kind == 4:
        return striped(rng, ncol)
    lens = rng.integers(0, 2 * mean_len, ncol)
    if ncol > 2 and kind != 3:
        lens[rng.integers(0, ncol, max(1, ncol // 8))] = 0          # some empty columns
    lens = np.maximum(lens, 0)
    need = 4096 * ncol + ncol - int(lens.sum())
    if need > 0:
        lens[int(np.argmax(lens))] += need                            # keep the average on the tall side
    span = int(lens.max())
    if kind == 0:
        nrow = int(span * rng.uniform(1.0, 30.0)) + 1                 # spread
    elif kind == 1:
        nrow = span + int(rng.integers(0, 64))                        # nearly dense columns
    elif kind == 2:
        nrow = int(span * rng.uniform(2.0, 6.0)) + 1                  # clustered: each column in its own window
    else:
        nrow = int(span * rng.uniform(1.0, 3.0)) + 1                  # all columns share most rows
    cols = []
    base = np.sort(rng.choice(nrow, size=span, replace=False)) if kind == 3 else None
    for c in range(ncol):
        n = int(lens[c])
        if n == 0:
            cols.append(np.zeros(0, np.int64))
        elif kind == 2:
            lo = int(rng.integers(0, nrow - n + 1)) if nrow > n else 0
            width = min(nrow - lo, int(n * rng.uniform(1.0, 1.5)) + 1)
            cols.append(np.sort(lo + rng.choice(width, size=n, replace=False)))
        elif kind == 3:
            cols.append(np.sort(rng.choice(base, size=n, replace=False)))
        else:
            cols.append(np.sort(rng.choice(nrow, size=n, replace=False)))
    i = np.concatenate(cols).astype(np.int32)
    p = np.concatenate(([0], np.cumsum(lens))).astype(np.int32)
    x = rng.standard_normal(i.size) * np.exp(rng.uniform(-20, 20))
    nonfinite = i.size > 0 and rng.random() < 0.25
    if nonfinite:          # the tall form raises its flag and the exact kernels, which stood by, produce the reference's result
        x[rng.integers(0, i.size, int(rng.integers(1, 4)))] = rng.choice([np.inf, -np.inf, np.nan])
    return x, i, p, nrow, ncol, kind, nonfinite


def one(rng, case, mincol=1, dump=None):
    x, i, p, nrow, ncol, kind, nonfinite = make(rng, mincol)
    if dump:
        np.savez(dump, x=x, i=i, p=p, nrow=nrow)
    ref = oracle.crossprod(x, i, p)
    xt, it, pt = torch.from_numpy(x).cuda(), torch.from_numpy(i).cuda(), torch.from_numpy(p).cuda()
    got = capi.crossprod_device(xt, it, pt, nrow).cpu().numpy().T
    again = capi.crossprod_device(xt, it, pt, nrow).cpu().numpy().T
    if nonfinite:
        ok = np.array_equal(got, ref, equal_nan=True) and np.array_equal(again, ref, equal_nan=True)
        if not ok:
            print(f"FAIL case {case} (non-finite values): {nrow}x{ncol} nnz {i.size} kind {kind}", flush=True)
        return ok
    scale = oracle.crossprod(np.abs(x), i, p)
    ok = (got.tobytes() == again.tobytes() and np.array_equal(got, got.T)
          and bool(np.all(np.abs(got - ref) <= 1e-12 * scale)) and bool(np.all(got[scale == 0] == 0)))
    if not ok:
        print(f"FAIL case {case}: {nrow}x{ncol} nnz {i.size} kind {kind} "
              f"max {np.max(np.abs(got - ref) / np.maximum(scale, 1e-300))}", flush=True)
    return ok


def main():
    secs = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    mincol = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    capi.load()
    rng = np.random.default_rng(seed)
    if len(sys.argv) > 4:                      # replay ONE case of this seed: [seconds] seed mincol case [dump.npz]
        case = int(sys.argv[4])
        for _ in range(case):
            make(rng, mincol)
        ok = one(rng, case, mincol, dump=sys.argv[5] if len(sys.argv) > 5 else None)
        print(f"soak_crossprod_tall: case {case} of seed {seed}, mincol {mincol}: {'ok' if ok else 'FAILED'}", flush=True)
        sys.exit(0 if ok else 1)
    t0, n, bad, last = time.time(), 0, 0, time.time()
    while time.time() - t0 < secs:
        bad += not one(rng, n, mincol)
        n += 1
        if time.time() - last > 30:
            print(f"... {n} cases, {bad} failures", flush=True)
            last = time.time()
    print(f"soak_crossprod_tall: {n} cases, {bad} failures, seed {seed}, mincol {mincol}", flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
