#!/usr/bin/env python3
"""Soak of the one-shot rowSums kernels on one GPU: random row counts around the block (16384), tile
(22528), form (4, 832, 2 / 4 / 8 x 832 blocks: direct, partition, coarse-block, two-level) and bucket (512 blocks)
edges, uniform / clustered / single-block / mostly-invalid row indices,
against numpy's bincount; every case also run twice for bit-stability and once as rowMeans.

    python3 tools/soak_rowsums.py [seconds] [seed]
"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from rcppsparse_amd import capi

RTOL = 1e-12


def one(rng, case):
    # around the block (16384), form (4 / 832 / 2 x 832 / 4 x 832 / 8 x 832 blocks) and bucket (512 blocks) edges;
    # the big ones (coarse-block and two-level forms) on every 8th case: their outputs are 0.2-1.3 GB
    small = [1, 5, 16383, 16384, 16385, 32768, 65536, 65537, 100_000, 1_000_000, 8_388_608, 8_388_609, 13_631_488,
             13_631_489, int(rng.integers(1, 14_000_000))]
    big = [27_262_976, 27_262_977, 54_525_952, 54_525_953, 109_051_904, 109_051_905, 117_440_513,
           int(rng.integers(14_000_000, 170_000_000))]
    nrow = int(rng.choice(big if case % 8 == 7 else small))
    nnz = int(rng.choice([0, 1, 63, 64, 65, 22527, 22528, 22529, 157_696, 157_697,
                          int(rng.integers(1, 3_000_000))]))
    kind = int(rng.integers(0, 6))
    if kind == 0:
        i = rng.integers(0, nrow, nnz)
    elif kind == 1:                      # a few hot rows
        hot = rng.integers(0, nrow, 8)
        i = hot[rng.integers(0, 8, nnz)]
    elif kind == 2:                      # one block
        b = int(rng.integers(0, (nrow + 16383) // 16384))
        i = rng.integers(b * 16384, min(nrow, (b + 1) * 16384), nnz)
    elif kind == 3:                      # sorted rows (long runs per block)
        i = np.sort(rng.integers(0, nrow, nnz))
    elif kind == 4:                      # many invalid
        i = rng.integers(-nrow - 3, 2 * nrow + 3, nnz)
    else:                                # first / last rows only
        i = np.where(rng.random(nnz) < 0.5, 0, nrow - 1)
    i = i.astype(np.int32)
    x = rng.standard_normal(nnz) * np.exp(rng.uniform(-30, 30, 1))
    xt = torch.from_numpy(x).cuda() if nnz else torch.zeros(2, dtype=torch.float64, device="cuda")[:0]
    it = torch.from_numpy(i).cuda() if nnz else torch.zeros(2, dtype=torch.int32, device="cuda")[:0]
    got = capi.row_sums_device(xt, it, nrow).cpu().numpy()
    again = capi.row_sums_device(xt, it, nrow).cpu().numpy()
    ncol = int(rng.integers(1, 1000))
    means = capi.row_sums_device(xt, it, nrow, ncol_for_means=ncol).cpu().numpy()
    keep = (i >= 0) & (i < nrow)
    ref = np.bincount(i[keep], weights=x[keep], minlength=nrow)
    scale = np.bincount(i[keep], weights=np.abs(x[keep]), minlength=nrow)
    ok = (got.tobytes() == again.tobytes() and means.tobytes() == (got / ncol).tobytes()
          and bool(np.all(np.abs(got - ref) <= RTOL * scale)) and not np.any(np.signbit(got[scale == 0])))
    if ok and 0 < nnz <= 200_000 and nrow <= 2_000_000:     # the handle (keeps the regrouped copy, repeats the accumulate pass)
        pc = np.sort(rng.integers(0, nnz + 1, ncol - 1)).astype(np.int32) if ncol > 1 else np.zeros(0, np.int32)
        p = np.concatenate(([0], pc, [nnz])).astype(np.int32)
        h = capi.DeviceCSC(x, p, (nrow, ncol), i=i)
        hs, hs2 = h.row_sums(), h.row_sums()
        h.close()
        ok = (hs.tobytes() == hs2.tobytes() and bool(np.all(np.abs(hs - ref) <= RTOL * scale))
              and not np.any(np.signbit(hs[scale == 0])))
        if not ok:
            print(f"(handle) maxerr {np.max(np.abs(hs - ref))}", flush=True)
    if not ok:
        print(f"FAIL case {case}: nrow {nrow} nnz {nnz} kind {kind} maxerr {np.max(np.abs(got - ref))}", flush=True)
    return ok


def main():
    secs = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    capi.load()
    rng = np.random.default_rng(seed)
    t0, n, bad = time.time(), 0, 0
    last = t0
    while time.time() - t0 < secs:
        bad += not one(rng, n)
        n += 1
        if time.time() - last > 30:
            print(f"... {n} cases, {bad} failures", flush=True)
            last = time.time()
    print(f"soak_rowsums: {n} cases, {bad} failures, seed {seed}", flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
