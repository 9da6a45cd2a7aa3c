#!/usr/bin/env python3
"""Timing of the "next"-row entries that reuse the main kernel template on one GPU (C3 shape):
column reductions with a transform (8 B/nnz) and row-restricted column sums (x + i = 12 B/nnz)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from bench import build_offsets, SEED
from rcppsparse_amd import capi


def ev_time(fn, reps=15):
    for _ in range(3):
        fn()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[len(ts) // 2]


def main():
    capi.load()
    wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
    nrow, ncol, nnz, shape, p = build_offsets(wl, 0)
    pt = torch.from_numpy(p).cuda()
    xt = torch.empty(nnz, dtype=torch.float64, device="cuda")
    it = torch.empty(nnz, dtype=torch.int32, device="cuda")
    capi.gen_values_device(xt, SEED, 0, 0)
    capi.gen_row_indices_device(it, pt, nrow, SEED)
    out = torch.empty(ncol, dtype=torch.float64, device="cuda")
    ws = capi.alloc_workspace(ncol, nnz)
    B8 = 8 * nnz + 12 * ncol
    B12 = 12 * nnz + 12 * ncol
    res = {"workload": wl, "nnz": nnz}
    for name, op in (("sum", 0), ("sum_squares", 1), ("sum_abs", 2)):
        ms = ev_time(lambda: capi.column_reduce_device(xt, pt, op, out, ws))
        res[name] = {"ms": ms, "GBps": B8 / ms / 1e6, "frac_of_8TBps": B8 / ms / 8e9}
    rng = np.random.default_rng(0)
    bits = torch.from_numpy(rng.integers(0, 2**32, size=(nrow + 31) // 32, dtype=np.uint32)).cuda()
    for name, comp in (("in_rows", False), ("not_in_rows", True)):
        ms = ev_time(lambda: capi.column_sums_in_rows_device(xt, it, pt, nrow, bits, comp, out, ws))
        res[name] = {"ms": ms, "GBps_12B_per_nnz": B12 / ms / 1e6, "frac_of_8TBps": B12 / ms / 8e9}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
