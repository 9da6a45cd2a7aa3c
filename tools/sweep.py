#!/usr/bin/env python3
"""Tuning sweep on one GPU: workload x chunk_rows -> ms, GB/s (algorithmic), % of 8 TB/s.
C2 (92 MB) fits in the 256 MiB Infinity Cache, so it is timed over a rotation of
distinct copies of x (>= 320 MB in total) to read from HBM; both numbers are printed."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from bench import WORKLOADS, build_offsets, SEED
from rcppsparse_amd import capi


def time_calls(xs, pt, out, ws, reps):
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for k in range(3):
        capi.column_sums_device(xs[k % len(xs)], pt, out, ws)
    torch.cuda.synchronize()
    for k, (a, b) in enumerate(evs):
        a.record()
        capi.column_sums_device(xs[k % len(xs)], pt, out, ws)
        b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in evs)
    return t[len(t) // 2], t[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workloads", default="c3,c5,c2")
    ap.add_argument("--chunk-rows", default="0,16,32,64,128,256,512,1024")
    ap.add_argument("--reps", type=int, default=20)
    a = ap.parse_args()
    capi.load()
    rows = []
    for wl in a.workloads.split(","):
        nrow, ncol, nnz, shape, p = build_offsets(wl, 0)
        pt = torch.from_numpy(p).cuda()
        ncopies = max(1, int(np.ceil(400e6 / (8.0 * nnz))))
        xs = []
        for k in range(ncopies):
            x = torch.empty(nnz, dtype=torch.float64, device="cuda")
            capi.gen_values_device(x, SEED + k, 0, 0)
            xs.append(x)
        out = torch.empty(ncol, dtype=torch.float64, device="cuda")
        B = 8 * nnz + 4 * (ncol + 1) + 8 * ncol
        for cr in [int(v) for v in a.chunk_rows.split(",")]:
            capi.set_tuning(cr)
            ws = capi.alloc_workspace(ncol, nnz)
            med, best = time_calls(xs, pt, out, ws, a.reps)
            row = {"workload": wl, "chunk_rows": cr, "copies": ncopies, "ms_median": med, "ms_min": best,
                   "GBps_median": B / med / 1e6, "frac_of_8TBps": B / med / 1e6 / 8000}
            if ncopies > 1:
                m1, b1 = time_calls(xs[:1], pt, out, ws, a.reps)
                row["ms_median_cache_resident"] = m1
            rows.append(row)
            print(json.dumps(row), flush=True)
        capi.set_tuning(0)
        del xs
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
