#!/usr/bin/env python3
"""No-cliff check on structures the BASELINE configs do not cover: mostly-empty columns, every
element its own column, one giant column among singletons, alternating long/empty."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from rcppsparse_amd import capi, synth


def ev_time(fn, reps=10):
    for _ in range(3):
        fn()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts) // 2]


def main():
    capi.load()
    rng = np.random.default_rng(0)
    cases = {}
    n = 50_000_000
    c = np.zeros(n, dtype=np.int64); c[rng.choice(n, 10_000_000, replace=False)] = 1
    cases["5e7 columns, 1e7 singletons, 80% empty"] = c
    cases["1e8 columns of exactly 1"] = np.ones(100_000_000, dtype=np.int64)
    c = np.ones(20_000_001, dtype=np.int64); c[10_000_000] = 500_000_000
    cases["one 5e8 column between 2e7 singletons"] = c
    c = np.zeros(2_000_000, dtype=np.int64); c[::2] = 400
    cases["1e6 columns of 400 alternating with empty ones"] = c
    c = rng.integers(0, 3, size=60_000_000).astype(np.int64)
    cases["6e7 columns of 0..2"] = c
    for name, counts in cases.items():
        p = synth.offsets_from_counts(counts)
        nnz, ncol = int(p[-1]), len(counts)
        xt = torch.empty(nnz, dtype=torch.float64, device="cuda")
        capi.gen_values_device(xt, 1, 0, 1)
        pt = torch.from_numpy(p).cuda()
        out = torch.empty(ncol, dtype=torch.float64, device="cuda")
        ws = capi.alloc_workspace(ncol, nnz)
        ms = ev_time(lambda: capi.column_sums_device(xt, pt, out, ws))
        B = 8 * nnz + 12 * ncol
        # spot parity: total of sums == total of x (all positive)
        tot = float(out.sum().item()); ref = float(xt.sum().item())
        print(json.dumps({"case": name, "nnz": nnz, "ncol": ncol, "ms": ms, "algorithmic_GBps": B / ms / 1e6,
                          "frac_of_8TBps": B / ms / 8e9, "rel_err_total": abs(tot - ref) / ref}), flush=True)
        del xt, pt, out, ws
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
