#!/usr/bin/env python3
"""Timing of Matrix::crossprod on the device for the shape of the reference's vignette benchmark
(rsparsematrix(100000, 1000, 0.1)) and two neighbours, with the oracle's pairwise-merge loop
(1 thread) timed on a column subset and scaled by the number of column pairs."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import oracle
from rcppsparse_amd import capi, synth


def main():
    capi.load()
    shapes = ((100_000, 1000, 10_000_000), (100_000, 4000, 8_000_000), (1_000_000, 500, 5_000_000))
    if len(sys.argv) > 1:                       # e.g. "0" or "0,2": a subset of the shapes
        shapes = [shapes[int(k)] for k in sys.argv[1].split(",")]
    for nrow, ncol, nnz in shapes:
        p = synth.offsets_from_counts(synth.uniform_counts(ncol, nnz, seed=42, nrow=nrow))
        pt = torch.from_numpy(p).cuda()
        xt = torch.empty(nnz, dtype=torch.float64, device="cuda")
        it = torch.empty(nnz, dtype=torch.int32, device="cuda")
        capi.gen_values_device(xt, 42, 0, 0)
        capi.gen_row_indices_device(it, pt, nrow, 42)
        out = torch.empty((ncol, ncol), dtype=torch.float64, device="cuda")
        ws = torch.empty(capi.load().rsp_crossprod_workspace_bytes(nrow, ncol, nnz), dtype=torch.uint8, device="cuda")

        def timed(**kw):
            capi.crossprod_device(xt, it, pt, nrow, out, **kw)
            torch.cuda.synchronize()
            ts = []
            for _ in range(5):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); capi.crossprod_device(xt, it, pt, nrow, out, **kw); b.record()
                torch.cuda.synchronize()
                ts.append(a.elapsed_time(b))
            return sorted(ts)[2]

        ms_tiles = timed(tiles=True)
        tiles_out = out.clone()
        ms = timed(workspace=ws)          # includes building the row-major form
        same = bool(torch.equal(tiles_out, out))
        # CPU: first 60 columns -> 1830 pairs, scaled to ncol*(ncol+1)/2 pairs
        sub = 60
        xs = oracle.gen_values(int(p[sub]), 42, 0, 0)
        is_ = oracle.gen_row_indices(p, nrow, 42, 0, sub)
        t0 = time.perf_counter()
        ref = oracle.crossprod(xs, is_, np.ascontiguousarray(p[:sub + 1]))
        cpu_s = (time.perf_counter() - t0) * (ncol * (ncol + 1) / 2) / (sub * (sub + 1) / 2)
        got = out[:sub, :sub].cpu().numpy().T
        pairs = ncol * (ncol + 1) // 2
        print(json.dumps({"shape": f"{nrow}x{ncol}, nnz {nnz}", "gpu_ms": ms, "gpu_ms_tile_kernel": ms_tiles,
                          "rows_and_tiles_same_bits": same, "column_pairs": pairs,
                          "pairs_per_s": pairs / ms * 1e3, "cpu_oracle_1thread_s_scaled": cpu_s,
                          "speedup_vs_1thread": cpu_s / (ms * 1e-3),
                          "first_60x60_bit_exact": bool(np.array_equal(got, ref))}), flush=True)


if __name__ == "__main__":
    main()
