#!/usr/bin/env python3
"""Timing of Matrix::crossprod on the device for the shape of the reference's vignette benchmark
(rsparsematrix(100000, 1000, 0.1)) and two neighbours, with the oracle's pairwise-merge loop
(1 thread) timed on a column subset and scaled by the number of column pairs; `tall` as an argument
times the tall (f64 MFMA) form against the bit-identical one on few-columns shapes instead."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import oracle
from rcppsparse_amd import capi, synth


def tall(only_tall=False, shapes=None):
    L = capi.load()
    for nrow, ncol, nnz in shapes or ((1_000_000, 64, 32_000_000), (10_000_000, 16, 80_000_000), (2_000_000, 128, 128_000_000),
                                      (4_000_000, 48, 190_000_000), (1_000_000, 192, 96_000_000), (1_000_000, 256, 128_000_000),
                                      (45_000_000, 48, 2**31 - 1)):
        p = np.linspace(0, nnz, ncol + 1).astype(np.int64).astype(np.int32)
        pt = torch.from_numpy(p).cuda()
        xt = torch.empty(nnz, dtype=torch.float64, device="cuda")
        it = torch.empty(nnz, dtype=torch.int32, device="cuda")
        capi.gen_values_device(xt, 3, 0, 0)
        capi.gen_row_indices_device(it, pt, nrow, 3)
        out = torch.empty((ncol, ncol), dtype=torch.float64, device="cuda")
        ws = torch.empty(int(L.rsp_crossprod_workspace_bytes(nrow, ncol, nnz)), dtype=torch.uint8, device="cuda")
        capi.crossprod_device(xt, it, pt, nrow, out, workspace=ws)
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); capi.crossprod_device(xt, it, pt, nrow, out, workspace=ws); b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        row = {"shape": f"{nrow}x{ncol}, nnz {nnz}", "tall_form_ms": sorted(ts)[2], "workspace_GB": ws.numel() / 1e9,
               "products": float(nnz) / nrow * nnz}
        if nnz < 2**31 - 1 and not only_tall:   # (the bit-identical form takes 25 s there)
            tall_out = out.clone()
            capi.set_crossprod_exact(True)
            ws2 = torch.empty(int(L.rsp_crossprod_workspace_bytes(nrow, ncol, nnz)), dtype=torch.uint8, device="cuda")
            capi.crossprod_device(xt, it, pt, nrow, out, workspace=ws2)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            capi.crossprod_device(xt, it, pt, nrow, out, workspace=ws2)
            torch.cuda.synchronize()
            row["bit_identical_form_ms"] = (time.perf_counter() - t0) * 1e3
            capi.set_crossprod_exact(False)
            row["max_abs_diff_over_max_abs"] = ((tall_out - out).abs().max() / out.abs().max()).item()
            del ws2
        print(json.dumps(row), flush=True)
        del xt, it, ws
        torch.cuda.empty_cache()


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "tall":
        return tall()
    if len(sys.argv) > 1 and sys.argv[1] == "tallfast":    # the tall form only, 1e6 rows x the given column counts
        cols = [int(a) for a in sys.argv[2:]] or [128, 160, 192, 224, 256]
        return tall(True, [(1_000_000, c, 500_000 * c) for c in cols])
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
