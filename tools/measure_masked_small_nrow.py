#!/usr/bin/env python3
"""Row-restricted column sums (rsp_column_sums_in_rows_device) when the matrix has few rows, the
shape the restricted iterators are used on (features x samples): the row bitmap is a few KB
and stays in L1, unlike the 1e7-row C3 shape of measure_next_rows.py."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from rcppsparse_amd import capi, synth


def main():
    capi.load()
    for nrow, ncol, nnz in ((30_000, 100_000, 500_000_000), (200_000, 1_000_000, 500_000_000),
                            (500_000, 1_000_000, 500_000_000), (900_000, 1_000_000, 500_000_000),
                            (1_000_000, 1_000_000, 500_000_000), (10_000_000, 1_000_000, 500_000_000)):
        p = synth.offsets_from_counts(synth.uniform_counts(ncol, nnz, seed=42, nrow=nrow))
        pt = torch.from_numpy(p).cuda()
        xt = torch.empty(nnz, dtype=torch.float64, device="cuda")
        it = torch.empty(nnz, dtype=torch.int32, device="cuda")
        capi.gen_values_device(xt, 42, 0, 0)
        capi.gen_row_indices_device(it, pt, nrow, 42)
        rng = np.random.default_rng(1)
        rows = np.flatnonzero(rng.random(nrow) < 0.5)
        bm = torch.from_numpy(capi.row_set_bitmap(rows, nrow)).cuda()
        out = torch.empty(ncol, dtype=torch.float64, device="cuda")
        ws = capi.alloc_workspace(ncol, nnz)
        res = {}
        for name, fn in (("plain", lambda: capi.column_sums_device(xt, pt, out, ws)),
                         ("in_rows", lambda: capi.column_sums_in_rows_device(xt, it, pt, nrow, bm, False, out, ws))):
            fn(); torch.cuda.synchronize()
            ts = []
            for _ in range(7):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); fn(); b.record(); torch.cuda.synchronize()
                ts.append(a.elapsed_time(b))
            res[name] = sorted(ts)[3]
        B = 12 * nnz + 12 * ncol
        print(json.dumps({"shape": f"{nrow}x{ncol}, nnz {nnz}", "bitmap_bytes": int(bm.numel() * 4),
                          "plain_ms": res["plain"], "in_rows_ms": res["in_rows"],
                          "in_rows_GBps_12B_per_nnz": B / res["in_rows"] / 1e6,
                          "frac_of_8TBps": B / res["in_rows"] / 1e6 / 8000}), flush=True)
        del xt, it, out, ws
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
