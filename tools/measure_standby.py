#!/usr/bin/env python3
"""rsp_crossprod_device on tall shapes (1e6 rows x ncol columns, 4096 ... 500000 entries per column), K calls back to
back on one stream: milliseconds per call.  The entry never synchronises, so behind the matrix-core form its exact
kernels stand by on a device flag (launch_crossprod_rows, crossprod.hip); a call of few columns is short enough for
those launches to show.    python3 tools/measure_standby.py [K]        (on the GPU box)"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from rcppsparse_amd import capi

K = int(sys.argv[1]) if len(sys.argv) > 1 else 50
L = capi.load()
rows = []
for nrow, ncol, per in ((1_000_000, 16, 500_000), (1_000_000, 32, 500_000), (1_000_000, 64, 500_000), (1_000_000, 96, 500_000),
                        (1_000_000, 128, 500_000), (1_000_000, 256, 500_000), (200_000, 32, 50_000), (200_000, 128, 50_000),
                        (4_000_000, 32, 500_000)):
    nnz = per * ncol
    p = np.linspace(0, nnz, ncol + 1).astype(np.int64).astype(np.int32)
    pt = torch.from_numpy(p).cuda()
    xt = torch.empty(nnz, dtype=torch.float64, device="cuda")
    it = torch.empty(nnz, dtype=torch.int32, device="cuda")
    capi.gen_values_device(xt, 3, 0, 0)
    capi.gen_row_indices_device(it, pt, nrow, 3)
    out = torch.empty((ncol, ncol), dtype=torch.float64, device="cuda")
    ws = torch.empty(int(L.rsp_crossprod_workspace_bytes(nrow, ncol, nnz)), dtype=torch.uint8, device="cuda")
    for _ in range(5):
        capi.crossprod_device(xt, it, pt, nrow, out, workspace=ws)
    ts = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(K):
            capi.crossprod_device(xt, it, pt, nrow, out, workspace=ws)
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / K)
    rows.append({"nrow": nrow, "ncol": ncol, "nnz": nnz, "form": capi.crossprod_form(nrow, ncol, nnz),
                 "ms_per_call": round(sorted(ts)[2], 4), "min": round(min(ts), 4)})
    print(json.dumps(rows[-1]), flush=True)
