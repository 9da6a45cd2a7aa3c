#!/usr/bin/env python3
"""The tall (f64 MFMA) crossprod on 1e6 rows x <ncol> columns, a few launches: the program rocprofv3 is pointed at
for its counters (tools/pmc_crossprod.sh).   python3 tools/run_crossprod_tall.py [ncol] [reps]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from rcppsparse_amd import capi

ncol = int(sys.argv[1]) if len(sys.argv) > 1 else 256
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
L = capi.load()
nrow, nnz = 1_000_000, 500_000 * ncol
p = np.linspace(0, nnz, ncol + 1).astype(np.int64).astype(np.int32)
pt = torch.from_numpy(p).cuda()
xt = torch.empty(nnz, dtype=torch.float64, device="cuda")
it = torch.empty(nnz, dtype=torch.int32, device="cuda")
capi.gen_values_device(xt, 3, 0, 0)
capi.gen_row_indices_device(it, pt, nrow, 3)
out = torch.empty((ncol, ncol), dtype=torch.float64, device="cuda")
ws = torch.empty(int(L.rsp_crossprod_workspace_bytes(nrow, ncol, nnz)), dtype=torch.uint8, device="cuda")
ts = []
for _ in range(reps):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); capi.crossprod_device(xt, it, pt, nrow, out, workspace=ws); b.record()
    torch.cuda.synchronize()
    ts.append(a.elapsed_time(b))
print(json.dumps({"ncol": ncol, "ms": sorted(ts)[len(ts) // 2]}))
