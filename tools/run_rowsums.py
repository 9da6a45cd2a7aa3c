#!/usr/bin/env python3
"""rsp_row_sums_device on a workload shape, a few launches: the program rocprofv3 is pointed at for the counters and
kernel times of the row-wise path (tools/pmc_rowsums.sh).   python3 tools/run_rowsums.py [workload] [nrow] [reps]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import build_offsets, SEED
from rcppsparse_amd import capi

wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
nrow_o = int(sys.argv[2]) if len(sys.argv) > 2 else 0
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
L = capi.load()
nrow, ncol, nnz, shape, p = build_offsets(wl, 0)
nrow = nrow_o or nrow
pt = torch.from_numpy(p).cuda()
xt = torch.empty(nnz, dtype=torch.float64, device="cuda")
it = torch.empty(nnz, dtype=torch.int32, device="cuda")
capi.gen_values_device(xt, SEED, 0, 0)
capi.gen_row_indices_device(it, pt, nrow, SEED)
out = torch.empty(nrow, dtype=torch.float64, device="cuda")
ws = torch.empty(int(L.rsp_row_sums_workspace_bytes(nrow, nnz)), dtype=torch.uint8, device="cuda")
capi.row_sums_device(xt, it, nrow, out, ws)
torch.cuda.synchronize()
ts = []
for _ in range(reps):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); capi.row_sums_device(xt, it, nrow, out, ws); b.record()
    torch.cuda.synchronize()
    ts.append(a.elapsed_time(b))
ts.sort()
print(json.dumps({"workload": wl, "nrow": nrow, "nnz": nnz, "workspace_GB": ws.numel() / 1e9, "ms_median": ts[len(ts) // 2],
                  "algorithmic_GBps": (12 * nnz + 8 * nrow) / ts[len(ts) // 2] / 1e6}))
