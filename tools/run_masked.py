#!/usr/bin/env python3
"""Row-restricted column sums (rsp_column_sums_in_rows_device) on a workload shape, a few launches: the program
rocprofv3 is pointed at for the counters of "next" row f4 (tools/pmc_masked.sh).  Prints the HIP-event time.
    python3 tools/run_masked.py [workload] [nrow_override] [reps]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from bench import build_offsets, SEED
from rcppsparse_amd import capi

wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
nrow_o = int(sys.argv[2]) if len(sys.argv) > 2 else 0
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
capi.load()
if len(sys.argv) > 4:
    capi.set_experiment(int(sys.argv[4]))
nrow, ncol, nnz, shape, p = build_offsets(wl, 0)
nrow = nrow_o or nrow
pt = torch.from_numpy(p).cuda()
xt = torch.empty(nnz, dtype=torch.float64, device="cuda")
it = torch.empty(nnz, dtype=torch.int32, device="cuda")
capi.gen_values_device(xt, SEED, 0, 0)
capi.gen_row_indices_device(it, pt, nrow, SEED)
out = torch.empty(ncol, dtype=torch.float64, device="cuda")
ws = torch.empty(capi.in_rows_workspace_bytes(nrow, ncol, nnz), dtype=torch.uint8, device="cuda")   # (RSP_ROW_SLICES=0: the general form)
bits = torch.from_numpy(np.random.default_rng(0).integers(0, 2**32, size=(nrow + 31) // 32, dtype=np.uint32)).cuda()
ts = []
for _ in range(reps):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    capi.column_sums_in_rows_device(xt, it, pt, nrow, bits, False, out, ws)
    b.record()
    torch.cuda.synchronize()
    ts.append(a.elapsed_time(b))
ts.sort()
print(json.dumps({"variant": int(sys.argv[4]) if len(sys.argv) > 4 else 0, "workload": wl, "nrow": nrow, "nnz": nnz, "bitmap_bytes": int(bits.numel() * 4), "ms_median": ts[len(ts) // 2],
                  "algorithmic_GBps": (12 * nnz + 12 * ncol) / ts[len(ts) // 2] / 1e6}))
