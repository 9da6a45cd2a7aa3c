#!/usr/bin/env python3
"""crossprod at 97-256 columns: the panel-table matrix-core form (RSP_CROSSPROD_PANEL_TABLE=2: 8, 12 and 16 column tiles) against the round-3 kernel it replaces
(RSP_CROSSPROD_PANEL_TABLE=0), same inputs: times (HIP events, median) and the largest difference relative to
sum |x1 x2|.   gpurun -- python3 tools/compare_crossprod_panels.py [out.json]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from rcppsparse_amd import capi

L = capi.load()
rows = []
SHAPES = ((1_000_000, 256, 0.5), (1_000_000, 224, 0.5), (1_000_000, 200, 0.5), (1_000_000, 256, 0.1),
          (1_000_000, 256, 0.9), (4_000_000, 256, 0.05), (250_000, 256, 0.5),
          (1_000_000, 192, 0.5), (1_000_000, 160, 0.5), (1_000_000, 192, 0.1), (1_000_000, 192, 0.9), (4_000_000, 192, 0.05),
          (1_000_000, 128, 0.5), (1_000_000, 100, 0.5), (1_000_000, 128, 0.1), (1_000_000, 128, 0.9), (4_000_000, 128, 0.05))
for nrow, ncol, dens in SHAPES:
    per = int(nrow * dens)
    nnz = per * ncol
    p = (np.arange(ncol + 1, dtype=np.int64) * per).astype(np.int32)
    pt = torch.from_numpy(p).cuda()
    xt = torch.empty(nnz, dtype=torch.float64, device="cuda")
    it = torch.empty(nnz, dtype=torch.int32, device="cuda")
    capi.gen_values_device(xt, 3, 0, 0)
    capi.gen_row_indices_device(it, pt, nrow, 3)
    rec = {"nrow": nrow, "ncol": ncol, "density": dens}
    outs = {}
    for name, env in (("panel_table", "2"), ("round3_kernel", "0")):
        os.environ["RSP_CROSSPROD_PANEL_TABLE"] = env
        out = torch.empty((ncol, ncol), dtype=torch.float64, device="cuda")
        ws = torch.empty(int(L.rsp_crossprod_workspace_bytes(nrow, ncol, nnz)), dtype=torch.uint8, device="cuda")
        ts = []
        for _ in range(7):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); capi.crossprod_device(xt, it, pt, nrow, out, workspace=ws); b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        rec[name + "_ms"] = round(sorted(ts)[3], 4)
        outs[name] = out.clone()
        again = torch.empty_like(out)
        capi.crossprod_device(xt, it, pt, nrow, again, workspace=ws)
        rec[name + "_same_bits_again"] = bool(torch.equal(again, out))
        del ws
    os.environ.pop("RSP_CROSSPROD_PANEL_TABLE")
    a, b = outs["panel_table"], outs["round3_kernel"]
    rec["max_abs_diff"] = float((a - b).abs().max())
    rec["max_abs_value"] = float(b.abs().max())
    rec["symmetric"] = bool(torch.equal(a, a.T))
    rows.append(rec)
    print(json.dumps(rec), flush=True)
    del xt, it, outs
if len(sys.argv) > 1:
    json.dump(rows, open(sys.argv[1], "w"), indent=1)
