#!/usr/bin/env python3
"""rsp_csc_upload with the plan inspection running beside the copies: wall time of the upload against the
inspection's own time (a plan made separately from the same p[]) on short-column matrices.
    python3 tools/measure_upload_plan.py"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bench import build_offsets, SEED
from rcppsparse_amd import capi, synth

capi.load()
for wl in ("c2", "m10_3e7", "m10_1e8"):
    nrow, ncol, nnz, shape, p = build_offsets(wl, 0)
    x = synth.gen_values(nnz, SEED, 0)
    capi.DeviceCSC(x[:1000], np.array([0, 1000], dtype=np.int32), (1000, 1)).close()   # warm the runtime
    ups = []
    for _ in range(3):
        t0 = time.perf_counter()
        h = capi.DeviceCSC(x, p, (nrow, ncol))
        ups.append((time.perf_counter() - t0) * 1e3)
        s = h.column_sums()
        h.close()
    plan = capi.ColumnSumsPlan(p)
    print(json.dumps({"workload": wl, "ncol": ncol, "nnz": nnz, "upload_with_plan_ms": round(min(ups), 2),
                      "inspection_alone_ms": round(plan.inspect_ms, 2), "plan_form": plan.form,
                      "x_bytes_over_upload_GBps": round(8 * nnz / min(ups) / 1e6, 1)}))
    plan.close()
