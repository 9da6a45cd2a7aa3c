#!/usr/bin/env python3
"""What does the HOST spend per call of the device entries?  A matrix small enough for the GPU never to be the limit
(300 000 entries in 30 000 columns: a 3 us kernel), 20 000 calls enqueued back to back on one stream, the loop's wall
time per call -- through the plan-free entry (its own lean plan settled first), through a caller's plan, and with the
entry's planning switched off (two launches: main + fix-up kernel).  bench.py times C2-sized calls (16 us of kernel)
the same way, so a host that needs longer than that per call shows up there as kernel time.
    python3 tools/measure_host_enqueue.py        (on the GPU box)"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from rcppsparse_amd import capi, synth

capi.load()
ncol, per = 30_000, 10
p = synth.offsets_from_counts(np.full(ncol, per))
nnz = int(p[-1])
xt = torch.from_numpy(synth.gen_values(nnz, 3, kind=0)).cuda()
pt = torch.from_numpy(p).cuda()
out = torch.empty(ncol, dtype=torch.float64, device="cuda")
ws = capi.alloc_workspace(ncol, nnz)
stream = torch.cuda.Stream()
K = 20_000
res = {}


def loop(launch):
    for _ in range(200):
        launch()
    torch.cuda.synchronize()
    best = None
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(K):
            launch()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        us = 1e6 * (t1 - t0) / K
        best = us if best is None else min(best, us)
        drain = 1e6 * (t2 - t1) / K
    return {"host_us_per_call": round(best, 3), "drain_us_per_call_last_run": round(drain, 3)}


with torch.cuda.stream(stream):
    capi.debug_set("auto_min_nnz", 1)
    capi.set_auto_plan(True)
    f = capi.prepared_column_sums(xt, pt, out, ws, stream=stream)
    for _ in range(50):
        f()
    torch.cuda.synchronize()
    res["form_of_the_entrys_own_plan"] = capi.column_sums_device_settle(pt, nnz, stream=stream)
    res["plan_free_entry_own_plan"] = loop(f)
    capi.set_auto_plan(False)
    res["plan_free_entry_planning_off_two_launches"] = loop(f)
    plan = capi.ColumnSumsPlan(p, nnz)
    g = plan.prepared(xt, pt, out, ws, stream=stream)
    res["callers_plan_form"] = plan.form if hasattr(plan, "form") else None
    res["callers_plan"] = loop(g)
    capi.set_auto_plan(True)
print(json.dumps(res))
