#!/usr/bin/env python3
"""Cost of a device-made plan (rsp_column_sums_plan_create_device) on the BASELINE shapes: host time of the call
(allocations + enqueue; nothing waits), device time of the inspection (events around its kernels; the plan's own
inspect_ms), and the time from "plan requested" to "first planned call finished" against two general calls.
    python tools/measure_device_plan.py [--reps 30]          (on the GPU box)
Prints one JSON line per shape; profiles/r04_device_plan.json keeps a run."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["RCPPSPARSE_REQUIRE_GPU"] = "1"

import numpy as np   # noqa: E402
import torch         # noqa: E402

from rcppsparse_amd import capi, synth   # noqa: E402

SHAPES = {
    "c2": (1_000_000, 1_000_000, 10_000_000, "uniform"),
    "m10_1e8": (10_000_000, 10_000_000, 100_000_000, "uniform"),
    "vignette": (100_000, 1_000, 10_000_000, "uniform"),
    "c4shard": (10_000_000, 125_000, 125_000_000, "uniform"),
    "c3": (10_000_000, 1_000_000, 1_000_000_000, "uniform"),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--shapes", default="c2,m10_1e8,vignette,c4shard,c3")
    args = ap.parse_args()
    capi.load()
    dev = torch.device("cuda", 0)
    s = torch.cuda.Stream()
    torch.cuda.set_stream(s)
    for name in args.shapes.split(","):
        nrow, ncol, nnz, _ = SHAPES[name]
        p = synth.offsets_from_counts(synth.uniform_counts(ncol, nnz, 42, nrow))
        pt = torch.from_numpy(p).to(dev)
        x = torch.empty(nnz, dtype=torch.float64, device=dev)
        capi.gen_values_device(x, 42, 0, 0)
        out = torch.empty(ncol, dtype=torch.float64, device=dev)
        ws = capi.alloc_workspace(ncol, nnz, dev)
        capi.ColumnSumsPlan(pt, nnz=nnz, stream=s).wait().close()          # first use loads the kernels
        host_ms, dev_ms, form = [], [], None
        for _ in range(args.reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            plan = capi.ColumnSumsPlan(pt, nnz=nnz, stream=s)
            host_ms.append((time.perf_counter() - t0) * 1e3)
            plan.wait()
            dev_ms.append(plan.inspect_ms)
            form = plan.form
            plan.close()
        # host-made plan of the same offsets, for scale (D2H of p[] not included: p is already on the host here)
        t0 = time.perf_counter()
        hp = capi.ColumnSumsPlan(p, nnz=nnz)
        host_plan_ms = (time.perf_counter() - t0) * 1e3
        # request -> first planned result, nothing waited for in between except at the very end
        general = capi.prepared_column_sums(x, pt, out, ws, stream=s)
        for _ in range(5):
            general()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        general(); general()
        torch.cuda.synchronize()
        two_general_ms = (time.perf_counter() - t0) * 1e3
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record(s)
        plan = capi.ColumnSumsPlan(pt, nnz=nnz, stream=s)
        plan.wait()                                      # (a caller that wants the planned form for its very first call)
        plan.column_sums(x, pt, out, ws, stream=s)
        e1.record(s)
        torch.cuda.synchronize()
        plan_and_first_ms = e0.elapsed_time(e1)
        eg0, eg1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        eg0.record(s)
        general(); general()
        eg1.record(s)
        torch.cuda.synchronize()
        plan.close()
        hp.close()
        med = lambda v: sorted(v)[len(v) // 2]   # noqa: E731
        print(json.dumps({"shape": name, "ncol": ncol, "nnz": nnz, "form": {3: "columns", 2: "lean", 1: "snapped", 0: "general"}[form],
                          "device_plan_ms_device_time_median": med(dev_ms), "device_plan_ms_device_time_min": min(dev_ms),
                          "device_plan_ms_host_call_median": med(host_ms),
                          "host_plan_ms_from_a_host_copy": host_plan_ms,
                          "plan_then_first_planned_call_ms_stream_time": plan_and_first_ms,
                          "two_general_calls_ms_stream_time": eg0.elapsed_time(eg1),
                          "two_general_calls_ms_wall": two_general_ms, "reps": args.reps}), flush=True)
        del x, out, ws, pt
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
