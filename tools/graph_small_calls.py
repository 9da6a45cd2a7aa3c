"""Measure whether a HIP graph of many C2-sized calls beats issuing them one by one (it does not: DESIGN §11).
   gpurun -- python tools/graph_small_calls.py"""
import os, sys, time, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ["RCPPSPARSE_REQUIRE_GPU"] = "1"
import numpy as np, torch
from rcppsparse_amd import capi, synth
capi.load()
nrow, ncol, nnz = 1_000_000, 1_000_000, 10_000_000
p = synth.offsets_from_counts(synth.uniform_counts(ncol, nnz, 42, nrow))
pt = torch.from_numpy(p).cuda()
xs = []
for k in range(6):
    x = torch.empty(nnz, dtype=torch.float64, device="cuda"); capi.gen_values_device(x, 42 + k, 0, 0); xs.append(x)
out = torch.empty(ncol, dtype=torch.float64, device="cuda")
ws = capi.alloc_workspace(ncol, nnz)
plan = capi.ColumnSumsPlan(p, nnz=nnz)
s = torch.cuda.Stream(); torch.cuda.set_stream(s)
def timed(fn, reps):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record(s)
    for _ in range(reps): fn()
    b.record(s); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
res = {}
for name, mk in (("general", lambda x: capi.prepared_column_sums(x, pt, out, ws, stream=s)), ("lean", lambda x: plan.prepared(x, pt, out, ws, stream=s))):
    runs = [mk(x) for x in xs]
    def eager():
        for r in runs: r()
    res[name + "_eager_us"] = timed(eager, 50) / 6 * 1e3
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        for _ in range(5):
            for r in runs: r()
    res[name + "_graph_us"] = timed(g.replay, 20) / 30 * 1e3
print(json.dumps(res))
