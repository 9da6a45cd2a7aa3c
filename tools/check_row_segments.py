#!/usr/bin/env python3
"""Row sums behind a handle: the segments form (no regrouped copy; every row block reads its own pieces of the
columns) against the forms it replaces (direct / partition) on the same matrix -- agreement relative to the row's
sum of |x|, identical bits on a second call, first-call and repeated-call wall times (upload excluded, the copy
of the result to the host included).
    python3 tools/check_row_segments.py nrow,ncol,nnz [reps]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from rcppsparse_amd import capi, synth

nrow, ncol, nnz = (int(float(v)) for v in sys.argv[1].split(","))
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
capi.load()
p = synth.offsets_from_counts(synth.uniform_counts(ncol, nnz, 7, nrow))
pt = torch.from_numpy(p).cuda()
xt = torch.empty(nnz, dtype=torch.float64, device="cuda")
it = torch.empty(nnz, dtype=torch.int32, device="cuda")
capi.gen_values_device(xt, 7, 0, 0)
capi.gen_row_indices_device(it, pt, nrow, 7)
l1 = torch.zeros(nrow, dtype=torch.float64, device="cuda").index_add_(0, it.long(), xt.abs()).cpu().numpy()
x, i = xt.cpu().numpy(), it.cpu().numpy()
del xt, it
torch.cuda.empty_cache()
res = {"nrow": nrow, "ncol": ncol, "nnz": nnz}
outs = {}
# every kernel of both paths once on small matrices of the same row count, so that no first call below pays for
# loading code (hundreds of ms for the first launch of a kernel in a process)
for mode in (0, 2):
    capi.set_row_segments(mode)
    wp = synth.offsets_from_counts(np.full(64, 3000, dtype=np.int64))
    wi = np.tile(np.sort(np.random.default_rng(0).choice(nrow, 3000, replace=False)).astype(np.int32), 64)
    wh = capi.DeviceCSC(np.ones(64 * 3000), wp, (nrow, 64), i=wi)
    wh.row_sums()
    wh.row_means()
    wh.close()
for mode, name in ((0, "other"), (1, "segments")):
    capi.set_row_segments(mode)
    h = capi.DeviceCSC(x, p, (nrow, ncol), i=i)
    t0 = time.perf_counter()
    first = h.row_sums()
    t_first = (time.perf_counter() - t0) * 1e3
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        again = h.row_sums()
        ts.append((time.perf_counter() - t0) * 1e3)
    assert first.tobytes() == again.tobytes()
    outs[name] = first
    res[name] = {"form": h.row_form(), "first_call_ms": round(t_first, 3), "repeated_call_ms": round(min(ts), 3),
                 "GBps_of_12B_per_nnz": round(12 * nnz / min(ts) / 1e6, 1)}
    h.close()
capi.set_row_segments(1)
err = float(np.max(np.abs(outs["other"] - outs["segments"]) / np.maximum(l1, 1e-300)))
res["max_err_over_l1"] = err
assert err <= 1e-12, err
print(json.dumps(res))
