#!/usr/bin/env python3
"""rowSums of the single-process multi-GPU handle (rsp_mcsc_row_sums): the shards' partial vectors added on the DEVICES
(round 6) against the host-side add (RSP_MCSC_ROWS=host), G shards on this box's one device.  Run once per mode (the
choice is read when the handle first sums its rows):

    python tools/measure_mcsc_rows.py [nrow] [ncol] [nnz] [shards] [reps]
    RSP_MCSC_ROWS=host python tools/measure_mcsc_rows.py ...
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["RCPPSPARSE_REQUIRE_GPU"] = "1"

import numpy as np   # noqa: E402
import torch         # noqa: E402

from rcppsparse_amd import capi, synth   # noqa: E402

nrow = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
ncol = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000
nnz = int(sys.argv[3]) if len(sys.argv) > 3 else 100_000_000
G = int(sys.argv[4]) if len(sys.argv) > 4 else 8
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 5
capi.load()
p = synth.offsets_from_counts(synth.uniform_counts(ncol, nnz, seed=42, nrow=nrow))
pt = torch.from_numpy(p).cuda()
xt = torch.empty(nnz, dtype=torch.float64, device="cuda")
it = torch.empty(nnz, dtype=torch.int32, device="cuda")
capi.gen_values_device(xt, 42, 0, 0)
capi.gen_row_indices_device(it, pt, nrow, 42)
x, i = xt.cpu().numpy(), it.cpu().numpy()
del xt, it
torch.cuda.empty_cache()
h = capi.MultiDeviceCSC(x, p, (nrow, ncol), devices=[0] * G, i=i)
first = h.row_sums()
ts = []
for _ in range(reps):
    t0 = time.perf_counter()
    got = h.row_sums()
    ts.append((time.perf_counter() - t0) * 1e3)
assert got.tobytes() == first.tobytes()
means = h.row_means()
assert means.tobytes() == (first / ncol).tobytes()
h.close()
ref = np.bincount(i, weights=x, minlength=nrow)
scale = np.bincount(i, weights=np.abs(x), minlength=nrow)
assert np.all(np.abs(first - ref) <= 1e-12 * scale)
print(json.dumps({"mode": os.environ.get("RSP_MCSC_ROWS", "devices"), "nrow": nrow, "ncol": ncol, "nnz": nnz, "shards": G,
                  "ms_per_call_median": round(sorted(ts)[len(ts) // 2], 3), "ms_all": [round(t, 3) for t in ts],
                  "max_err_over_l1": float(np.max(np.abs(first - ref) / np.maximum(scale, 1e-300)))}))
