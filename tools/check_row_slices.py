#!/usr/bin/env python3
"""Row-restricted column sums over more than 2^20 rows: the slice-major form against the general form (bitmap
probed in L2) on the same matrix -- agreement relative to the column's sum of |x|, identical bits on a second
run, and the HIP-event time of both.
    python3 tools/check_row_slices.py [workload|nrow,ncol,nnz,structure] [reps]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from bench import build_offsets, SEED
from rcppsparse_amd import capi, synth

spec = sys.argv[1] if len(sys.argv) > 1 else "c3"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
capi.load()
if "," in spec:
    nrow, ncol, nnz, structure = spec.split(",")
    nrow, ncol, nnz = int(float(nrow)), int(float(ncol)), int(float(nnz))
    counts = (synth.uniform_counts(ncol, nnz, SEED, nrow) if structure == "uniform"
              else synth.zipf_counts(ncol, nnz, SEED, nrow))
    p = synth.offsets_from_counts(counts)
else:
    nrow, ncol, nnz, structure, p = build_offsets(spec, 0)
pt = torch.from_numpy(p).cuda()
xt = torch.empty(nnz, dtype=torch.float64, device="cuda")
it = torch.empty(nnz, dtype=torch.int32, device="cuda")
capi.gen_values_device(xt, SEED, 0, 0)
capi.gen_row_indices_device(it, pt, nrow, SEED)
ws = torch.empty(capi.in_rows_workspace_bytes(nrow, ncol, nnz), dtype=torch.uint8, device="cuda")
bits = torch.from_numpy(np.random.default_rng(0).integers(0, 2**32, size=(nrow + 31) // 32, dtype=np.uint32)).cuda()
l1 = capi.column_reduce_device(xt, pt, capi.OP_SUM_ABS)


def timed(on, comp):
    capi.set_row_slices(on)
    out = torch.empty(ncol, dtype=torch.float64, device="cuda")
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        capi.column_sums_in_rows_device(xt, it, pt, nrow, bits, comp, out, ws)
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    again = torch.empty_like(out)
    capi.column_sums_in_rows_device(xt, it, pt, nrow, bits, comp, again, ws)
    assert out.cpu().numpy().tobytes() == again.cpu().numpy().tobytes()
    ts.sort()
    return out, ts[len(ts) // 2]


res = {"spec": spec, "nrow": nrow, "ncol": ncol, "nnz": nnz}
for comp in (False, True):
    g, tg = timed(False, comp)
    s, tsl = timed(True, comp)
    err = float(((g - s).abs() / l1.clamp_min(1e-300)).max().item())
    res["complement" if comp else "in"] = {"general_ms": tg, "slices_ms": tsl, "max_err_over_l1": err,
                                           "slices_GBps": (12 * nnz + 12 * ncol) / tsl / 1e6}
    assert err <= 1e-12, err
print(json.dumps(res))
