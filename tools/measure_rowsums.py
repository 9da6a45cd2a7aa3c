#!/usr/bin/env python3
"""Timing of the row-wise "next" entries on one GPU: rsp_row_sums_device (sort + reduce every
call) and the cached row-major form (what rsp_csc_row_sums costs after its first call), plus
the oracle's rowSums loop (reference RcppSparse.h:138-144, 1 thread) on a bounded sample."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import oracle
from bench import build_offsets, SEED
from rcppsparse_amd import capi


def ev_time(fn, reps):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[len(ts) // 2], ts[0]


def main():
    capi.load()
    L = capi.load()
    for wl in sys.argv[1:] or ["c2", "c3"]:
        nrow, ncol, nnz, shape, p = build_offsets(wl, 0)
        pt = torch.from_numpy(p).cuda()
        xt = torch.empty(nnz, dtype=torch.float64, device="cuda")
        it = torch.empty(nnz, dtype=torch.int32, device="cuda")
        capi.gen_values_device(xt, SEED, 0, 0)
        capi.gen_row_indices_device(it, pt, nrow, SEED)
        out = torch.empty(nrow, dtype=torch.float64, device="cuda")
        nbytes = int(L.rsp_row_sums_workspace_bytes(nrow, nnz))
        ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
        med, best = ev_time(lambda: capi.row_sums_device(xt, it, nrow, out, ws), 5)
        B = 12 * nnz + 8 * nrow
        row = {"workload": wl, "nnz": nnz, "nrow": nrow, "workspace_GB": nbytes / 1e9,
               "row_sums_device_ms": med, "row_sums_device_nnz_per_s": nnz / med * 1e3,
               "algorithmic_GBps": B / med / 1e6}
        # repeated sums on a resident matrix: the row-wise form is already in the workspace, only the
        # last kernel runs.  Timed here as "second call with the same workspace minus the build": the C ABI
        # keeps that form behind the handle (rsp_csc_row_sums), whose call also copies nrow doubles back.
        if nnz <= 200_000_000:
            h = capi.DeviceCSC(xt.cpu().numpy(), p, (nrow, ncol), i=it.cpu().numpy())
            h.row_sums()
            t0 = time.perf_counter()
            for _ in range(5):
                h.row_sums()
            row["handle_repeat_ms_incl_D2H"] = (time.perf_counter() - t0) / 5 * 1e3
            h.close()
        # CPU: oracle rowSums on the first columns holding ~5e7 nnz
        ncs = max(1, int(np.searchsorted(p, 50_000_000, side="right")) - 1)
        nz = int(p[ncs])
        xs = oracle.gen_values(nz, SEED, 0, 0)
        is_ = oracle.gen_row_indices(p, nrow, SEED, 0, ncs)
        ps = np.ascontiguousarray(p[:ncs + 1])
        oracle.row_sums(xs, is_, ps, nrow)
        t0 = time.perf_counter()
        ref = oracle.row_sums(xs, is_, ps, nrow)
        row["cpu_oracle_nnz_per_s"] = nz / (time.perf_counter() - t0)
        print(json.dumps(row), flush=True)
        del xt, it, ws
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
