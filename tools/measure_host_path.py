#!/usr/bin/env python3
"""PCIe-inclusive timings of the host entry points (what an R caller sees):
one-shot rsp_column_sums_host (pageable host x/p in, host sums out) and the
upload-once handle (rsp_csc_upload, then rsp_csc_column_sums incl. the D2H of the
sums).  These are NOT bench.py's `value` (which is device-resident throughput)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from bench import build_offsets, SEED
from rcppsparse_amd import capi


def main():
    capi.load()
    for wl, nnz_override in (("c2", 0), ("c3", 100_000_000), ("c3", 0)):
        nrow, ncol, nnz, shape, p = build_offsets(wl, nnz_override)
        xt = torch.empty(nnz, dtype=torch.float64, device="cuda")
        capi.gen_values_device(xt, SEED, 0, 0)
        x = xt.cpu().numpy()          # pageable host memory, like an R vector
        del xt
        torch.cuda.empty_cache()
        capi.column_sums_host(x[:1000], np.array([0, 1000], dtype=np.int32))   # warm the runtime
        t0 = time.perf_counter()
        s1 = capi.column_sums_host(x, p)
        t_oneshot = time.perf_counter() - t0
        t0 = time.perf_counter()
        h = capi.DeviceCSC(x, p, (nrow, ncol))
        t_upload = time.perf_counter() - t0
        h.column_sums()
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            s2 = h.column_sums()
            ts.append(time.perf_counter() - t0)
        h.close()
        assert s1.tobytes() == s2.tobytes()
        B = 8 * nnz + 4 * (ncol + 1) + 8 * ncol
        print(json.dumps({
            "workload": f"{wl} nnz={nnz}", "one_shot_s": t_oneshot, "one_shot_nnz_per_s": nnz / t_oneshot,
            "one_shot_GBps": B / t_oneshot / 1e9, "upload_s": t_upload, "upload_GBps": (8 * nnz + 4 * ncol) / t_upload / 1e9,
            "resident_sums_incl_d2h_ms": sorted(ts)[2] * 1e3, "resident_nnz_per_s": nnz / sorted(ts)[2]}), flush=True)
        del x


if __name__ == "__main__":
    main()
