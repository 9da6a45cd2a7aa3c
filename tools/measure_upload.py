#!/usr/bin/env python3
"""Upload path of the resident handle (rsp_csc_upload): pageable host memory (what R vectors are)
against page-locked memory, and what locking costs.  Decides whether the library should stage
uploads through pinned buffers."""
import ctypes
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from bench import build_offsets, SEED
from rcppsparse_amd import capi
import oracle


def upload_ms(x, p, dim, reps=3):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        h = capi.DeviceCSC(x, p, dim)
        ts.append((time.perf_counter() - t0) * 1e3)
        h.close()
    return min(ts)


def main():
    capi.load()
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipHostRegister.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint]
    hip.hipHostUnregister.argtypes = [ctypes.c_void_p]
    for wl, nnz_override in (("c2", 0), ("c3", 100_000_000), ("c3", 0)):
        nrow, ncol, nnz, shape, p = build_offsets(wl, nnz_override)
        nth = max(1, len(os.sched_getaffinity(0)))
        x = oracle.gen_values_threads(nnz, SEED, 0, 0, min(nth, 16))        # pageable, like an R vector
        nbytes = x.nbytes + p.nbytes
        row = {"workload": wl, "nnz": nnz, "bytes": nbytes}
        ms = upload_ms(x, p, (nrow, ncol))
        row["pageable_ms"] = ms
        row["pageable_GBps"] = nbytes / ms / 1e6
        # lock the same pages (hipHostRegister), upload, unlock
        t0 = time.perf_counter()
        rc = hip.hipHostRegister(ctypes.c_void_p(x.ctypes.data), ctypes.c_size_t(x.nbytes), 0)
        row["hipHostRegister_ms"] = (time.perf_counter() - t0) * 1e3
        if rc == 0:
            ms = upload_ms(x, p, (nrow, ncol))
            row["registered_ms"] = ms
            row["registered_GBps"] = nbytes / ms / 1e6
            t0 = time.perf_counter()
            hip.hipHostUnregister(ctypes.c_void_p(x.ctypes.data))
            row["hipHostUnregister_ms"] = (time.perf_counter() - t0) * 1e3
            row["one_shot_with_locking_ms"] = row["hipHostRegister_ms"] + row["registered_ms"] + row["hipHostUnregister_ms"]
        else:
            row["hipHostRegister_error"] = rc
        # copy into a pinned staging buffer first (what chunked pinned staging would do), then upload
        if x.nbytes <= 2_000_000_000:
            pin = torch.empty(nnz, dtype=torch.float64).pin_memory()
            t0 = time.perf_counter()
            pin.numpy()[:] = x
            row["host_copy_into_pinned_ms"] = (time.perf_counter() - t0) * 1e3
            row["host_copy_GBps"] = x.nbytes / row["host_copy_into_pinned_ms"] / 1e6
            del pin
        print(json.dumps(row), flush=True)
        del x


if __name__ == "__main__":
    main()
