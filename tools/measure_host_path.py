#!/usr/bin/env python3
"""PCIe-inclusive timings of the host entry points (what an R caller sees) -- NOT bench.py's `value`
(device-resident throughput).

Part 1 (round 5, VERDICT round 4 missing 2): the one-shot drop-in on SMALL inputs.  For the reference's own example
shapes (README.md:33-38: 10 x 10; src/example.cpp:10: 10 x 5) and uniform matrices of 1e3 ... 1e7 stored entries:
  one_shot_ms   rsp_column_sums_host (pageable x / p in, sums out), median of the calls after the first
  host_loop_ms  the loop the Rcpp layer runs below the offload threshold (columnsums_impl.hpp: the reference's
                double loop over this package's InnerIterator, 1 thread), same matrix, same box
and the crossover between the two = the default of RcppSparse.min_nnz / RCPPSPARSE_MIN_NNZ.
Part 2: upload-once handle against one-shot at C2 / 1e8 / C3 sizes (as in earlier rounds).

    python tools/measure_host_path.py [--small-only] > profiles/r05_one_shot.json"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from bench import build_offsets, SEED
from rcppsparse_amd import capi, hostseam, synth


def med(fn, reps):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    return ts[len(ts) // 2] * 1e3, ts[0] * 1e3


def small_part():
    rows = []
    shapes = [("c1 10x10 d0.1 (README.md:33-38)", synth.rsparsematrix(10, 10, density=0.1, seed=1)),
              ("man 10x5 d0.5 (example.cpp:10)", synth.rsparsematrix(10, 5, density=0.5, seed=2))]
    for nnz in (1_000, 10_000, 100_000, 300_000, 1_000_000, 3_000_000, 10_000_000):
        ncol = max(1, nnz // 10)
        p = synth.offsets_from_counts(synth.uniform_counts(ncol, nnz, SEED, None))
        shapes.append((f"uniform nnz={nnz} ncol={ncol}", {"x": synth.gen_values(nnz, SEED), "p": p,
                                                         "i": np.zeros(nnz, dtype=np.int32), "Dim": np.array([ncol, ncol], dtype=np.int32)}))
    for name, m in shapes:
        x, p = np.ascontiguousarray(m["x"]), np.ascontiguousarray(m["p"], dtype=np.int32)
        nnz, ncol = int(x.size), len(p) - 1
        reps = 200 if nnz <= 100_000 else (50 if nnz <= 1_000_000 else 15)
        first_t0 = time.perf_counter()
        got = capi.column_sums_host(x, p)                    # the first call at this size: grows the library's buffers
        first_ms = (time.perf_counter() - first_t0) * 1e3
        one_ms, one_min = med(lambda: capi.column_sums_host(x, p), reps)
        loop = hostseam.columnSums_opt2(m, require_gpu=0, min_nnz=2**40)       # the host loop, whatever the size
        assert hostseam.backend(last=True) == "cpu"
        loop_ms, loop_min = med(lambda: hostseam.columnSums_opt2(m, require_gpu=0, min_nnz=2**40), reps)
        scale = np.add.reduceat(np.abs(x), p[:-1][np.diff(p) > 0]) if nnz else np.zeros(0)
        err = np.abs(got - loop)[np.diff(p) > 0]
        assert np.all(err <= 1e-12 * scale)
        rows.append({"shape": name, "nnz": nnz, "ncol": ncol, "one_shot_ms": one_ms, "one_shot_min_ms": one_min,
                     "one_shot_first_call_ms": first_ms, "host_loop_ms": loop_ms, "host_loop_min_ms": loop_min,
                     "one_shot_over_loop": one_ms / loop_ms})
        print(json.dumps(rows[-1]), file=sys.stderr, flush=True)
    # crossover: where one_shot_ms == host_loop_ms, interpolated in log(nnz) between the two uniform sizes around it
    uni = [r for r in rows if r["shape"].startswith("uniform")]
    cross = None
    for a, b in zip(uni, uni[1:]):
        fa, fb = np.log(a["one_shot_over_loop"]), np.log(b["one_shot_over_loop"])
        if fa > 0 >= fb:
            t = fa / (fa - fb)
            cross = float(np.exp(np.log(a["nnz"]) + t * (np.log(b["nnz"]) - np.log(a["nnz"]))))
    return rows, cross


def big_part():
    rows = []
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
        t_first = time.perf_counter() - t0
        t0 = time.perf_counter()
        s1 = capi.column_sums_host(x, p)
        t_oneshot = time.perf_counter() - t0
        capi.release_cached()
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
        assert np.allclose(s1, s2, rtol=0, atol=1e-9)      # (the handle runs its planned form: same sums, not always the same bits)
        B = 8 * nnz + 4 * (ncol + 1) + 8 * ncol
        rows.append({
            "workload": f"{wl} nnz={nnz}", "one_shot_s": t_oneshot, "one_shot_first_call_s": t_first,
            "one_shot_nnz_per_s": nnz / t_oneshot,
            "one_shot_GBps": B / t_oneshot / 1e9, "upload_s": t_upload, "upload_GBps": (8 * nnz + 4 * ncol) / t_upload / 1e9,
            "resident_sums_incl_d2h_ms": sorted(ts)[2] * 1e3, "resident_nnz_per_s": nnz / sorted(ts)[2],
            "one_shot_minus_upload_minus_resident_ms": (t_oneshot - t_upload) * 1e3 - sorted(ts)[2] * 1e3})
        print(json.dumps(rows[-1]), file=sys.stderr, flush=True)
        del x
    return rows


def main():
    os.environ.pop("RCPPSPARSE_REQUIRE_GPU", None)
    capi.load()
    capi.column_sums_host(np.ones(8), np.array([0, 8], dtype=np.int32))     # the runtime's own first-call costs
    small, cross = small_part()
    out = {"what": "one-shot columnSums through the device (rsp_column_sums_host, arena kept between calls) against the Rcpp "
                   "layer's host loop on the same box; ms, median",
           "host_cpus": os.cpu_count(), "small": small, "crossover_nnz": cross,
           "default_min_nnz": hostseam.min_nnz()}
    if "--small-only" not in sys.argv:
        out["large"] = big_part()
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
