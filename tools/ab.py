#!/usr/bin/env python3
"""Interleaved A/B of kernel variants in ONE process on the same data
(cdna_hip_programming.md rule 24): rounds x variants, median / min per variant."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from bench import build_offsets, SEED
from rcppsparse_amd import capi, _build

if os.environ.get("RSP_AB_LIB"):        # time another build of the library (e.g. a saved baseline .so)
    _build.LIB_PATH = os.path.abspath(os.environ["RSP_AB_LIB"])
    _build.build_library = lambda *a, **k: _build.LIB_PATH


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workloads", default="c3")
    ap.add_argument("--variants", default="0,1")
    ap.add_argument("--chunk-rows", type=int, default=0)
    ap.add_argument("--rounds", type=int, default=12)
    ap.add_argument("--per-round", type=int, default=5)
    a = ap.parse_args()
    capi.load()
    variants = [int(v) for v in a.variants.split(",")]
    for wl in a.workloads.split(","):
        nrow, ncol, nnz, shape, p = build_offsets(wl, 0)
        pt = torch.from_numpy(p).cuda()
        ncopies = max(1, int(np.ceil(400e6 / (8.0 * nnz))))
        xs = []
        for k in range(ncopies):
            x = torch.empty(nnz, dtype=torch.float64, device="cuda")
            capi.gen_values_device(x, SEED + k, 0, 0)
            xs.append(x)
        out = torch.empty(ncol, dtype=torch.float64, device="cuda")
        capi.set_tuning(a.chunk_rows)
        ws = capi.alloc_workspace(ncol, nnz)
        B = 8 * nnz + 4 * (ncol + 1) + 8 * ncol
        times = {v: [] for v in variants}
        call = 0
        for rnd in range(a.rounds + 1):
            for v in variants:
                capi.set_experiment(v)
                evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                       for _ in range(a.per_round)]
                for ea, eb in evs:
                    ea.record()
                    capi.column_sums_device(xs[call % ncopies], pt, out, ws)
                    eb.record()
                    call += 1
                torch.cuda.synchronize()
                if rnd > 0:   # round 0 is warm-up
                    times[v] += [ea.elapsed_time(eb) for ea, eb in evs]
        capi.set_experiment(0)
        for v in variants:
            t = sorted(times[v])
            med = t[len(t) // 2]
            print(json.dumps({"workload": wl, "variant": v, "n": len(t), "ms_median": med, "ms_min": t[0],
                              "ms_p90": t[int(len(t) * 0.9)], "GBps_median": B / med / 1e6,
                              "frac_of_8TBps": B / med / 8e9}), flush=True)
        del xs
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
