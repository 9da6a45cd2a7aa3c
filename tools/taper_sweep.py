#!/usr/bin/env python3
"""Interleaved A/B of chunking settings in ONE process on the same data: rounds x settings,
median / min per setting.  A setting is "permille,tail_rows[,chunk_rows]" (chunk_rows 0 = auto)."""
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
    ap.add_argument("--settings", default="0,0;150,64;150,32;300,64")
    ap.add_argument("--rounds", type=int, default=10)
    ap.add_argument("--per-round", type=int, default=5)
    a = ap.parse_args()
    capi.load()
    settings = [tuple(int(v) for v in s.split(",")) for s in a.settings.split(";")]
    settings = [tuple(list(s) + [0] * (4 - len(s))) for s in settings]   # permille, tail_rows, chunk_rows, experiment
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
        wss = {}
        for s in settings:
            capi.set_taper(s[0], s[1])
            capi.set_tuning(s[2])
            wss[s] = capi.alloc_workspace(ncol, nnz)
        B = 8 * nnz + 4 * (ncol + 1) + 8 * ncol
        times = {s: [] for s in settings}
        call = 0
        for rnd in range(a.rounds + 1):
            for s in settings:
                capi.set_taper(s[0], s[1])
                capi.set_tuning(s[2])
                capi.set_experiment(s[3])
                evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                       for _ in range(a.per_round)]
                for ea, eb in evs:
                    ea.record()
                    capi.column_sums_device(xs[call % ncopies], pt, out, wss[s])
                    eb.record()
                    call += 1
                torch.cuda.synchronize()
                if rnd > 0:   # round 0 is warm-up
                    times[s] += [ea.elapsed_time(eb) for ea, eb in evs]
        capi.set_taper(-1, -1)
        capi.set_tuning(0)
        capi.set_experiment(0)
        for s in settings:
            t = sorted(times[s])
            med = t[len(t) // 2]
            print(json.dumps({"workload": wl, "taper_permille": s[0], "tail_rows": s[1], "chunk_rows": s[2], "experiment": s[3],
                              "n": len(t), "ms_median": round(med, 5), "ms_min": round(t[0], 5),
                              "ms_p90": round(t[int(len(t) * 0.9)], 5), "GBps_median": round(B / med / 1e6, 1),
                              "frac_of_8TBps": round(B / med / 8e9, 4)}), flush=True)
        del xs
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
