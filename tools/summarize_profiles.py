#!/usr/bin/env python3
"""gpurun_out/prof_<tag>_{stats,fetch,write} -> profiles/<round>_<tag>_*.{csv,json}

    python tools/summarize_profiles.py r01 c3 --workload c3

HBM traffic follows MI355X_MICROARCH.md section HBM: FETCH_SIZE / WRITE_SIZE are in
KiB; on gfx950 FETCH_SIZE reports exactly half of the bytes of a wide (16 B/lane)
coalesced streaming read, so the read side is doubled; WRITE_SIZE is exact
(cross-checked here on gen_values_kernel, a pure 8 B/element streaming store).
"""
import argparse
import collections
import csv
import glob
import json
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def newest(paths):
    """gpurun merges a call's output into gpurun_out/ next to what earlier calls left there: only the newest
    run's files of a directory count."""
    return max(paths, key=os.path.getmtime) if paths else None


def counters(path):
    d = collections.defaultdict(list)
    for f in [newest(glob.glob(os.path.join(path, "**", "*_counter_collection.csv"), recursive=True))]:
        if f is None:
            continue
        for r in csv.DictReader(open(f)):
            d[(r["Kernel_Name"].split("(")[0].replace("void ", ""), r["Counter_Name"])].append(
                float(r["Counter_Value"]))
    return d


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("round")
    ap.add_argument("tag")
    ap.add_argument("--workload", required=True)
    a = ap.parse_args()
    src = os.path.join(ROOT, "gpurun_out")
    dst = os.path.join(ROOT, "profiles")
    os.makedirs(dst, exist_ok=True)
    base = f"{a.round}_{a.tag}"
    stats = glob.glob(os.path.join(src, f"prof_{a.tag}_stats", "**", "*_kernel_stats.csv"), recursive=True)
    shutil.copy(newest(stats), os.path.join(dst, base + "_kernel_stats.csv"))
    log = os.path.join(src, f"prof_{a.tag}_stats.log")
    bench_line = [l for l in open(log) if l.startswith('{"metric"')]
    if bench_line:
        open(os.path.join(dst, base + "_bench_under_rocprof.json"), "w").write(bench_line[-1])
    fetch = counters(os.path.join(src, f"prof_{a.tag}_fetch"))
    write = counters(os.path.join(src, f"prof_{a.tag}_write"))
    rows, main_kernel, main_launches = [], None, 0
    for (k, c), v in sorted({**fetch, **write}.items()):
        rows.append({"kernel": k, "counter": c, "launches": len(v), "mean_KiB": sum(v) / len(v),
                     "min_KiB": min(v), "max_KiB": max(v)})
        # the kernel of the profiled calls: the one launched most often (round 5: the plan-free entry plans for itself, so a
        # command's first calls run colsums_chunks_kernel and the rest the lean / columns kernel it settles on)
        if ("colsums_chunks_kernel" in k or "colsums_lean_kernel" in k or "colsums_columns_kernel" in k) and len(v) > main_launches:
            main_kernel, main_launches = k, len(v)
    with open(os.path.join(dst, base + "_pmc_summary.csv"), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0]))
        w.writeheader()
        w.writerows(rows)
    mean = lambda d, k, c: sum(d[(k, c)]) / len(d[(k, c)])
    fx = [k for (k, c) in fetch if "fixup" in k] if "colsums_chunks_kernel" in main_kernel else []
    rd = 2 * 1024 * mean(fetch, main_kernel, "FETCH_SIZE")
    wr = 1024 * mean(write, main_kernel, "WRITE_SIZE")
    if fx:
        rd += 2 * 1024 * mean(fetch, fx[0], "FETCH_SIZE")
        wr += 1024 * mean(write, fx[0], "WRITE_SIZE")
    out = {
        "workload": a.workload, "kernel": main_kernel, "hbm_bytes_per_launch": rd + wr,
        "read_bytes": rd, "write_bytes": wr,
        "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes; KiB -> bytes; "
                  "FETCH_SIZE doubled (gfx950 counts 128-B requests of a 16 B/lane stream as 64 B); "
                  "main kernel (+ fix-up kernel, when the form has one) of one columnSums call",
    }
    gen = [k for (k, c) in write if "gen_values" in k]
    if gen:
        out["write_calibration_gen_values_bytes"] = 1024 * mean(write, gen[0], "WRITE_SIZE")
    json.dump(out, open(os.path.join(dst, base + "_traffic.json"), "w"), indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
