#!/usr/bin/env python3
"""Column-length regimes at a given size through the general and the planned form (bench.py as a child per point).
    python3 tools/regime_sweep.py [workload ...]"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for wl in sys.argv[1:] or ["c2", "m10", "m30", "m100"]:
    for planned in (False, True):
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--workload", wl, "--steps", "20" if wl != "c2" else "300",
               "--warmup", "3", "--no-cpu-baseline", "--latency-calls", "3"] + (["--planned"] if planned else [])
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if not lines:
            print(json.dumps({"workload": wl, "planned": planned, "error": r.stderr[-300:]}), flush=True)
            continue
        d = json.loads(lines[0])
        print(json.dumps({"workload": wl, "planned": planned, "form": (d["config"].get("planned") or {}).get("form"),
                          "plan_ms": (d["config"].get("planned") or {}).get("plan_ms"),
                          "ms_per_step": round(d["ms_per_step"], 5), "frac": round(d["roofline"]["frac"], 4),
                          "algorithmic_GB": round(d["roofline"]["algorithmic_bytes_per_launch"] / 1e9, 3),
                          "parity_max_err_over_l1": d["parity"]["max_abs_err_over_l1"]}), flush=True)
