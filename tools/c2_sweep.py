#!/usr/bin/env python3
"""C2-sized calls: chunk length sweep of the planned and the general form (bench.py as a child per point).
    python3 tools/c2_sweep.py [workload] [rows ...]"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
wl = sys.argv[1] if len(sys.argv) > 1 else "c2"
rows = [int(a) for a in sys.argv[2:]] or [0, 8, 12, 16, 24, 32]
for planned in (True, False):
    for r in rows:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--workload", wl, "--steps", "200", "--warmup", "20",
               "--no-cpu-baseline", "--latency-calls", "5", "--chunk-rows", str(r)] + (["--planned"] if planned else [])
        out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True).stdout
        d = json.loads([l for l in out.splitlines() if l.startswith("{")][0])
        print(json.dumps({"planned": planned, "chunk_rows": r, "ms_per_step": round(d["ms_per_step"], 5),
                          "kernel_ms": round(d["roofline"]["kernel_ms"], 5), "kernel_ms_min": round(d["roofline"]["kernel_ms_min"], 5),
                          "frac": round(d["roofline"]["frac"], 4), "pipelined_ms": round(d["pipelined"]["ms_per_step"], 5),
                          "snapped": (d["config"].get("planned") or {}).get("snapped")}), flush=True)
