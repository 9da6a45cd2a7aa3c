#!/bin/bash
# Run ON THE GPU BOX: the lean planned form against the plan-free kernel on columns of ~10 / ~30 entries from 1e7 to
# 1e9 entries (bench.py lines, no profiler): profiles/rNN_lean_sizes.jsonl.
R=/root/repo; O=$R/gpurun_out; rm -f $O/lean_sizes.jsonl
for spec in "c2 300" "m10_3e7 150" "m10_1e8 60" "m10 12" "m30 12"; do
  set -- $spec
  for mode in "" "--planned"; do
    timeout -k 10 300 python3 $R/bench.py --workload $1 --steps $2 --warmup 5 --no-cpu-baseline --latency-calls 0 --no-pipelined $mode >> $O/lean_sizes.jsonl 2>> $O/lean_sizes.err || echo "bench $1 $mode failed"
  done
done
python3 - <<PY
import json
for l in open("$O/lean_sizes.jsonl"):
    d = json.loads(l)
    pl = d["config"].get("planned")
    print(d["config"]["workload"].split(":")[0], pl["form"] if isinstance(pl, dict) else "general", round(d["ms_per_step"], 4), round(d["roofline"]["frac"], 3), d["parity"]["max_abs_err_over_l1"])
PY
