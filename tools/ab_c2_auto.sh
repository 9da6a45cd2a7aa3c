#!/bin/bash
# Run ON THE GPU BOX: c2 through the plan-free entry (settles on its own lean plan) and through the caller's plan, three times each.
for k in 1 2 3; do
python bench.py --workload c2 --steps 300 --warmup 30 --no-cpu-baseline --no-also --no-pipelined --latency-calls 1 --traffic-pass off 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('auto', d['config']['shards'][0]['form'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['config']['regions_ms'])" &&
python bench.py --workload c2 --planned --steps 300 --warmup 30 --no-cpu-baseline --no-also --no-pipelined --latency-calls 1 --traffic-pass off 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('planned', d['roofline']['kernel_ms'], d['roofline']['frac'])" || exit 1
done
