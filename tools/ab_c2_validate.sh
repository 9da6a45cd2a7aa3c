#!/bin/bash
# Run ON THE GPU BOX: c2 through the plan-free entry with a saved build of the library (RSP_AB_LIB_PREV) and with the
# tree's, alternating, four times each; and once through the caller's plan.   bash tools/ab_c2_validate.sh <prev.so>
prev=$1
one() { python bench.py --workload c2 --steps 300 --warmup 30 --no-cpu-baseline --no-also --no-pipelined --latency-calls 1 --traffic-pass off $2 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('$1', d['config']['shards'][0]['form'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['config']['regions_ms'])"; }
for k in 1 2 3 4; do
  RSP_AB_LIB=$prev one prev "" && one new "" || exit 1
done
one planned --planned
