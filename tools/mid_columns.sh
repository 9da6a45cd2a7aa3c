#!/bin/bash
# Run ON THE GPU BOX: column-length regimes at 1e9 entries through the plan-free entry with and without its own planning,
# and through the caller's plan (form, kernels per call in ms, fraction of 8 TB/s, parity).
run() {
python bench.py --workload $1 $2 --steps 20 --warmup 3 --no-cpu-baseline --no-also --no-pipelined --latency-calls 1 --traffic-pass off --ceiling-reps 0 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('$1', '$2', 'auto_plan', d['config']['auto_plan'], d['config']['shards'][0]['form'], (d['config'].get('planned') or {}).get('form'), round(d['roofline']['kernel_ms'],4), round(d['roofline']['frac'],3), d['parity']['max_abs_err_over_l1'])" || exit 1
}
for wl in m10 m30 m100 m300; do
  RSP_AUTO_PLAN=0 run $wl "" || exit 1
  run $wl "" || exit 1
  run $wl "--planned" || exit 1
done
