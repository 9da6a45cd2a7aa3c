#!/bin/bash
# A/B of the folded fix-up (one launch) against the two-launch form on BASELINE config 2 through the general kernels
# (RSP_AUTO_PLAN=0: what a key's first two calls and every one-shot host call run).  usage: tools/ab_c2_fold.sh [workload]
W=${1:-c2}
for fold in 0 1 0 1; do
  RSP_AUTO_PLAN=0 RSP_FOLD_FIXUP=$fold python bench.py --workload $W --steps 200 --warmup 20 --no-cpu-baseline --no-also --traffic-pass off --ceiling-reps 0 2>/dev/null \
    | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=l['roofline']; print(json.dumps({'workload':'$W','fold_fixup':$fold,'ms_per_call':l['ms_per_step'],'kernel_ms':r['kernel_ms'],'frac':r['frac'],'regions_ms':l['config']['regions_ms'],'parity_err':l['parity']['max_abs_err_over_l1']}))"
done
