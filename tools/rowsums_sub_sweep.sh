#!/bin/bash
# Run ON THE GPU BOX: the one-shot rowSums call on C3 with the partition pass regrouping by coarse blocks of
# 1 / 2 / 4 / 8 row blocks (RSP_ROWS_SUB = 0..3), each under rocprofv3 --kernel-trace --stats; the bench line (with its
# whole-matrix parity) and the per-kernel averages go to gpurun_out/rowsums_sub_<k>.*  (profiles/r04_rowsums.md)
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for k in 0 1 2 3; do
  export RSP_ROWS_SUB=$k
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_rowsums_sub$k -o sub$k -- \
      python3 $R/bench.py --op rowsums --workload c3 --steps 5 --warmup 2 > $O/rowsums_sub_$k.json 2> $O/rowsums_sub_$k.err || echo "sub $k failed"
  echo "== RSP_ROWS_SUB=$k"; cut -c1-160 $O/prof_rowsums_sub$k/sub${k}_kernel_stats.csv | head -8
done
