#!/bin/bash
# Run ON THE GPU BOX: instruction mix / stall counters of the column-sum kernel for one workload.
#   bash /root/repo/tools/pmc_mix.sh <workload>
# One rocprofv3 --pmc pass per counter group (never combined with tracing options).
set -e -o pipefail
WL=$1
cd /tmp && export TMPDIR=/tmp
O=/root/repo/gpurun_out
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR" "SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE"; do
  tag=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --pmc $grp --output-format csv -d $O/pmc_${WL}_$tag -- \
    python3 /root/repo/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pipelined --latency-calls 1 --workload $WL > $O/pmc_${WL}_$tag.log 2>&1
done
python3 - <<PY
import csv, glob, collections
tot = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob("$O/pmc_${WL}_*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "colsums_chunks_kernel" in r["Kernel_Name"]:
            t = tot[r["Counter_Name"]]; t[0] += float(r["Counter_Value"]); t[1] += 1
for k, (v, n) in sorted(tot.items()):
    print(f"{k:28s} {v / n:16.0f} per launch ({n} launches)")
PY
