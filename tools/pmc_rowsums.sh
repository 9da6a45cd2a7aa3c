#!/bin/bash
# Run ON THE GPU BOX: kernel times and LDS / stall counters of the row-wise path's kernels.
#   bash /root/repo/tools/pmc_rowsums.sh <tag> <workload> <nrow>
set -e -o pipefail
TAG=$1; WL=$2; NROW=$3
cd /tmp && export TMPDIR=/tmp
O=/root/repo/gpurun_out
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/pmcr_${TAG}_stats -- \
    python3 /root/repo/tools/run_rowsums.py $WL $NROW 5 > $O/pmcr_${TAG}_stats.log 2>&1 || echo "stats pass failed"
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES" \
           "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_ATOMIC_RETURN"; do
  g=$(echo $grp | cut -d' ' -f1)
  timeout -k 10 200 rocprofv3 --pmc $grp --output-format csv -d $O/pmcr_${TAG}_$g -- \
    python3 /root/repo/tools/run_rowsums.py $WL $NROW 2 > $O/pmcr_${TAG}_$g.log 2>&1 || echo "pass $g failed"
done
python3 - <<PY
import csv, glob, collections, json
tot = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob("$O/pmcr_${TAG}_SQ*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].split("<")[0].replace("void ", "").replace("rsp::", "")
        if k.startswith("rows_"):
            t = tot[k][r["Counter_Name"]]; t[0] += float(r["Counter_Value"]); t[1] += 1
stats = {}
for f in glob.glob("$O/pmcr_${TAG}_stats/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "rows_" in r["Name"]:
            stats[r["Name"].split("(")[0].replace("void ", "").replace("rsp::", "")] = {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3, "min_us": float(r["MinNs"]) / 1e3, "max_us": float(r["MaxNs"]) / 1e3}
print(json.dumps({"tag": "$TAG", "workload": "$WL", "nrow": $NROW, "kernel_stats": stats,
                  "counters_per_launch": {k: {c: v / n for c, (v, n) in sorted(d.items())} for k, d in tot.items()}}))
PY
