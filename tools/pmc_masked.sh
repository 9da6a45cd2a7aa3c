#!/bin/bash
# Run ON THE GPU BOX: cache-path counters of the row-restricted column sums ("next" row f4) in one bitmap regime.
#   bash /root/repo/tools/pmc_masked.sh <tag> <workload> <nrow>
# One rocprofv3 --pmc pass per counter group (never combined with tracing options); the profiled program
# is python3 itself.  Prints per-launch averages for the masked kernel.
# (No TA_* group: a pass with TA_TA_BUSY_sum / TA_ADDR_STALLED_BY_TC_CYCLES_sum / TA_BUFFER_* aborted inside
# rocprofv3 with signal 6 on this pool and sat silent until the box's watchdog killed the call -- round 3.)
set -e -o pipefail
TAG=$1; WL=$2; NROW=$3
cd /tmp && export TMPDIR=/tmp
O=/root/repo/gpurun_out
for grp in "TCP_TCC_READ_REQ_sum TCC_REQ_sum TCC_READ_sum TCC_HIT_sum" \
           "TCC_MISS_sum TCC_READ_SECTORS_sum TCC_EA0_RDREQ_sum TCC_BUSY_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum GRBM_GUI_ACTIVE" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_WAVES"; do
  g=$(echo $grp | cut -d' ' -f1)
  timeout -k 10 200 rocprofv3 --pmc $grp --output-format csv -d $O/pmcm_${TAG}_$g -- \
    python3 /root/repo/tools/run_masked.py $WL $NROW 3 > $O/pmcm_${TAG}_$g.log 2>&1 || echo "pass $g failed"
done
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O/pmcm_${TAG}_trace -- \
  python3 /root/repo/tools/run_masked.py $WL $NROW 5 > $O/pmcm_${TAG}_trace.log 2>&1 || echo "trace pass failed"
python3 - <<PY
import csv, glob, collections, json
tot = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob("$O/pmcm_${TAG}_*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        for name in ("colsums_chunks_kernel", "colsums_rowslices_kernel"):   # (slice form: the general kernel is launched too and returns at once)
            if name in r["Kernel_Name"]:
                t = tot[name][r["Counter_Name"]]; t[0] += float(r["Counter_Value"]); t[1] += 1
out = {name: {k: v / n for k, (v, n) in sorted(d.items())} for name, d in tot.items()}
dur = collections.defaultdict(list)
for f in glob.glob("$O/pmcm_${TAG}_trace/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        dur[r["Kernel_Name"].split("(")[0][:60]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print(json.dumps({"tag": "$TAG", "workload": "$WL", "nrow": $NROW, "per_launch": out,
                  "kernel_us_median": {k: sorted(v)[len(v) // 2] for k, v in dur.items() if "gen_" not in k}}))
PY
