#!/bin/bash
# Run ON THE GPU BOX: where the tall crossprod kernel's cycles go (rocprofv3 --pmc, one pass per group).
#   bash /root/repo/tools/pmc_crossprod.sh <ncol>
set -e -o pipefail
NC=$1
cd /tmp && export TMPDIR=/tmp
O=${GRAFT_REPO_ROOT:-/root/repo}/gpurun_out
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" \
           "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_WAVES"; do
  g=$(echo $grp | cut -d' ' -f1)
  timeout -k 10 200 rocprofv3 --pmc $grp --output-format csv -d $O/pmcx_${NC}_$g -- \
    python3 ${GRAFT_REPO_ROOT:-/root/repo}/tools/run_crossprod_tall.py $NC 3 > $O/pmcx_${NC}_$g.log 2>&1 || echo "pass $g failed"
done
python3 - <<PY
import csv, glob, collections, json
tot = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob("$O/pmcx_${NC}_*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "crossprod_tall_kernel" in r["Kernel_Name"] or "crossprod_panels_kernel" in r["Kernel_Name"]:
            t = tot[r["Counter_Name"]]; t[0] += float(r["Counter_Value"]); t[1] += 1
print(json.dumps({"ncol": $NC, "per_launch": {k: v / n for k, (v, n) in sorted(tot.items())}}))
PY
