#!/bin/bash
# A/B of the 16-bit block-local rows in the regrouped copy of the row sums (RSP_ROWS16=0 / 1): kernel times of one-shot
# rsp_row_sums_device at BASELINE config 3's shape under rocprofv3 --kernel-trace --stats.  Run ON THE GPU BOX.
cd /tmp && export TMPDIR=/tmp
O=/root/repo/gpurun_out
for v in 0 1; do
  RSP_ROWS16=$v timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ab_rows16_$v -- python3 /root/repo/tools/run_rowsums.py c3 10000000 5 > $O/ab_rows16_$v.log 2>&1
  python3 - <<PY
import csv, glob, json
out = {"RSP_ROWS16": $v}
for f in glob.glob("$O/ab_rows16_$v/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "rows_" in r["Name"]:
            out[r["Name"].split("(")[0].replace("void ", "").replace("rsp::", "")] = round(float(r["AverageNs"]) / 1e3, 1)
print(json.dumps(out))
PY
done
