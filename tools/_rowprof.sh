#!/bin/bash
# usage: _rowprof.sh tag [workloads...]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/rowprof_$tag -- python3 /root/repo/tools/measure_rowsums.py "$@" > /root/repo/gpurun_out/rowprof_$tag.log 2>&1
grep workload /root/repo/gpurun_out/rowprof_$tag.log | cut -c1-170
python3 -c "
import csv,glob
f=glob.glob('/root/repo/gpurun_out/rowprof_$tag/*/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    if 'rows_' in r['Name'] or 'rocprim' in r['Name']: print(r['Name'][:60].ljust(60), r['Calls'], round(float(r['AverageNs'])/1e6,3), round(float(r['MaxNs'])/1e6,3))
"
