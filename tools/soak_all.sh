#!/bin/bash
# Run ON THE GPU BOX: every soak of the repository (about 32 minutes in all, so in two calls: part 1, part 2); results under
# gpurun_out/.  A step that fails or times out ends the call: no further GPU step is started after it.
#   bash tools/soak_all.sh 1        bash tools/soak_all.sh 2
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out
step() {   # step <seconds> <log> <command...>
  local t=$1 log=$2; shift 2
  timeout -k 10 $t "$@" > $O/$log 2>&1 || { echo "FAILED: $* (see $log)"; tail -5 $O/$log; exit 1; }
  tail -1 $O/$log
}
if [ "${1:-1}" = "1" ]; then
  step 700 soak_all_fuzz.log python3 $R/tools/soak_fuzz.py --seconds 500 --mode mixed --seed 77
  step 300 soak_all_next.log python3 $R/tools/soak_fuzz.py --seconds 150 --what next --seed 78
  step 300 soak_all_mcsc.log python3 $R/tools/soak_mcsc.py 120 5
  step 200 soak_all_rewrites.log python3 $R/tools/soak_auto_plan.py rewrites 10000
else
  step 400 soak_all_rowsums.log python3 $R/tools/soak_rowsums.py 200
  step 300 soak_all_segments.log python3 $R/tools/soak_row_segments.py 600 9
  step 400 soak_all_slices.log python3 $R/tools/soak_row_slices.py 300 9
  step 300 soak_all_xp.log python3 $R/tools/soak_crossprod_tall.py
  step 300 soak_all_devplan.log python3 $R/tools/soak_device_plan.py 150 11
  step 300 soak_all_autoplan.log python3 $R/tools/soak_auto_plan.py 150 5
  step 300 soak_all_hostpath.log python3 $R/tools/soak_host_path.py 150 5
  step 300 soak_all_autothreads.log python3 $R/tools/soak_auto_plan_threads.py 120 3 5
fi
