#!/bin/bash
# Run ON THE GPU BOX: every soak of the repository, one after the other (about 20 minutes); results under gpurun_out/.
R=/root/repo; O=$R/gpurun_out
timeout -k 10 700 python3 $R/tools/soak_fuzz.py --seconds 500 --mode mixed --seed 77 > $O/soak_all_fuzz.log 2>&1 && tail -1 $O/soak_all_fuzz.log
timeout -k 10 300 python3 $R/tools/soak_fuzz.py --seconds 150 --what next --seed 78 > $O/soak_all_next.log 2>&1 && tail -1 $O/soak_all_next.log
timeout -k 10 400 python3 $R/tools/soak_rowsums.py 200 > $O/soak_all_rowsums.log 2>&1 && tail -1 $O/soak_all_rowsums.log
timeout -k 10 300 python3 $R/tools/soak_row_segments.py 600 9 > $O/soak_all_segments.log 2>&1 && tail -1 $O/soak_all_segments.log
timeout -k 10 400 python3 $R/tools/soak_row_slices.py 300 9 > $O/soak_all_slices.log 2>&1 && tail -1 $O/soak_all_slices.log
timeout -k 10 300 python3 $R/tools/soak_crossprod_tall.py > $O/soak_all_xp.log 2>&1 && tail -1 $O/soak_all_xp.log
