#!/bin/bash
# Run ON THE GPU BOX: rocprofv3 --kernel-trace --stats (and the FETCH_SIZE / WRITE_SIZE passes for the bench.py ones) of
# round 3's new kernels: lean form at 1e9 entries, columns form, slice-major row-restricted sums, segments row sums.
# Raw output under gpurun_out/prof_<tag>_*; tools/summarize_profiles.py makes the committed files for the first two.
set -o pipefail
R=/root/repo; O=$R/gpurun_out
timeout -k 10 400 bash $R/tools/profile_gpu.sh m10planned --workload m10 --planned --steps 10 || echo "profile m10planned failed"
timeout -k 10 300 bash $R/tools/profile_gpu.sh vignetteplanned --workload vignette --planned --steps 200 || echo "profile vignetteplanned failed"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_rowslices_stats -- \
    python3 $R/tools/check_row_slices.py c3 5 > $O/prof_rowslices_stats.log 2>&1 || echo "profile rowslices failed"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_rowsegments_stats -- \
    python3 $R/tools/check_row_segments.py 6e4,1e5,5e8 5 > $O/prof_rowsegments_stats.log 2>&1 || echo "profile rowsegments failed"
ls $O | grep -c prof_
