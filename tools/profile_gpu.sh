#!/bin/bash
# Run ON THE GPU BOX (through gpurun) from any directory:
#   bash /root/repo/tools/profile_gpu.sh <tag> [bench args...]
# Writes raw rocprofv3 output under /root/repo/gpurun_out/prof_<tag>_{stats,fetch,write};
# tools/summarize_profiles.py then turns those into the committed files in profiles/.
# Counters are collected in their own passes (never together with --stats), and the
# profiled program is python3 itself (no env/bash hop after `--`).
set -e -o pipefail
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
O=/root/repo/gpurun_out
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${TAG}_stats -- \
    python3 /root/repo/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-pipelined --latency-calls 1 --no-also --traffic-pass off "$@" > $O/prof_${TAG}_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/prof_${TAG}_fetch -- \
    python3 /root/repo/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pipelined --latency-calls 1 --no-also --traffic-pass off "$@" > $O/prof_${TAG}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/prof_${TAG}_write -- \
    python3 /root/repo/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pipelined --latency-calls 1 --no-also --traffic-pass off "$@" > $O/prof_${TAG}_write.log 2>&1
grep '"metric"' $O/prof_${TAG}_stats.log | cut -c1-300
