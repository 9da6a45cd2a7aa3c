#!/bin/bash
# Run ON THE GPU BOX: the bench lines (no profiler attached) and the rocprofv3 evidence of a round's headline
# workloads, raw under gpurun_out/; tools/summarize_profiles.py + tools/doc_numbers.py turn them into profiles/.
#   bash tools/collect_round.sh
set -o pipefail
R=/root/repo
O=$R/gpurun_out
rm -f $O/bench_lines.jsonl
# (the c3 line is the default command: with its read ceiling, its in-run traffic passes and the also-records)
timeout -k 10 400 python3 $R/bench.py --steps 20 --warmup 5 >> $O/bench_lines.jsonl 2>> $O/bench_lines.err || { echo "bench c3 failed: stopping (no further GPU step after a failed one)"; exit 1; }
for wl in c5 c5desc c4shard vignette; do
  timeout -k 10 400 python3 $R/bench.py --workload $wl --steps 20 --warmup 5 --no-also >> $O/bench_lines.jsonl 2>> $O/bench_lines.err || { echo "bench $wl failed: stopping (no further GPU step after a failed one)"; exit 1; }
done
timeout -k 10 200 python3 $R/bench.py --workload c2 --steps 300 --warmup 30 --no-cpu-baseline --no-also >> $O/bench_lines.jsonl 2>> $O/bench_lines.err || exit 1
# (the plan-free entry plans for itself since round 5; the general kernels alone: RSP_AUTO_PLAN=0)
RSP_AUTO_PLAN=0 timeout -k 10 200 python3 $R/bench.py --workload c2 --steps 300 --warmup 30 --no-cpu-baseline --no-also >> $O/bench_lines.jsonl 2>> $O/bench_lines.err || exit 1
timeout -k 10 200 python3 $R/bench.py --workload c2 --steps 300 --warmup 30 --no-cpu-baseline --no-also --planned >> $O/bench_lines.jsonl 2>> $O/bench_lines.err || exit 1
timeout -k 10 200 python3 $R/bench.py --workload c2 --steps 300 --warmup 30 --no-cpu-baseline --no-also --planned --no-lean >> $O/bench_lines.jsonl 2>> $O/bench_lines.err || exit 1
timeout -k 10 200 python3 $R/bench.py --workload vignette --steps 300 --warmup 30 --no-cpu-baseline --no-also --planned >> $O/bench_lines.jsonl 2>> $O/bench_lines.err || exit 1
echo "bench lines: $(wc -l < $O/bench_lines.jsonl)"
timeout -k 10 500 bash $R/tools/profile_gpu.sh c3 --workload c3 || { echo "profile c3 failed: stopping (no further GPU step after a failed one)"; exit 1; }
timeout -k 10 500 bash $R/tools/profile_gpu.sh c5 --workload c5 || { echo "profile c5 failed: stopping (no further GPU step after a failed one)"; exit 1; }
timeout -k 10 300 bash $R/tools/profile_gpu.sh c4shard --workload c4shard || { echo "profile c4shard failed: stopping (no further GPU step after a failed one)"; exit 1; }
timeout -k 10 300 bash $R/tools/profile_gpu.sh c2 --workload c2 --steps 200 || { echo "profile c2 failed: stopping (no further GPU step after a failed one)"; exit 1; }
RSP_AUTO_PLAN=0 timeout -k 10 300 bash $R/tools/profile_gpu.sh c2general --workload c2 --steps 200 || { echo "profile c2general failed: stopping (no further GPU step after a failed one)"; exit 1; }
timeout -k 10 300 bash $R/tools/profile_gpu.sh vignetteplanned --workload vignette --planned --steps 200 || { echo "profile vignetteplanned failed: stopping"; exit 1; }
timeout -k 10 300 bash $R/tools/profile_gpu.sh c2planned --workload c2 --planned --steps 200 || { echo "profile c2planned failed: stopping (no further GPU step after a failed one)"; exit 1; }
