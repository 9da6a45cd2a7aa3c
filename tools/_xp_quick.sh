cd /root/repo
timeout -k 10 400 python -m pytest tests/test_gpu_crossprod.py -x -q -m gpu 2>&1 | tail -3 || exit 1
timeout -k 10 300 python3 tools/compare_crossprod_panels.py gpurun_out/xp_panels.json || exit 1
cd /tmp && export TMPDIR=/tmp
rm -rf /root/repo/gpurun_out/xp_prof
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/xp_prof -o xp -- python3 /root/repo/tools/run_crossprod_tall.py 256 6 > /root/repo/gpurun_out/xp_prof.log 2>&1 || exit 1
