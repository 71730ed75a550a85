timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 600 python bench.py --steps 8 --warmup 3 --no-cpu-baseline > gpurun_out/bench_c3_v16.log 2>&1; tail -1 gpurun_out/bench_c3_v16.log | cut -c1-200
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/prof16 -o b -- python /root/repo/bench.py --steps 4 --warmup 2 --no-cpu-baseline > /root/repo/gpurun_out/prof16.log 2>&1
