timeout 1200 python -m pytest tests/test_kernels.py -m gpu -x -q 2>&1 | tail -2
timeout 300 python tools/bench_gn.py 16 2>&1 | tail -6
timeout 600 python bench.py --steps 8 --warmup 3 --no-cpu-baseline > gpurun_out/bench_c3_v20.log 2>&1; tail -1 gpurun_out/bench_c3_v20.log | cut -c1-200
