timeout 900 python -m pytest tests/test_kernels.py -m gpu -x -q 2>&1 | tail -3
timeout 300 python tools/bench_conv.py bf16 16 2>&1 | grep -v "^$" | cut -c1-170 > gpurun_out/conv_v14.log
timeout 600 python bench.py --steps 6 --warmup 3 --no-cpu-baseline --conv-table gpurun_out/conv_table_c3.txt > gpurun_out/bench_c3_v14.log 2>&1; tail -1 gpurun_out/bench_c3_v14.log | cut -c1-200
