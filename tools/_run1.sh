timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
timeout 600 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --conv-table gpurun_out/conv_table_c3.txt > gpurun_out/bench_c3_v21.log 2>&1; tail -1 gpurun_out/bench_c3_v21.log | cut -c1-200
