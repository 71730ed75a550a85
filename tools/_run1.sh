timeout 900 python bench.py --workload c5 --steps 4 --warmup 2 --conv-table gpurun_out/conv_table_c5.txt > gpurun_out/bench_c5_v0.log 2>&1; tail -1 gpurun_out/bench_c5_v0.log | cut -c1-400
