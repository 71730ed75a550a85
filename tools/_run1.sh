timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for t in 0 8; do echo "=== VQ_TILE=$t"; VQ_TILE=$t timeout 300 python tools/bench_conv.py bf16 16 2>&1 | grep -v "^$" | cut -c1-150; done > gpurun_out/conv_ab3.log 2>&1
timeout 600 python bench.py --steps 10 --warmup 3 > gpurun_out/bench_c3_v10.log 2>&1; tail -1 gpurun_out/bench_c3_v10.log | cut -c1-600
