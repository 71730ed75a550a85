for t in 0 5; do echo "=== VQ_TILE=$t"; VQ_TILE=$t timeout 300 python tools/bench_conv.py bf16 16 2>&1 | grep -v "^$" | cut -c1-150; done > gpurun_out/t256tap3.log 2>&1
