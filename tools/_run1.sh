for t in 0 2 1; do echo "=== VQ_TILE=$t"; VQ_TILE=$t timeout 300 python tools/bench_conv.py bf16 16 2>&1 | grep "512-> 512 @ 16\|512-> 512 @  8\|512-> 512 @ 32" | cut -c1-170; done
