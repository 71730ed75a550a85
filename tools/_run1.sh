timeout 1200 python -m pytest tests/test_kernels.py -m gpu -x -q 2>&1 | tail -2
timeout 300 python tools/bench_conv.py bf16 16 2>&1 | grep "k[13] up" | awk -F'wgrad' '{print substr($1,1,34) " wgrad" $2}'
timeout 600 python bench.py --steps 8 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['wgrad']['achieved'])"
