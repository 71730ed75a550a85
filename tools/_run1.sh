timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 300 python tools/bench_attn.py 2>&1 | tail -2
