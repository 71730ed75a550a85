#!/bin/bash
# Round 2, GPU call 18: streaming (nt) output stores in the conv epilogue (default) vs ordinary stores (VQ_TILE = 8195 << 4), streaming residual loads (8196 << 4)
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels.py tests/test_model.py -m gpu -x -q -k "conv or golden or configs0 or deterministic" > gpurun_out/tests_r2t.log 2>&1; tail -2 gpurun_out/tests_r2t.log
( for v in 0 131120 131136; do echo "== VQ_TILE=$v"; VQ_TILE=$v timeout 200 python tools/bench_epi.py 2>&1 | grep -v amdgpu.ids; done ) | tee gpurun_out/epi_nt_r2t.log
for rep in 1 2; do for v in 0 131120 131136; do echo "== VQ_TILE=$v rep $rep"; VQ_TILE=$v timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print(d['value'], d['ms_per_step'], 'conv3x3', r['conv3x3']['frac'], 'igemm', r['frac'], 'wgrad', r['wgrad']['frac'])"; done; done | tee gpurun_out/bench_r2t.log
