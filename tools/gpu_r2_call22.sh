#!/bin/bash
# Round 2, GPU call 22: nine-tap kernel at three blocks per CU (VQ_TILE = 16384 << 4) vs two
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_kernels.py -m gpu -x -q -k "nine_tap" > gpurun_out/tests_r2x.log 2>&1; tail -2 gpurun_out/tests_r2x.log
( for rep in 1 2; do for v in 0 262144; do echo "== VQ_TILE=$v rep $rep"; VQ_TILE=$v timeout 200 python tools/bench_epi.py 2>&1 | grep -v amdgpu.ids | grep "128->128"; VQ_ITERS=30 VQ_TILE=$v timeout 100 python tools/bench_conv.py bf16 16 0,3,7 2>&1 | grep -v amdgpu.ids | sed 's/| wgrad.*//'; done; done
  for v in 0 262144; do echo "== fp16 VQ_TILE=$v"; VQ_ITERS=30 VQ_TILE=$v timeout 100 python tools/bench_conv.py fp16 16 0,3 2>&1 | grep -v amdgpu.ids | sed 's/| wgrad.*//'; done ) | tee gpurun_out/tap9_dense_r2x.log
for rep in 1 2; do for v in 0 262144; do echo "== VQ_TILE=$v rep $rep"; VQ_TILE=$v timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print(d['value'], d['ms_per_step'], 'conv3x3', r['conv3x3']['frac'], 'igemm', r['frac'], 'wgrad', r['wgrad']['frac'])"; done; done | tee gpurun_out/bench_r2x.log
