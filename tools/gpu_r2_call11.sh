#!/bin/bash
# Round 2, GPU call 11: the patch-staged 256 x 256 tile (conv_igemm_p9_kernel) vs the one-tap tile (VQ_TILE = 512 << 4): tests, micro, bench
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels.py tests/test_model.py -m gpu -x -q -k "patch_staged or tile_modes or groupnorm_statistics or conv_layers_at_full_size or adjoint or golden or configs0" > gpurun_out/tests_r2l.log 2>&1; tail -3 gpurun_out/tests_r2l.log
( for rep in 1 2; do for v in 0 8192; do echo "== VQ_TILE=$v rep $rep"; VQ_ITERS=30 VQ_TILE=$v timeout 100 python tools/bench_conv.py bf16 16 1,2,6,7 2>&1 | grep -v amdgpu.ids; done; done
  for v in 0 8192; do echo "== fp16 VQ_TILE=$v"; VQ_ITERS=30 VQ_TILE=$v timeout 100 python tools/bench_conv.py fp16 16 1,2 2>&1 | grep -v amdgpu.ids; done ) | tee gpurun_out/p9_micro_r2l.log
for rep in 1 2; do for v in 0 8192; do echo "== VQ_TILE=$v rep $rep"; VQ_TILE=$v timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print(d['value'], d['ms_per_step'], 'conv3x3', r['conv3x3']['frac'], 'igemm', r['frac'], 'wgrad', r['wgrad']['frac'])"; done; done | tee gpurun_out/bench_r2l.log
