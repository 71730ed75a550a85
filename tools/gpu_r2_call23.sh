#!/bin/bash
# Round 2, GPU call 23: three-tap weight-gradient kernel with tile-owning XCDs (any split count) vs split-owning XCDs (VQ_WGTILE=32)
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_kernels.py -m gpu -x -q -k "tile_owning or wgrad or split_reduction or adjoint" > gpurun_out/tests_r2y.log 2>&1; tail -2 gpurun_out/tests_r2y.log
( for rep in 1 2; do for v in 0 512 32; do echo "== VQ_WGTILE=$v rep $rep"; VQ_ITERS=30 VQ_WGTILE=$v timeout 100 python tools/bench_conv.py bf16 16 0,1,2,3,6,7,13 2>&1 | grep -v amdgpu.ids | sed 's/.*| wgrad/wgrad/'; done; done
  for sp in 5; do echo "== forced splits $sp"; VQ_ITERS=30 VQ_WGSPLIT=$sp timeout 100 python tools/bench_conv.py bf16 16 3,2 2>&1 | grep -v amdgpu.ids | sed 's/.*| wgrad/wgrad/'; done ) | tee gpurun_out/wgrad_xt_micro_r2y.log
for rep in 1 2; do for v in 0 512 32; do echo "== VQ_WGTILE=$v rep $rep"; VQ_WGTILE=$v timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print(d['value'], d['ms_per_step'], 'conv3x3', r['conv3x3']['frac'], 'igemm', r['frac'], 'wgrad', r['wgrad']['frac'])"; done; done | tee gpurun_out/bench_r2y.log
