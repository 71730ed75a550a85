#!/bin/bash
# Round 2, GPU call 13: three-tap weight-gradient kernel as a three-buffer ring with interleaved staging (default) vs its two-buffer form (VQ_WGTILE=2)
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels.py tests/test_model.py -m gpu -x -q -k "wgrad_lds_dma" > gpurun_out/tests_r2n.log 2>&1; tail -3 gpurun_out/tests_r2n.log
( for rep in 1 2; do for v in 0 2 8; do echo "== VQ_WGTILE=$v rep $rep"; VQ_ITERS=30 VQ_WGTILE=$v timeout 100 python tools/bench_conv.py bf16 16 0,1,2,3,6,7 2>&1 | grep -v amdgpu.ids | sed 's/.*| wgrad/wgrad/'; done; done
  for v in 0 2; do echo "== fp16 VQ_WGTILE=$v"; VQ_ITERS=30 VQ_WGTILE=$v timeout 100 python tools/bench_conv.py fp16 16 0,1,2,3 2>&1 | grep -v amdgpu.ids | sed 's/.*| wgrad/wgrad/'; done ) | tee gpurun_out/wgrad_ring_micro_r2n.log
for rep in 1; do for v in 0 8; do echo "== VQ_WGTILE=$v rep $rep"; VQ_WGTILE=$v timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print(d['value'], d['ms_per_step'], 'conv3x3', r['conv3x3']['frac'], 'igemm', r['frac'], 'wgrad', r['wgrad']['frac'])"; done; done | tee gpurun_out/bench_r2n.log
