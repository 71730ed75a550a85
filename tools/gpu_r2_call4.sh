#!/bin/bash
# Round 2, GPU call 4: nine-tap kernel weight-stream variants (A/B on one box): correctness, then per-shape timings.
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_kernels.py -m gpu -x -q -k "nine_tap or tile_modes or fp16_storage" > gpurun_out/tests_r2d.log 2>&1; tail -3 gpurun_out/tests_r2d.log
for rep in 1 2; do for dbg in 0 2048 4096 8192; do
  echo "== VQ_TILE=$dbg rep $rep"; VQ_TILE=$dbg timeout 120 python tools/bench_conv.py bf16 16 0,3,12 2>&1 | grep -v amdgpu.ids
done; done | tee gpurun_out/tap9_variants_r2d.log
echo "== fp16"; for dbg in 0 2048 8192; do echo "== VQ_TILE=$dbg"; VQ_TILE=$dbg timeout 120 python tools/bench_conv.py fp16 16 0,3 2>&1 | grep -v amdgpu.ids; done | tee -a gpurun_out/tap9_variants_r2d.log
echo "== knob 5 (64-row nine-tap tile) on 64->64"; for t in 0 5; do VQ_TILE=$t timeout 120 python tools/bench_conv.py bf16 16 12,12 2>&1 | grep -v amdgpu.ids; done | tee -a gpurun_out/tap9_variants_r2d.log
