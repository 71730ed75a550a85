#!/bin/bash
# Round 2, GPU call 15: single-phase 128-row patch kernel (VQ_TILE = 2048 << 4) vs the nine-tap register-weight kernel and the two-phase form
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_kernels.py -m gpu -x -q -k "patch_staged" > gpurun_out/tests_r2p.log 2>&1; tail -2 gpurun_out/tests_r2p.log
( for rep in 1 2; do for v in 0 16384 32768; do echo "== VQ_TILE=$v rep $rep"; VQ_ITERS=30 VQ_TILE=$v timeout 100 python tools/bench_conv.py bf16 16 0,3,7,8 2>&1 | grep -v amdgpu.ids | sed 's/| wgrad.*//'; done; done
  for v in 0 32768; do echo "== fp16 VQ_TILE=$v"; VQ_ITERS=30 VQ_TILE=$v timeout 100 python tools/bench_conv.py fp16 16 0,3 2>&1 | grep -v amdgpu.ids | sed 's/| wgrad.*//'; done ) | tee gpurun_out/p9s_micro_r2p.log
