#!/bin/bash
# Round 2, GPU call 17: the 128 x 512 patch tile (VQ_TILE = 4096 << 4 = 65536) against the nine-tap register-weight kernel
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_kernels.py -m gpu -x -q -k "128x512 or patch_staged" > gpurun_out/tests_r2r.log 2>&1; tail -2 gpurun_out/tests_r2r.log
( for rep in 1 2; do for v in 0 65536; do echo "== VQ_TILE=$v rep $rep"; VQ_ITERS=30 VQ_TILE=$v timeout 100 python tools/bench_conv.py bf16 16 0,3,7,8 2>&1 | grep -v amdgpu.ids | sed 's/| wgrad.*//'; done; done
  for v in 0 65536; do echo "== fp16 VQ_TILE=$v"; VQ_ITERS=30 VQ_TILE=$v timeout 100 python tools/bench_conv.py fp16 16 0,3 2>&1 | grep -v amdgpu.ids | sed 's/| wgrad.*//'; done ) | tee gpurun_out/p12_micro_r2r.log
