#!/bin/bash
# Round 2, GPU call 20: nine-tap kernel with 2 / 3 / 4 tiles per block (VQ_TILE = (16384 + n) << 4) vs one tile per block
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_kernels.py -m gpu -x -q -k "nine_tap" > gpurun_out/tests_r2v.log 2>&1; tail -2 gpurun_out/tests_r2v.log
( for rep in 1 2; do for v in 0 262176 262192 262208; do echo "== VQ_TILE=$v rep $rep"; VQ_TILE=$v timeout 200 python tools/bench_epi.py 2>&1 | grep -v amdgpu.ids | grep "128->128"; VQ_ITERS=30 VQ_TILE=$v timeout 100 python tools/bench_conv.py bf16 16 0,3,12 2>&1 | grep -v amdgpu.ids | sed 's/| wgrad.*//'; done; done ) | tee gpurun_out/tap9_tpb_r2v.log
