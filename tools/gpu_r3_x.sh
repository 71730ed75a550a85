#!/bin/bash
# Round 3, GPU call X: the round-2 candidates for the Cout = 128 layers (csrc/experimental: 128-row patch tiles two-phase / single-phase,
# the 128 x 512 tile) again, now that they share the lighter epilogue (`make ablate` library)
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
A=$GRAFT_REPO_ROOT/build/ablate/libvqhip_ablate.so
( for rep in 1 2; do for pr in fp16 bf16; do for v in 0 16384 32768 65536; do
    echo "== $pr VQ_TILE=$v rep $rep"; VQ_ABLATE_LIB=$A VQ_ITERS=30 VQ_TILE=$v timeout 100 python tools/bench_conv.py $pr 16 0,7 2>&1 | grep -v amdgpu.ids | sed 's/| wgrad.*//'
  done; done; done ) > gpurun_out/r3x_cout128_candidates.txt 2>&1
cat gpurun_out/r3x_cout128_candidates.txt
