#!/bin/bash
# Round 3, GPU call G: c64 kernel's epilogue (43 % of its time, call F): streaming vs ordinary output stores, no stores, no LDS writes
# (`make ablate` library; VQ_TILE = (8192 + k) << 4: k = 1 no stores, 2 no transposition writes, 3 ordinary stores)
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
A=$GRAFT_REPO_ROOT/build/ablate/libvqhip_ablate.so
( for rep in 1 2; do for v in 0 131088 131104 131120 131072; do
    echo "== fp16 VQ_TILE=$v rep $rep"; VQ_ABLATE_LIB=$A VQ_ITERS=30 VQ_TILE=$v timeout 100 python tools/bench_conv.py fp16 16 12,0 2>&1 | grep -v amdgpu.ids | sed 's/| wgrad.*//'
  done; done ) > gpurun_out/r3g_c64_epilogue.txt 2>&1
cat gpurun_out/r3g_c64_epilogue.txt
