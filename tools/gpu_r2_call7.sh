#!/bin/bash
# Round 2, GPU call 7: split-K plan exploration of the short-reduction weight gradients (512 channels at 32x32 / 16x16 / 64x64)
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp VQ_ITERS=30
( for sp in 0 1 2 3 4 6 8 12 16 24; do echo "== three-tap kernel, forced splits $sp"; VQ_WGSPLIT=$sp timeout 100 python tools/bench_conv.py fp16 16 3,13,2 2>&1 | grep -v amdgpu.ids | sed 's/.*| wgrad/wgrad/'; done
  for bt in 64 128 256; do for sp in 0 1 2 4 8; do echo "== one-tap tile $bt, forced splits $sp"; VQ_WGTILE=$bt VQ_WGSPLIT=$sp timeout 100 python tools/bench_conv.py fp16 16 3,13,2 2>&1 | grep -v amdgpu.ids | sed 's/.*| wgrad/wgrad/'; done; done ) | tee gpurun_out/wgrad_split_r2g.log
