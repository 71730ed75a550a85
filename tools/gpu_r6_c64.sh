#!/bin/bash
# round 6: Cin = 64 nine-tap launches with ONE halo buffer of LDS (4 blocks per CU) against two (hint dbg 72 = the two-buffer launch)
cd "${GRAFT_REPO_ROOT:-.}"
for prec in bf16 fp16; do
  for t in 0 1152; do
    echo "== $prec VQ_TILE=$t B=16 (c3: VGG conv1_2 @256), B=12 (l1: HR decoder @512)"
    VQ_TILE=$t python tools/bench_conv.py $prec 16 12,12 2>&1 | grep -v amdgpu.ids
    VQ_TILE=$t python tools/bench_conv.py $prec 12 16,17 2>&1 | grep -v amdgpu.ids
  done
done
