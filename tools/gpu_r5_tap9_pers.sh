#!/bin/bash
# A/B of the nine-tap kernel's persistent form (default) against one tile per block (kernel_hint 72 << 4) on the layers it serves
# usage (GPU box): bash tools/gpu_r5_tap9_pers.sh <tag>
O=gpurun_out/${1:-r5}_tap9_pers.txt; mkdir -p gpurun_out; : > $O
for prec in bf16 fp16 f16x3; do for rep in 1 2; do for hint in 0 1152; do
  echo "== $prec rep $rep VQ_TILE=$hint ($([ $hint = 0 ] && echo persistent || echo one tile per block))" >> $O
  VQ_TILE=$hint timeout 300 python tools/bench_conv.py $prec 16 0,7,12 2>&1 | grep -v amdgpu.ids >> $O
done; done; done
cat $O
