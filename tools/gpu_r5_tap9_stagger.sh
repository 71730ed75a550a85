#!/bin/bash
# The nine-tap kernel's two blocks per CU out of lockstep: VQ_TAP9_STAGGER = start-up delay (~us) of the first wave's second block per CU.
# usage (GPU box): bash tools/gpu_r5_tap9_stagger.sh <tag>
O=gpurun_out/${1:-r5}_tap9_stagger.txt; mkdir -p gpurun_out; : > $O
for prec in bf16 fp16; do for rep in 1 2; do for st in 0 3 5 7 10; do
  echo "== $prec rep $rep VQ_TAP9_STAGGER=$st" >> $O
  VQ_TAP9_STAGGER=$st timeout 300 python tools/bench_conv.py $prec 16 0,7,12 2>&1 | grep -v amdgpu.ids | sed 's/| wgrad.*//' >> $O
done; done; done
cat $O
