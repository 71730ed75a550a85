#!/bin/bash
# A/B of conv_wgrad3_kernel's pixel-window form (default) against one X fragment read per (k-step, tap) (vq_conv2d_wgrad hint +32):
# micro-benchmark of the layers it serves, then the step.  usage (GPU box): bash tools/gpu_r5_wgrad_window.sh <tag>
O=gpurun_out/${1:-r5}_wgrad_window.txt; mkdir -p gpurun_out; : > $O
for prec in bf16 fp16 f16x3; do for rep in 1 2; do for hint in 0 32; do
  echo "== $prec rep $rep VQ_WGTILE=$hint ($([ $hint = 0 ] && echo 'pixel window' || echo 'per-tap reads'))" >> $O
  VQ_WGTILE=$hint timeout 300 python tools/bench_conv.py $prec 16 0,1,2,3 2>&1 | grep -v amdgpu.ids | sed 's/.*| wgrad/wgrad/' >> $O
done; done; done
cat $O
