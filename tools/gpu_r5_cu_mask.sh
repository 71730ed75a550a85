#!/bin/bash
# A/B: the weight-gradient stream with a CU mask of n CUs (VQ_SIDE_CU_MASK=n) and the split plans sized for n CUs (VQ_WGRAD_CUS=n)
# against the ordinary stream.  usage (GPU box): bash tools/gpu_r5_cu_mask.sh <tag>
O=gpurun_out/${1:-r5}_ab_cu_mask.txt; mkdir -p gpurun_out; : > $O
for rep in 1 2; do for n in 0 224 192 160; do
  if [ $n = 0 ]; then E=""; else E="VQ_SIDE_CU_MASK=$n VQ_WGRAD_CUS=$n"; fi
  env $E timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > gpurun_out/cu_mask_$n.log 2>&1
  echo "mask $n rep $rep: $(grep -o '"value": [0-9.]*' gpurun_out/cu_mask_$n.log | head -1) $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/cu_mask_$n.log | head -1) $(grep -o '"wgrad_frac[a-z_]*": [0-9.]*' gpurun_out/cu_mask_$n.log | tr '\n' ' ')" | tee -a $O
done; done
