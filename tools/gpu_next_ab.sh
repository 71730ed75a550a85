#!/bin/bash
# First GPU call of the next round: hardware check + micro-benchmark of the two kernel candidates that so far only ran on the
# host emulator (DESIGN.md §6 "cheap follow-ups"), then the full-step bench with the nine-tap kernel on / off.
#   knob 5            nine-tap kernel wherever the shape allows, incl. the 64-row tile (4 waves x 64c x 32p) for 64-channel layers
#   knob 1 + (32<<4)  register-weight one-tap 128x128 tile as 2 x 2 waves of 64c x 64p
#   knob 6            nine-tap and three-tap kernels off (the tiles adopted before v35)
set -u
mkdir -p gpurun_out
( VQ_TAP9_MODE=5 timeout 60 python tools/check_tap9.py 2>&1 | grep -v amdgpu.ids ) | tee gpurun_out/next_check.log
for t in 0 5 513 6; do
  echo "== VQ_TILE=$t"; ( VQ_TILE=$t timeout 60 python tools/bench_conv.py bf16 16 13 2>&1 | grep -v amdgpu.ids ) | tee gpurun_out/next_bench_conv_$t.log | cut -c1-120
done
for t in 0 6; do
  ( VQ_TILE=$t timeout 100 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --conv-table gpurun_out/next_conv_table_$t.txt > gpurun_out/next_bench_$t.log 2>&1 )
  echo "VQ_TILE=$t: $(grep -o '"value": [0-9.]*' gpurun_out/next_bench_$t.log) $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/next_bench_$t.log)"
done
