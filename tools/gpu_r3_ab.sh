#!/bin/bash
# Round 3, GPU call AB: price of folding the GroupNorm backward sums into the data-gradient conv's epilogue against
# the reduction pass it would replace (per shape: rocprofv3 kernel statistics of tools/bench_gn.py)
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 200 python tools/bench_gn_fusion_price.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r3ab_gn_fusion_price.txt
for i in 0 1 2; do
  ( cd /tmp && timeout 120 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_gn -o p -- python $GRAFT_REPO_ROOT/tools/bench_gn.py 16 $i > $GRAFT_REPO_ROOT/gpurun_out/prof_gn_run.log 2>&1 )
  db=$(find gpurun_out/prof_gn -name "*.db" | head -1)
  echo "== reduction / apply passes of vq_gn_silu_bwd, shape $i of tools/bench_gn.py (bf16): $(grep 'tensor' gpurun_out/prof_gn_run.log | cut -c1-60)" >> gpurun_out/r3ab_gn_fusion_price.txt
  [ -n "$db" ] && python tools/rocpd_stats.py "$db" /tmp/gn_$i.csv 2>&1 | grep "gn_" >> gpurun_out/r3ab_gn_fusion_price.txt
  rm -rf gpurun_out/prof_gn
done
cat gpurun_out/r3ab_gn_fusion_price.txt
