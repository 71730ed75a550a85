#!/bin/bash
mkdir -p gpurun_out
( VQ_TAP9_MODE=${1:-5} timeout 40 python tools/check_tap9.py 2>&1 | grep -v amdgpu.ids ) > gpurun_out/tap9_check_${1:-5}.log; cat gpurun_out/tap9_check_${1:-5}.log
for t in 0 5 ${1:-5}; do ( VQ_TILE=$t timeout 30 python tools/bench_conv.py bf16 16 4 2>&1 | grep -v amdgpu.ids ) > gpurun_out/tap9_bench_$t.log; echo "VQ_TILE=$t"; cat gpurun_out/tap9_bench_$t.log | cut -c1-112; done
