#!/bin/bash
# Round 2, GPU call 21: epilogue transposition with 16-byte LDS writes (v_permlane32_swap; default) vs the 8-byte form (VQ_TILE = 8197 << 4)
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_hw_layout.py tests/test_kernels.py tests/test_model.py -m gpu -x -q -k "permlane or conv or golden or configs0 or deterministic or patch or nine_tap or groupnorm_statistics" > gpurun_out/tests_r2w.log 2>&1; tail -2 gpurun_out/tests_r2w.log
( for rep in 1 2; do for v in 0 131152; do echo "== VQ_TILE=$v rep $rep"; VQ_TILE=$v timeout 200 python tools/bench_epi.py 2>&1 | grep -v amdgpu.ids | grep "plain\|bias+res "; VQ_ITERS=30 VQ_TILE=$v timeout 100 python tools/bench_conv.py bf16 16 3,12 2>&1 | grep -v amdgpu.ids | sed 's/| wgrad.*//'; done; done ) | tee gpurun_out/epi_swap_r2w.log
for rep in 1 2; do for v in 0 131152; do echo "== VQ_TILE=$v rep $rep"; VQ_TILE=$v timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print(d['value'], d['ms_per_step'], 'conv3x3', r['conv3x3']['frac'], 'igemm', r['frac'], 'wgrad', r['wgrad']['frac'])"; done; done | tee gpurun_out/bench_r2w.log
