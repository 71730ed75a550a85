#!/bin/bash
# Round 2, GPU call 9: per-wave GroupNorm partial rows in the conv epilogue: tests, epilogue micro, bench A/B (fusion on / off)
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels.py tests/test_model.py -m gpu -x -q -k "groupnorm or gn_ or statistics_ride or golden or conv" > gpurun_out/tests_r2i.log 2>&1; tail -3 gpurun_out/tests_r2i.log
timeout 200 python tools/bench_epi.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/epi_r2i.log
for rep in 1 2; do for v in 1 0; do echo "== VQ_GN_FUSED=$v rep $rep"; VQ_GN_FUSED=$v timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print(d['value'], d['ms_per_step'], 'conv3x3', r['conv3x3']['frac'], 'igemm', r['frac'], 'wgrad', r['wgrad']['frac'])"; done; done | tee gpurun_out/bench_r2i.log
