#!/bin/bash
# Round 2, GPU call 14: LPIPS tap kernels with one hash round per 8 channels: tests + the bench line's hbm list
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_kernels.py tests/test_model.py -m gpu -x -q -k "lpips or deterministic" > gpurun_out/tests_r2o.log 2>&1; tail -2 gpurun_out/tests_r2o.log
timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/bench_r2o.json
python -c "
import json
d=json.loads(open('gpurun_out/bench_r2o.json').read()); r=d['roofline']
print(d['value'], d['ms_per_step'], 'conv3x3', r['conv3x3']['frac'], 'igemm', r['frac'], 'wgrad', r['wgrad']['frac'])
for h in d['hbm']: print(h['kernel'], h.get('ms_per_step'), h.get('GB/s'))"
