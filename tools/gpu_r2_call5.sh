#!/bin/bash
# Round 2, GPU call 5: GroupNorm statistics from the conv epilogue (A/B), 16 B/lane split reduction (A/B), GN sample-chunking probe.
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels.py tests/test_model.py -m gpu -x -q -k "groupnorm or statistics_ride or wgrad or nine_tap or fp16_storage or tile_modes" > gpurun_out/tests_r2e.log 2>&1; tail -3 gpurun_out/tests_r2e.log
for rep in 1 2; do
  for v in "VQ_GN_FUSED=1 VQ_WGTILE=0" "VQ_GN_FUSED=0 VQ_WGTILE=0" "VQ_GN_FUSED=1 VQ_WGTILE=1"; do
    echo "== $v rep $rep"
    env $v timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print(d['value'], d['ms_per_step'], 'conv3x3', r['conv3x3']['frac'], 'igemm', r['frac'], 'wgrad', r['wgrad']['frac'])
print([(h['kernel'], h['ms_per_step']) for h in d.get('hbm', [])])"
  done
done 2>&1 | tee gpurun_out/ab_r2e.log
timeout 200 python tools/bench_gn_chunk.py fp16 2>&1 | grep -v amdgpu.ids | tee gpurun_out/gn_chunk_r2e.log
