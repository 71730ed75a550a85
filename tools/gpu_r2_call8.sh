#!/bin/bash
# Round 2, GPU call 8: nine-taps-per-thread split reduction, GN backward second level folded into the apply kernel: tests, micro, bench, kernel stats
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels.py tests/test_model.py tests/test_tae.py -m gpu -x -q -k "groupnorm or gn_ or statistics_ride or wgrad or split_reduction or golden or tvae" > gpurun_out/tests_r2h.log 2>&1; tail -3 gpurun_out/tests_r2h.log
( for v in 0 1; do echo "== VQ_WGTILE=$v"; VQ_ITERS=30 VQ_WGTILE=$v timeout 100 python tools/bench_conv.py fp16 16 1,2,3,13 2>&1 | grep -v amdgpu.ids | sed 's/.*| wgrad/wgrad/'; done ) | tee gpurun_out/reduce9_r2h.log
for rep in 1 2; do timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print(d['value'], d['ms_per_step'], 'conv3x3', r['conv3x3']['frac'], 'igemm', r['frac'], 'wgrad', r['wgrad']['frac'])"; done | tee gpurun_out/bench_r2h.log
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r2h -o p -- \
    python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > $GRAFT_REPO_ROOT/gpurun_out/prof_r2h_run.log 2>&1 )
db=$(find gpurun_out/prof_r2h -name "*.db" | head -1)
[ -n "$db" ] && python tools/rocpd_stats.py "$db" gpurun_out/kernel_stats_r2h_ref.csv > gpurun_out/kernel_stats_r2h_ref.txt 2>&1
head -45 gpurun_out/kernel_stats_r2h_ref.txt | cut -c1-150
rm -rf gpurun_out/prof_r2h
