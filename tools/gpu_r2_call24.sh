#!/bin/bash
# Round 2, GPU call 24: range-owning XCD map in the one-tap weight-gradient kernel too (default) vs the round-1 plan (VQ_WGTILE=32): tests, bench, conv table
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels.py tests/test_model.py -m gpu -x -q -k "wgrad or conv or tile_owning or golden or configs0 or deterministic or subpixel" > gpurun_out/tests_r2z.log 2>&1; tail -2 gpurun_out/tests_r2z.log
for rep in 1 2; do for v in 0 32; do echo "== VQ_WGTILE=$v rep $rep"; VQ_WGTILE=$v timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --conv-table gpurun_out/conv_table_r2z_$v.txt 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print(d['value'], d['ms_per_step'], 'conv3x3', r['conv3x3']['frac'], 'igemm', r['frac'], 'wgrad', r['wgrad']['frac'])"; done; done | tee gpurun_out/bench_r2z.log
grep wgrad gpurun_out/conv_table_r2z_0.txt | head -24 > gpurun_out/wgrad_rows_r2z_new.txt; grep wgrad gpurun_out/conv_table_r2z_32.txt | head -24 > gpurun_out/wgrad_rows_r2z_old.txt
paste -d'|' gpurun_out/wgrad_rows_r2z_new.txt gpurun_out/wgrad_rows_r2z_old.txt | cut -c1-200 | head -24
