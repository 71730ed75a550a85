#!/bin/bash
# Round 2, GPU call 16 (r2zz): the round's closing run — PMC traffic of the ref policy, the full bench line that reads it, conv table, kernel stats,
# configs[1] / configs[4] lines, the full -m gpu suite
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
bash tools/gpu_traffic.sh r2zz ref > gpurun_out/traffic_r2zz_run.log 2>&1; tail -3 gpurun_out/traffic_r2zz_run.log
[ -f gpurun_out/traffic_r2zz.json ] && cp gpurun_out/traffic_r2zz.json profiles/r2zz_traffic.json
timeout 900 python bench.py --steps 20 --warmup 5 --conv-table gpurun_out/conv_table_r2zz_ref.txt > gpurun_out/bench_r2zz_ref.log 2>&1
tail -c 5500 gpurun_out/bench_r2zz_ref.log
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r2zz -o p -- \
    python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > $GRAFT_REPO_ROOT/gpurun_out/prof_r2zz_run.log 2>&1 )
db=$(find gpurun_out/prof_r2zz -name "*.db" | head -1)
[ -n "$db" ] && python tools/rocpd_stats.py "$db" gpurun_out/kernel_stats_r2zz_ref.csv > gpurun_out/kernel_stats_r2zz_ref.txt 2>&1
rm -rf gpurun_out/prof_r2zz
for w in c2 c5; do timeout 400 python bench.py --workload $w --steps 8 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 > gpurun_out/bench_r2zz_$w.json; python -c "
import json; d=json.loads(open('gpurun_out/bench_r2zz_$w.json').read()); print('$w', d['value'], d['ms_per_step'], d['roofline']['conv3x3']['frac'] if d.get('roofline') else None)"; done
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/tests_r2zz.log 2>&1; tail -4 gpurun_out/tests_r2zz.log
