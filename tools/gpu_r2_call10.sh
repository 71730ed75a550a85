#!/bin/bash
# Round 2, GPU call 10: PMC traffic of the default (ref) precision, then the full bench line that reads it, kernel stats, full -m gpu suite
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
bash tools/gpu_traffic.sh r2j ref > gpurun_out/traffic_r2j_run.log 2>&1; tail -5 gpurun_out/traffic_r2j_run.log
[ -f gpurun_out/traffic_r2j.json ] && cp gpurun_out/traffic_r2j.json profiles/r2j_traffic.json
timeout 900 python bench.py --steps 20 --warmup 5 --conv-table gpurun_out/conv_table_r2j_ref.txt > gpurun_out/bench_r2j_ref.log 2>&1
tail -c 6000 gpurun_out/bench_r2j_ref.log
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r2j -o p -- \
    python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > $GRAFT_REPO_ROOT/gpurun_out/prof_r2j_run.log 2>&1 )
db=$(find gpurun_out/prof_r2j -name "*.db" | head -1)
[ -n "$db" ] && python tools/rocpd_stats.py "$db" gpurun_out/kernel_stats_r2j_ref.csv > gpurun_out/kernel_stats_r2j_ref.txt 2>&1
rm -rf gpurun_out/prof_r2j
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/tests_r2j.log 2>&1; tail -5 gpurun_out/tests_r2j.log
