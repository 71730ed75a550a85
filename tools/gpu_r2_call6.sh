#!/bin/bash
# Round 2, GPU call 6: current bench line with conv table + HBM families, kernel stats of the same command.
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --conv-table gpurun_out/conv_table_r2f_ref.txt > gpurun_out/bench_r2f_ref.log 2>&1
tail -c 3000 gpurun_out/bench_r2f_ref.log
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r2f -o p -- \
    python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > $GRAFT_REPO_ROOT/gpurun_out/prof_r2f_run.log 2>&1 )
db=$(find gpurun_out/prof_r2f -name "*.db" | head -1)
[ -n "$db" ] && python tools/rocpd_stats.py "$db" gpurun_out/kernel_stats_r2f_ref.csv > gpurun_out/kernel_stats_r2f_ref.txt 2>&1
head -60 gpurun_out/kernel_stats_r2f_ref.txt
rm -rf gpurun_out/prof_r2f
