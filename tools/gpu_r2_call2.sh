#!/bin/bash
# Round 2, GPU call 2: full -m gpu suite (incl. the binary16 twins), the "ref" (fp16-operand) bench line, its kernel stats.
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/tests_r2b.log 2>&1
tail -15 gpurun_out/tests_r2b.log
timeout 400 python bench.py --steps 10 --warmup 3 --conv-table gpurun_out/conv_table_r2b_ref.txt > gpurun_out/bench_r2b_ref.log 2>&1
tail -c 4500 gpurun_out/bench_r2b_ref.log
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r2b -o p -- \
    python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > $GRAFT_REPO_ROOT/gpurun_out/prof_r2b_run.log 2>&1 )
db=$(find gpurun_out/prof_r2b -name "*.db" | head -1)
[ -n "$db" ] && python tools/rocpd_stats.py "$db" gpurun_out/kernel_stats_r2b_ref.csv > gpurun_out/kernel_stats_r2b_ref.txt 2>&1
head -45 gpurun_out/kernel_stats_r2b_ref.txt
rm -rf gpurun_out/prof_r2b
