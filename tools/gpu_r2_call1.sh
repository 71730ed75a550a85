#!/bin/bash
# Round 2, GPU call 1: reference-precision (ref3) bench line with parity, full fp32x3 throughput, bf16 kernel stats, PMC traffic.
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 400 python bench.py --steps 10 --warmup 3 --conv-table gpurun_out/conv_table_r2a_ref3.txt > gpurun_out/bench_r2a_ref3.log 2>&1
tail -c 3000 gpurun_out/bench_r2a_ref3.log
timeout 200 python bench.py --precision fp32x3 --steps 4 --warmup 2 --no-secondary --no-cpu-baseline > gpurun_out/bench_r2a_fp32x3.log 2>&1
tail -c 1200 gpurun_out/bench_r2a_fp32x3.log
timeout 200 python bench.py --precision bf16 --steps 10 --warmup 3 --no-cpu-baseline --conv-table gpurun_out/conv_table_r2a_bf16.txt > gpurun_out/bench_r2a_bf16.log 2>&1
tail -c 1500 gpurun_out/bench_r2a_bf16.log
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r2a -o p -- \
    python $GRAFT_REPO_ROOT/bench.py --precision bf16 --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > $GRAFT_REPO_ROOT/gpurun_out/prof_r2a_run.log 2>&1 )
find gpurun_out/prof_r2a -name "*kernel_stats*" | head
f=$(find gpurun_out/prof_r2a -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/kernel_stats_r2a_bf16.csv && head -40 gpurun_out/kernel_stats_r2a_bf16.csv
rm -rf gpurun_out/prof_r2a
bash tools/gpu_traffic.sh r2a bf16
