#!/bin/bash
# Short GPU call: selected -m gpu tests (KEXPR), a micro-benchmark (MICRO, optional), the bench line and the kernel stats.
# usage: bash tools/gpu_quick.sh "<pytest -k expr>" "<micro-benchmark command or empty>" <tag>
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
K="$1"; MICRO="$2"; TAG="$3"
( timeout 200 python -m pytest tests -m gpu -x -q -k "$K" > gpurun_out/tests_$TAG.log 2>&1; echo "pytest rc=$?" >> gpurun_out/tests_$TAG.log )
tail -4 gpurun_out/tests_$TAG.log
if [ -n "$MICRO" ]; then ( timeout 120 $MICRO > gpurun_out/micro_$TAG.log 2>&1 ); cat gpurun_out/micro_$TAG.log | tail -12; fi
( timeout 100 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --conv-table gpurun_out/conv_table_$TAG.txt > gpurun_out/bench_$TAG.log 2>&1; echo "rc=$?" >> gpurun_out/bench_$TAG.log )
tail -2 gpurun_out/bench_$TAG.log | cut -c1-330
( cd /tmp && timeout 150 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_run_$TAG.log 2>&1
  db=$(ls $GRAFT_REPO_ROOT/gpurun_out/prof/*/*.db $GRAFT_REPO_ROOT/gpurun_out/prof/*.db 2>/dev/null | head -1)
  [ -n "$db" ] && python $GRAFT_REPO_ROOT/tools/rocpd_stats.py "$db" $GRAFT_REPO_ROOT/gpurun_out/kernel_stats_$TAG.csv > $GRAFT_REPO_ROOT/gpurun_out/kernel_stats_$TAG.txt 2>&1
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof )
grep -i "gn_\|columns" -v gpurun_out/kernel_stats_$TAG.txt | head -4; grep "gn_" gpurun_out/kernel_stats_$TAG.txt
