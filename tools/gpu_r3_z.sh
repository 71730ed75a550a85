#!/bin/bash
# Round 3, closing run: PMC traffic pass (so that the bench line carries roofline.traffic of THESE kernel sources), the driver-style
# bench line (20 steps, cpu_baseline + parity + parity_randomized + HBM pass) with the per-layer table, configs[1] and the configs[4]
# share with its parity object, kernel statistics (rocprofv3 --kernel-trace --stats)
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG="${1:-r3z}"
bash tools/gpu_traffic.sh $TAG ref > gpurun_out/${TAG}_traffic_run.log 2>&1; tail -3 gpurun_out/${TAG}_traffic_run.log
cp gpurun_out/traffic_$TAG.json profiles/${TAG}_traffic.json 2>/dev/null      # bench.py reads profiles/*_traffic.json (keyed to the kernel sources)
( timeout 900 python bench.py --steps 20 --warmup 5 --conv-table gpurun_out/${TAG}_conv_table_c3_ref.txt > gpurun_out/${TAG}_bench_c3_ref.json.log 2>&1; echo "rc=$?" >> gpurun_out/${TAG}_bench_c3_ref.json.log )
tail -2 gpurun_out/${TAG}_bench_c3_ref.json.log | cut -c1-600
( timeout 300 python bench.py --workload c2 --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > gpurun_out/${TAG}_bench_c2_ref.json.log 2>&1 ); tail -1 gpurun_out/${TAG}_bench_c2_ref.json.log | cut -c1-200
( timeout 600 python bench.py --workload c5 --steps 6 --warmup 2 --no-secondary > gpurun_out/${TAG}_bench_c5_ref.json.log 2>&1 ); tail -1 gpurun_out/${TAG}_bench_c5_ref.json.log | cut -c1-200
( cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-secondary > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof_run.log 2>&1
  db=$(find $GRAFT_REPO_ROOT/gpurun_out/prof -name "*.db" | head -1)
  [ -n "$db" ] && python $GRAFT_REPO_ROOT/tools/rocpd_stats.py "$db" $GRAFT_REPO_ROOT/gpurun_out/${TAG}_bench_c3_kernel_stats_ref.csv > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_kernel_stats.txt 2>&1
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof )
head -14 gpurun_out/${TAG}_kernel_stats.txt
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/${TAG}_gpu_suite.log 2>&1; grep -n "passed\|failed" gpurun_out/${TAG}_gpu_suite.log | tail -2
