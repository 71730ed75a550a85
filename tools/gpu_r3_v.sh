#!/bin/bash
# Round 3, GPU call V: PMC traffic pass on the final kernel sources (comment-only change since r3zz: the file is keyed to a source hash)
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
bash tools/gpu_traffic.sh r3zz ref > gpurun_out/r3zz_traffic_run.log 2>&1; tail -2 gpurun_out/r3zz_traffic_run.log
timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | cut -c1-200
