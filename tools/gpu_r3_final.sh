#!/bin/bash
# Round 3, last GPU call: the full -m gpu suite, smoke() and the default bench line on the committed state
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/r3final_gpu_suite.log 2>&1; grep -n "passed\|failed" gpurun_out/r3final_gpu_suite.log | tail -2
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r3final_smoke.log 2>&1; tail -1 gpurun_out/r3final_smoke.log
timeout 900 python bench.py > gpurun_out/r3final_bench_default.json.log 2>&1; tail -1 gpurun_out/r3final_bench_default.json.log | cut -c1-400
