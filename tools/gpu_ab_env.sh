#!/bin/bash
# A/B of one environment knob on the same box: tests first, then bench.py with VAR=1 and VAR=0 (twice each, interleaved).
# usage: bash tools/gpu_ab_env.sh VAR "<pytest -k expr>" tag
set -u
mkdir -p gpurun_out
VAR="$1"; K="$2"; TAG="$3"
( timeout 150 python -m pytest tests -m gpu -x -q -k "$K" > gpurun_out/tests_$TAG.log 2>&1; echo "pytest rc=$?" >> gpurun_out/tests_$TAG.log ); tail -3 gpurun_out/tests_$TAG.log
for rep in 1 2; do for v in 1 0; do
  ( env $VAR=$v timeout 100 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --conv-table gpurun_out/conv_table_${TAG}_$v.txt > gpurun_out/bench_${TAG}_${v}_$rep.log 2>&1 )
  echo "$VAR=$v rep $rep: $(grep -o '"value": [0-9.]*' gpurun_out/bench_${TAG}_${v}_$rep.log) $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/bench_${TAG}_${v}_$rep.log)"
done; done
