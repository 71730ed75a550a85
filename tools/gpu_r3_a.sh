#!/bin/bash
# Round 3, GPU call A: the -m gpu suite after the hint / range-event / pool-with-tap refactor, the 4-wave three-tap weight-gradient kernel
# against its 8-wave form (VQ_WGTILE=16) per layer and on the whole step, then the full bench line (parity legs included).
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/tests_r3a.log 2>&1; tail -5 gpurun_out/tests_r3a.log
( for rep in 1 2; do for v in 0 16; do echo "== VQ_WGTILE=$v rep $rep"; VQ_ITERS=30 VQ_WGTILE=$v timeout 100 python tools/bench_conv.py bf16 16 0,1,2,3,6,7 2>&1 | grep -v amdgpu.ids | sed 's/.*| wgrad/wgrad/'; done; done
  for v in 0 16; do echo "== fp16 VQ_WGTILE=$v"; VQ_ITERS=30 VQ_WGTILE=$v timeout 100 python tools/bench_conv.py fp16 16 0,1,2,3 2>&1 | grep -v amdgpu.ids | sed 's/.*| wgrad/wgrad/'; done ) > gpurun_out/r3a_wgrad4_micro.txt 2>&1
cat gpurun_out/r3a_wgrad4_micro.txt
for rep in 1 2; do for v in 0 16; do
  VQ_WGTILE=$v timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 > gpurun_out/bench_r3a_wg${v}_$rep.json
  python - <<PY
import json
d = json.loads(open("gpurun_out/bench_r3a_wg${v}_$rep.json").read())
print("VQ_WGTILE=$v rep $rep:", d["value"], "img/s", d["ms_per_step"], "ms conv3x3", d["roofline"]["conv3x3"]["frac"], "wgrad", d["roofline"]["wgrad"]["frac"])
PY
done; done 2>&1 | tee gpurun_out/r3a_wgrad4_bench_ab.txt
timeout 900 python bench.py --steps 20 --warmup 5 --conv-table gpurun_out/conv_table_r3a.txt > gpurun_out/bench_r3a.log 2>&1
tail -c 9000 gpurun_out/bench_r3a.log
