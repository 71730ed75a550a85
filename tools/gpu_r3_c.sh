#!/bin/bash
# Round 3, GPU call C: staggered staging in the three-tap weight-gradient kernel (default) vs unstaggered (VQ_WGTILE=16); 64x64 tiles on
# the short-M layers (default) vs 64x128 (VQ_TILE=256 = dbg 16); the -m gpu suite incl. the new full-size / trajectory / yardstick tests.
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
( for rep in 1 2; do for v in 0 16; do echo "== VQ_WGTILE=$v rep $rep"; VQ_ITERS=30 VQ_WGTILE=$v timeout 100 python tools/bench_conv.py bf16 16 0,1,2,3,6,7 2>&1 | grep -v amdgpu.ids | sed 's/.*| wgrad/wgrad/'; done; done
  for v in 0 16; do echo "== fp16 VQ_WGTILE=$v"; VQ_ITERS=30 VQ_WGTILE=$v timeout 100 python tools/bench_conv.py fp16 16 0,1,2,3 2>&1 | grep -v amdgpu.ids | sed 's/.*| wgrad/wgrad/'; done ) > gpurun_out/r3c_wgrad_stagger_micro.txt 2>&1
cat gpurun_out/r3c_wgrad_stagger_micro.txt
( for rep in 1 2; do for v in 0 256; do echo "== VQ_TILE=$v rep $rep"; VQ_ITERS=50 VQ_TILE=$v timeout 100 python tools/bench_conv.py bf16 16 13,14 2>&1 | grep -v amdgpu.ids | sed 's/| wgrad.*//'; VQ_ITERS=50 VQ_TILE=$v timeout 100 python tools/bench_conv.py fp16 16 13 2>&1 | grep -v amdgpu.ids | sed 's/| wgrad.*//'; done; done ) > gpurun_out/r3c_short_m_micro.txt 2>&1
cat gpurun_out/r3c_short_m_micro.txt
for rep in 1 2; do for k in "0 0" "16 0" "0 256"; do set -- $k
  VQ_WGTILE=$1 VQ_TILE=$2 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 > gpurun_out/bench_r3c_$1_$2_$rep.json
  python - <<PY
import json
d = json.loads(open("gpurun_out/bench_r3c_$1_$2_$rep.json").read())
print("VQ_WGTILE=$1 VQ_TILE=$2 rep $rep:", d["value"], "img/s", d["ms_per_step"], "ms conv3x3", d["roofline"]["conv3x3"]["frac"], "wgrad", d["roofline"]["wgrad"]["frac"], "dropped", d["config"]["fp16_after_run"]["optimizer_steps_dropped"])
PY
done; done 2>&1 | tee gpurun_out/r3c_bench_ab.txt
timeout 1200 python -m pytest tests -m gpu -q -s > gpurun_out/tests_r3c.log 2>&1; grep -E "passed|failed|configs0 parity|trajectory, worst|Error|FAILED" gpurun_out/tests_r3c.log | cut -c1-400 | head -30
