#!/bin/bash
# Round 3, GPU call D: three-tap weight gradient with two / three-step fragment prefetch (default) vs one step (VQ_WGTILE=32), with and
# without the staging stagger (+16); the multi-lane split reduction is in all of them (compare with r3c).  Bench A/B in alternating order.
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python -m pytest tests -m gpu -x -q -k "wgrad or adjoint or full_size or trajectory" > gpurun_out/tests_r3d.log 2>&1; tail -3 gpurun_out/tests_r3d.log | head -1
( for rep in 1 2; do for v in 0 32 16 48; do echo "== VQ_WGTILE=$v rep $rep"; VQ_ITERS=30 VQ_WGTILE=$v timeout 100 python tools/bench_conv.py bf16 16 0,1,2,3,6,7 2>&1 | grep -v amdgpu.ids | sed 's/.*| wgrad/wgrad/'; done; done
  for v in 0 32; do echo "== fp16 VQ_WGTILE=$v"; VQ_ITERS=30 VQ_WGTILE=$v timeout 100 python tools/bench_conv.py fp16 16 0,1,2,3 2>&1 | grep -v amdgpu.ids | sed 's/.*| wgrad/wgrad/'; done ) > gpurun_out/r3d_wgrad_deep_micro.txt 2>&1
cat gpurun_out/r3d_wgrad_deep_micro.txt
for k in "0 1" "32 1" "48 1" "48 2" "32 2" "0 2"; do set -- $k
  VQ_WGTILE=$1 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 > gpurun_out/bench_r3d_$1_$2.json
  python - <<PY
import json
d = json.loads(open("gpurun_out/bench_r3d_$1_$2.json").read())
r = d["roofline"]
print("VQ_WGTILE=$1 rep $2:", d["value"], "img/s", d["ms_per_step"], "ms igemm", r["frac"], "conv3x3", r["conv3x3"]["frac"], "wgrad", r["wgrad"]["frac"])
PY
done 2>&1 | tee gpurun_out/r3d_bench_ab.txt
