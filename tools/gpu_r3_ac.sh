#!/bin/bash
# Round 3, GPU call AC: GroupNorm backward with its sums formed in the data-gradient conv's epilogue (VqGnBwdFuse, ABI v7): the -m gpu
# tests that touch it, and the step with VQ_GN_BWD_FUSED=1 (default) / 0 in alternating order on one box
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 500 python -m pytest tests -m gpu -x -q -k "groupnorm or train_step or configs0 or trajectory or full_size_step or conv_fwd_dgrad or abi or resnet or golden" > gpurun_out/tests_r3ac.log 2>&1; tail -1 gpurun_out/tests_r3ac.log
for k in "1 1" "0 1" "0 2" "1 2"; do set -- $k
  VQ_GN_BWD_FUSED=$1 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 > gpurun_out/bench_r3ac_$1_$2.json
  python - <<PY
import json
d = json.loads(open("gpurun_out/bench_r3ac_$1_$2.json").read())
r = d["roofline"]
print("VQ_GN_BWD_FUSED=$1 rep $2:", d["value"], "img/s", d["ms_per_step"], "ms igemm", r["frac"], "conv3x3", r["conv3x3"]["frac"], "wgrad", r["wgrad"]["frac"])
PY
done 2>&1 | tee gpurun_out/r3ac_bench_ab.txt
