#!/bin/bash
# Round 3, GPU call AD: does the PRESENCE of the fused-sum path (off by default) cost the default step anything?  The tree of the commit
# before it (build/prevtree, ABI v6) against the current one, alternating; then the PMC traffic pass on the final kernel sources.
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 400 python -m pytest tests -m gpu -x -q -k "groupnorm or conv_fwd_dgrad or abi or train_step or configs0" > gpurun_out/tests_r3ad.log 2>&1; tail -1 gpurun_out/tests_r3ad.log
for k in "new 1" "prev 1" "prev 2" "new 2"; do set -- $k
  if [ $1 = prev ]; then dir=$GRAFT_REPO_ROOT/build/prevtree; else dir=$GRAFT_REPO_ROOT; fi
  ( cd $dir && timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 ) > gpurun_out/bench_r3ad_$1_$2.json
  python - <<PY
import json
d = json.loads(open("gpurun_out/bench_r3ad_$1_$2.json").read())
r = d["roofline"]
print("$1 rep $2:", d["value"], "img/s", d["ms_per_step"], "ms igemm", r["frac"], "conv3x3", r["conv3x3"]["frac"], "wgrad", r["wgrad"]["frac"])
PY
done 2>&1 | tee gpurun_out/r3ad_bench_ab2.txt
bash tools/gpu_traffic.sh r3zz ref > gpurun_out/r3zz_traffic_run.log 2>&1; tail -2 gpurun_out/r3zz_traffic_run.log
