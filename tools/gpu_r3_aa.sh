#!/bin/bash
# Round 3, GPU call AA: LDS writes and stores batched behind their packed data (write-after-read on store operands): tests, cycle
# stamps, per layer and in the step against the build before (build/prev)
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 400 python -m pytest tests -m gpu -x -q -k "conv or tile or nine_tap or persistent_patch or 32_row or groupnorm or range_events or subpixel or adjoint or full_size or lpips or configs0" > gpurun_out/tests_r3aa.log 2>&1; tail -1 gpurun_out/tests_r3aa.log
( for pr in fp16 bf16; do for sh in 1 0; do timeout 60 python tools/stamps.py $pr $sh; done; done ) 2>&1 | grep -v amdgpu.ids > gpurun_out/r3aa_stamps.txt
cat gpurun_out/r3aa_stamps.txt
P=$GRAFT_REPO_ROOT/build/prev/libvqhip_prev.so
( for rep in 1 2; do for pr in fp16 bf16; do
    echo "== new $pr rep $rep"; VQ_ITERS=30 timeout 100 python tools/bench_conv.py $pr 16 0,1,2,3,12 2>&1 | grep -v amdgpu.ids | sed 's/| wgrad.*//'
    echo "== prev $pr rep $rep"; VQ_ABLATE_LIB=$P VQ_ITERS=30 timeout 100 python tools/bench_conv.py $pr 16 0,1,2,3,12 2>&1 | grep -v amdgpu.ids | sed 's/| wgrad.*//'
  done; done ) > gpurun_out/r3aa_simple_micro.txt 2>&1
cat gpurun_out/r3aa_simple_micro.txt
for k in "new 1" "prev 1" "prev 2" "new 2"; do set -- $k
  if [ $1 = prev ]; then export VQ_BENCH_AB_LIB=$P; else unset VQ_BENCH_AB_LIB; fi
  timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 > gpurun_out/bench_r3aa_$1_$2.json
  python - <<PY
import json
d = json.loads(open("gpurun_out/bench_r3aa_$1_$2.json").read())
r = d["roofline"]
print("$1 rep $2:", d["value"], "img/s", d["ms_per_step"], "ms igemm", r["frac"], "conv3x3", r["conv3x3"]["frac"], "wgrad", r["wgrad"]["frac"])
PY
done 2>&1 | tee gpurun_out/r3aa_bench_ab.txt
