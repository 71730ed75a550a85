#!/bin/bash
# Round 3, GPU call J: weight-gradient kernels with swapped MFMA operands (D^T) so that the split-K partial slabs are written with
# 16-byte stores (24 per lane instead of 96): per layer and in the step against the build before (build/prev/libvqhip_prev.so)
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python -m pytest tests -m gpu -x -q -k "wgrad or adjoint or full_size or conv_fwd_dgrad or fuzz" > gpurun_out/tests_r3j.log 2>&1; tail -1 gpurun_out/tests_r3j.log
P=$GRAFT_REPO_ROOT/build/prev/libvqhip_prev.so
( for rep in 1 2; do for pr in fp16 bf16; do
    echo "== new $pr rep $rep"; VQ_ITERS=30 timeout 100 python tools/bench_conv.py $pr 16 0,1,2,3,6,7,8,13 2>&1 | grep -v amdgpu.ids | sed 's/fwd.*| wgrad/wgrad/'
    echo "== prev $pr rep $rep"; VQ_ABLATE_LIB=$P VQ_ITERS=30 timeout 100 python tools/bench_conv.py $pr 16 0,1,2,3,6,7,8,13 2>&1 | grep -v amdgpu.ids | sed 's/fwd.*| wgrad/wgrad/'
  done; done ) > gpurun_out/r3j_wgrad_store16_micro.txt 2>&1
cat gpurun_out/r3j_wgrad_store16_micro.txt
for k in "new 1" "prev 1" "prev 2" "new 2"; do set -- $k
  if [ $1 = prev ]; then export VQ_BENCH_AB_LIB=$P; else unset VQ_BENCH_AB_LIB; fi
  VQ_TILE=512 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 > gpurun_out/bench_r3j_$1_$2.json
  python - <<PY
import json
d = json.loads(open("gpurun_out/bench_r3j_$1_$2.json").read())
r = d["roofline"]
print("$1 rep $2:", d["value"], "img/s", d["ms_per_step"], "ms igemm", r["frac"], "conv3x3", r["conv3x3"]["frac"], "wgrad", r["wgrad"]["frac"])
PY
done 2>&1 | tee gpurun_out/r3j_bench_ab.txt
