#!/bin/bash
# Round 3, GPU call K: in-step A/B of builds — current vs the epilogue rewrite alone (build/prev), current without SLP vectorisation
# (-fno-slp-vectorize: the guide prices v_pk_* f32 beside MFMAs as an anti-lever), current with all 8 epilogue items of the nine-tap
# kernel in one round (build/u8)
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python -m pytest tests -m gpu -x -q -k "conv_fwd_dgrad or fp16_storage or tile_modes or nine_tap or patch_staged or persistent_patch or 32_row or subpixel or groupnorm_statistics or range_events" > gpurun_out/tests_r3k.log 2>&1; tail -1 gpurun_out/tests_r3k.log
B=$GRAFT_REPO_ROOT/build
( for pr in fp16 bf16; do for v in new u8 noslp; do
    if [ $v = new ]; then unset VQ_ABLATE_LIB; else export VQ_ABLATE_LIB=$B/$v/libvqhip_$v.so; fi
    echo "== $v $pr"; VQ_ITERS=30 timeout 100 python tools/bench_conv.py $pr 16 0,1,2,3 2>&1 | grep -v amdgpu.ids | sed 's/| wgrad.*//'
  done; done ) > gpurun_out/r3k_variants_micro.txt 2>&1
unset VQ_ABLATE_LIB
cat gpurun_out/r3k_variants_micro.txt
for k in "new 1" "prev 1" "noslp 1" "u8 1" "u8 2" "noslp 2" "prev 2" "new 2"; do set -- $k
  if [ $1 = new ]; then unset VQ_BENCH_AB_LIB; else export VQ_BENCH_AB_LIB=$B/$1/libvqhip_$1.so; fi
  timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 > gpurun_out/bench_r3k_$1_$2.json
  python - <<PY
import json
d = json.loads(open("gpurun_out/bench_r3k_$1_$2.json").read())
r = d["roofline"]
print("$1 rep $2:", d["value"], "img/s", d["ms_per_step"], "ms igemm", r["frac"], "conv3x3", r["conv3x3"]["frac"], "wgrad", r["wgrad"]["frac"])
PY
done 2>&1 | tee gpurun_out/r3k_bench_ab.txt
