#!/bin/bash
# Round 3, GPU call E: three kernels for the layers the tile kernels serve badly — conv_igemm_c64_kernel (64 -> 64, resident weights;
# VQ_TILE=512 = hint 32 << 4 = the nine-tap 64-row tile it replaces), the nine-tap kernel with 32-row tiles for <= 32 output
# channels (VQ_TILE=640 = 40 << 4 = the one-tap tile), conv_patch_dgrad_kernel (VQ_TILE=768 = 48 << 4 = the generic kernels).
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 420 python -m pytest tests -m gpu -x -q -k "resident_weight or 32_row or persistent_patch or full_size or adjoint or tile_modes or nine_tap or fp16_storage or conv_fwd_dgrad or trajectory or fuzz" > gpurun_out/tests_r3e.log 2>&1; tail -3 gpurun_out/tests_r3e.log | head -2
( for rep in 1 2; do for v in 0 512 640; do for pr in fp16 bf16; do echo "== VQ_TILE=$v $pr rep $rep"; VQ_ITERS=30 VQ_TILE=$v timeout 100 python tools/bench_conv.py $pr 16 12,10,15,11 2>&1 | grep -v amdgpu.ids | sed 's/| wgrad.*//'; done; done; done
  for v in 0 512; do echo "== VQ_TILE=$v fp16 B=32"; VQ_ITERS=30 VQ_TILE=$v timeout 100 python tools/bench_conv.py fp16 32 12 2>&1 | grep -v amdgpu.ids | sed 's/| wgrad.*//'; done ) > gpurun_out/r3e_small_layers_micro.txt 2>&1
cat gpurun_out/r3e_small_layers_micro.txt
( for v in 0 768; do echo "== VQ_TILE=$v"; VQ_TILE=$v timeout 100 python tools/bench_patch_dgrad.py 2>&1 | grep -v amdgpu.ids; done ) > gpurun_out/r3e_patch_dgrad_micro.txt 2>&1
cat gpurun_out/r3e_patch_dgrad_micro.txt
for k in "0 1" "512 1" "640 1" "768 1" "768 2" "640 2" "512 2" "0 2"; do set -- $k
  VQ_TILE=$1 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 > gpurun_out/bench_r3e_$1_$2.json
  python - <<PY
import json
d = json.loads(open("gpurun_out/bench_r3e_$1_$2.json").read())
r = d["roofline"]
print("VQ_TILE=$1 rep $2:", d["value"], "img/s", d["ms_per_step"], "ms igemm", r["frac"], "conv3x3", r["conv3x3"]["frac"], "wgrad", r["wgrad"]["frac"])
PY
done 2>&1 | tee gpurun_out/r3e_bench_ab.txt
timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --conv-table gpurun_out/r3e_conv_table_c3_ref.txt > gpurun_out/r3e_bench_c3_ref.json.log 2>&1
