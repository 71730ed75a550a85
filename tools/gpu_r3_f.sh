#!/bin/bash
# Round 3, GPU call F: where does conv_igemm_c64_kernel's time go?  It ran no faster than the nine-tap 64-row tile it was meant to
# beat (call E).  Ablations of it (tools-only `make ablate` library: epilogue skipped, no halo DMA after the first patch, no MFMAs),
# the same on the nine-tap tile, SQ counters of both, and the in-step A/B again with the patch location hoisted out of the DMA pieces.
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 200 python -m pytest tests -m gpu -x -q -k "resident_weight or 32_row or persistent_patch" > gpurun_out/tests_r3f.log 2>&1; tail -1 gpurun_out/tests_r3f.log
A=$GRAFT_REPO_ROOT/build/ablate/libvqhip_ablate.so
( for pr in fp16 bf16; do
    for v in 0 131072 131200 131216 131232 131248 5 131077 512; do
      echo "== $pr VQ_TILE=$v"; VQ_ABLATE_LIB=$A VQ_ITERS=30 VQ_TILE=$v timeout 100 python tools/bench_conv.py $pr 16 12,12 2>&1 | grep -v amdgpu.ids | sed 's/| wgrad.*//' | head -1
    done
  done
  echo "== 128 ch for reference: default, epilogue skipped"
  for v in 0 131072; do VQ_ABLATE_LIB=$A VQ_ITERS=30 VQ_TILE=$v timeout 100 python tools/bench_conv.py bf16 16 0,0 2>&1 | grep -v amdgpu.ids | sed 's/| wgrad.*//' | head -1; done ) > gpurun_out/r3f_c64_ablations.txt 2>&1
cat gpurun_out/r3f_c64_ablations.txt
rm -f gpurun_out/r3f_c64_sq.txt
for v in 0 512; do
  for set in "SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS" "SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"; do
    ( cd /tmp && VQ_TILE=$v VQ_ITERS=5 timeout 200 rocprofv3 --kernel-trace --pmc $set -d $GRAFT_REPO_ROOT/gpurun_out/pmc_c64 -o p -- \
        python $GRAFT_REPO_ROOT/tools/bench_conv.py fp16 16 12,12 > $GRAFT_REPO_ROOT/gpurun_out/pmc_c64_run.log 2>&1 )
    db=$(find gpurun_out/pmc_c64 -name "*.db" | head -1)
    echo "== VQ_TILE=$v" >> gpurun_out/r3f_c64_sq.txt
    [ -n "$db" ] && python tools/pmc_dump.py "$db" 2>&1 | grep -E "conv_igemm_c64|conv_igemm_tap9" >> gpurun_out/r3f_c64_sq.txt
    rm -rf gpurun_out/pmc_c64
  done
done
cat gpurun_out/r3f_c64_sq.txt
for k in "0 1" "512 1" "512 2" "0 2"; do set -- $k
  VQ_TILE=$1 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 > gpurun_out/bench_r3f_$1_$2.json
  python - <<PY
import json
d = json.loads(open("gpurun_out/bench_r3f_$1_$2.json").read())
r = d["roofline"]
print("VQ_TILE=$1 rep $2:", d["value"], "img/s", d["ms_per_step"], "ms igemm", r["frac"], "conv3x3", r["conv3x3"]["frac"], "wgrad", r["wgrad"]["frac"])
PY
done 2>&1 | tee gpurun_out/r3f_bench_ab.txt
