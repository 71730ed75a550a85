#!/bin/bash
# Round 2, GPU call 26: SQ counters of the patch-staged 256x256 tile (default) and the one-tap tile (VQ_TILE=8192) on the same two layers
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
for v in 0 8192; do
  ( cd /tmp && VQ_TILE=$v VQ_ITERS=5 timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS -d $GRAFT_REPO_ROOT/gpurun_out/pmc_p9_$v -o p -- \
      python $GRAFT_REPO_ROOT/tools/bench_conv.py bf16 16 1,2 > $GRAFT_REPO_ROOT/gpurun_out/pmc_p9_run_$v.log 2>&1 )
  db=$(find gpurun_out/pmc_p9_$v -name "*.db" | head -1)
  echo "== VQ_TILE=$v" >> gpurun_out/pmc_p9_sq.txt
  [ -n "$db" ] && python tools/pmc_dump.py "$db" 2>&1 | grep -E "conv_igemm_p9|conv_igemm_glds_kernel<0, 256" >> gpurun_out/pmc_p9_sq.txt
  rm -rf gpurun_out/pmc_p9_$v
done
cat gpurun_out/pmc_p9_sq.txt
