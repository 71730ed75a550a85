#!/bin/bash
# Round 3: SQ counters of the four kernels that make 55 % of the step (patch-staged 256x256 tile, nine-tap 128-row kernel, three-tap weight
# gradient), final build, for the next round's analysis: per-XCD-instance averages (tools/pmc_dump.py)
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/r3_final_sq_counters.txt
for pr in fp16 bf16; do
  for set in "SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS" "SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD"; do
    ( cd /tmp && VQ_ITERS=5 timeout 200 rocprofv3 --kernel-trace --pmc $set -d $GRAFT_REPO_ROOT/gpurun_out/pmc_fin -o p -- \
        python $GRAFT_REPO_ROOT/tools/bench_conv.py $pr 16 0,1,2 > $GRAFT_REPO_ROOT/gpurun_out/pmc_fin_run.log 2>&1 )
    db=$(find gpurun_out/pmc_fin -name "*.db" | head -1)
    echo "== $pr" >> gpurun_out/r3_final_sq_counters.txt
    [ -n "$db" ] && python tools/pmc_dump.py "$db" 2>&1 | grep -E "conv_igemm_p9|conv_igemm_tap9|conv_wgrad3" >> gpurun_out/r3_final_sq_counters.txt
    rm -rf gpurun_out/pmc_fin
  done
done
wc -l gpurun_out/r3_final_sq_counters.txt
