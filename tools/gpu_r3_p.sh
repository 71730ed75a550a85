#!/bin/bash
# Round 3, GPU call P: SQ counters of the patch-staged 256x256 tile with and without its epilogue (`make ablate` library, hint 8192 << 4):
# what the epilogue's cycles are made of (instruction issue by type, waits, LDS conflicts, instruction fetch)
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
A=$GRAFT_REPO_ROOT/build/ablate/libvqhip_ablate.so
rm -f gpurun_out/r3p_p9_sq.txt
for v in 0 131072; do
  for set in "SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES" "SQ_IFETCH SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM"; do
    ( cd /tmp && VQ_ABLATE_LIB=$A VQ_TILE=$v VQ_ITERS=5 timeout 200 rocprofv3 --kernel-trace --pmc $set -d $GRAFT_REPO_ROOT/gpurun_out/pmc_p9 -o p -- \
        python $GRAFT_REPO_ROOT/tools/bench_conv.py fp16 16 1,1 > $GRAFT_REPO_ROOT/gpurun_out/pmc_p9_run.log 2>&1 )
    db=$(find gpurun_out/pmc_p9 -name "*.db" | head -1)
    echo "== VQ_TILE=$v" >> gpurun_out/r3p_p9_sq.txt
    [ -n "$db" ] && python tools/pmc_dump.py "$db" 2>&1 | grep -E "conv_igemm_p9" >> gpurun_out/r3p_p9_sq.txt
    rm -rf gpurun_out/pmc_p9
  done
done
cat gpurun_out/r3p_p9_sq.txt
