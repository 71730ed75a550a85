#!/bin/bash
# Round 2, GPU call 3: full -m gpu suite (new full-size oracle tests), per-shape conv micro-benchmarks in bf16 / fp16 / zero data,
# SQ counters of the four biggest layer shapes, the two knob-only tile candidates.
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q -s -k "not full_size_step_is_finite" > gpurun_out/tests_r2c.log 2>&1
grep -E "configs0 parity|passed|failed" gpurun_out/tests_r2c.log | tail -8
for p in bf16 fp16; do timeout 200 python tools/bench_conv.py $p 16 > gpurun_out/micro_r2c_$p.log 2>&1; cat gpurun_out/micro_r2c_$p.log; done
VQ_ZERO=1 timeout 120 python tools/bench_conv.py bf16 16 5 > gpurun_out/micro_r2c_bf16_zero.log 2>&1; cat gpurun_out/micro_r2c_bf16_zero.log
VQ_TILE=5 timeout 120 python tools/bench_conv.py bf16 16 13 2>&1 | tail -1 > gpurun_out/micro_r2c_tile5.log; cat gpurun_out/micro_r2c_tile5.log
VQ_TILE=513 timeout 120 python tools/bench_conv.py bf16 16 9 > gpurun_out/micro_r2c_tile513.log 2>&1; cat gpurun_out/micro_r2c_tile513.log
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $GRAFT_REPO_ROOT/gpurun_out/pmc_r2c -o p -- \
    python $GRAFT_REPO_ROOT/tools/bench_conv.py bf16 16 4 > $GRAFT_REPO_ROOT/gpurun_out/pmc_r2c_run.log 2>&1 )
db=$(find gpurun_out/pmc_r2c -name "*.db" | head -1)
[ -n "$db" ] && python tools/pmc_dump.py "$db" > gpurun_out/pmc_r2c_sq.txt 2>&1
cat gpurun_out/pmc_r2c_sq.txt | head -80
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_BUSY_CYCLES -d $GRAFT_REPO_ROOT/gpurun_out/pmc_r2c2 -o p -- \
    python $GRAFT_REPO_ROOT/tools/bench_conv.py bf16 16 4 > $GRAFT_REPO_ROOT/gpurun_out/pmc_r2c2_run.log 2>&1 )
db=$(find gpurun_out/pmc_r2c2 -name "*.db" | head -1)
[ -n "$db" ] && python tools/pmc_dump.py "$db" > gpurun_out/pmc_r2c_sq2.txt 2>&1
cat gpurun_out/pmc_r2c_sq2.txt | head -80
rm -rf gpurun_out/pmc_r2c gpurun_out/pmc_r2c2
