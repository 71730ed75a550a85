#!/bin/bash
# the random sweeps of tools/fuzz_*.py on the GPU (product library): -> gpurun_out/<tag>_fuzz_*.txt
#   bash tools/gpu_fuzz.sh <tag> [step|model|all]
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; TAG=${1:-fuzz}; WHAT=${2:-all}
export FUZZ_DEVICE=cuda
if [ $WHAT != model ]; then
python tools/fuzz_step.py 40 2 fp32x6 2>&1 | grep "^ok\|^FAIL\|ok$" > gpurun_out/${TAG}_fuzz_step_fp32x6.txt
python tools/fuzz_step.py 40 3 f16x3 2>&1 | grep "^ok\|^FAIL\|ok$" > gpurun_out/${TAG}_fuzz_step_f16x3.txt
fi
if [ $WHAT != step ]; then
python tools/fuzz_model.py 60 5 f16x3 2>&1 | grep "^ok\|^FAIL\|ok$" > gpurun_out/${TAG}_fuzz_model_f16x3.txt
python tools/fuzz_model.py 40 6 fp32x6 2>&1 | grep "^ok\|^FAIL\|ok$" > gpurun_out/${TAG}_fuzz_model_fp32x6.txt
fi
tail -n 1 gpurun_out/${TAG}_fuzz_*.txt; grep -h "^FAIL" gpurun_out/${TAG}_fuzz_*.txt | cut -c1-400
