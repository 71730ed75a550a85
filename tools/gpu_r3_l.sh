#!/bin/bash
# Round 3, GPU call L: the full -m gpu suite on the current build; cycle stamps of the three-tap weight-gradient kernel
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
( for pr in bf16 fp16; do
    timeout 60 python tools/stamps_wgrad.py $pr 128 128 256; timeout 60 python tools/stamps_wgrad.py $pr 256 256 128
    timeout 60 python tools/stamps_wgrad.py $pr 512 512 64; timeout 60 python tools/stamps_wgrad.py $pr 512 512 32
  done ) 2>&1 | grep -v amdgpu.ids > gpurun_out/r3l_wgrad_stamps.txt
cat gpurun_out/r3l_wgrad_stamps.txt
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/tests_r3l_full.log 2>&1; tail -3 gpurun_out/tests_r3l_full.log
