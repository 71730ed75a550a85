#!/bin/bash
# Round 3, GPU call Y: stamps of the 256x256 tile's epilogue with its LDS transposition writes / its global stores removed (pricing hints),
# and (second run) with the main loop removed (hint 8210 << 4): does the epilogue run faster when no MFMA loop precedes it?
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
( for v in 0 131360; do VQ_TILE=$v timeout 60 python tools/stamps.py fp16 1; VQ_TILE=$v timeout 60 python tools/stamps.py bf16 1; done ) 2>&1 | grep -v amdgpu.ids > gpurun_out/r3y_p9_epilogue_without_main_loop.txt
cat gpurun_out/r3y_p9_epilogue_without_main_loop.txt
