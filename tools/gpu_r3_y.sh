#!/bin/bash
# Round 3, GPU call Y: stamps of the 256x256 tile's epilogue with its LDS transposition writes / its global stores removed (pricing hints)
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
( for v in 0 131104 131088; do VQ_TILE=$v timeout 60 python tools/stamps.py fp16 1; done ) 2>&1 | grep -v amdgpu.ids > gpurun_out/r3y_p9_epilogue_pricing_stamps.txt
cat gpurun_out/r3y_p9_epilogue_pricing_stamps.txt
