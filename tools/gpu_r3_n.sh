#!/bin/bash
# Round 3, GPU call N: cycle stamps of the patch-staged 256x256 tile (one block per CU) and the nine-tap kernel after the epilogue work;
# second run: stamps inside the epilogue's two phases (30 + a: after row block a of the transposition, 40 + r: after round r)
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
( for pr in fp16 bf16; do for sh in 1 2 0; do timeout 60 python tools/stamps.py $pr $sh; done; done ) 2>&1 | grep -v amdgpu.ids > gpurun_out/r3n_stamps_last_block.txt
cat gpurun_out/r3n_stamps_last_block.txt
