#!/bin/bash
# Round 3, GPU call O: stand-alone timing of the conv epilogue's transposition phase (tools/micro/epi_phase1.hip)
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -Wno-unused-value tools/micro/epi_phase1.hip -o /tmp/epi1 2>/dev/null
( /tmp/epi1; /tmp/epi1 ) 2>&1 | grep -v amdgpu.ids > gpurun_out/r3o_epi_phase1_mfma.txt
cat gpurun_out/r3o_epi_phase1_mfma.txt
