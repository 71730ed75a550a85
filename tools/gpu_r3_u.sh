#!/bin/bash
# Round 3, GPU call U: instruction-cache capacity as a kernel sees it (tools/micro/icache.hip)
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -Wno-unused-value tools/micro/icache.hip -o /tmp/icache 2>/dev/null
timeout 120 /tmp/icache 2>&1 | grep -v amdgpu.ids > gpurun_out/r3u_icache.txt
cat gpurun_out/r3u_icache.txt
