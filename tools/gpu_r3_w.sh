#!/bin/bash
# Round 3, GPU call W: __graft_entry__.smoke() on the final build
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r3w_smoke.log 2>&1; tail -3 gpurun_out/r3w_smoke.log
