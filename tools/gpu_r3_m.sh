#!/bin/bash
# Round 3, GPU call M: is the discriminator stack's gradient picture after the timed run (fp16_after_run: tensors that are all zero, flushed
# waves) a property of the training dynamics or of the new kernels?  The same bench on the build from before the small-layer kernels
# and the epilogue rewrite (build/old) and on the current one.
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
for k in new old; do
  if [ $k = new ]; then unset VQ_BENCH_AB_LIB; else export VQ_BENCH_AB_LIB=$GRAFT_REPO_ROOT/build/old/libvqhip_old.so; fi
  VQ_TILE=512 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 > gpurun_out/bench_r3m_$k.json
  python - <<PY
import json
d = json.loads(open("gpurun_out/bench_r3m_$k.json").read())
c = d["config"]
print("$k", d["value"], "img/s", "final", c.get("final_losses"))
for s in c["fp16_after_run"]["stacks"]: print("   ", s)
PY
done 2>&1 | tee gpurun_out/r3m_after_run.txt
