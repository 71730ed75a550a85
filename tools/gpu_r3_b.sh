#!/bin/bash
# Round 3, GPU call B: the -m gpu suite; three-tap weight gradient as a three-buffer ring with the barrier in mid-chunk (default) against its
# two-buffer form (VQ_WGTILE=16), per layer and whole step; the sub-pixel Upsample forward over the staged patch (default) against the
# one-tap 256x256 tile (VQ_TILE=8192 = dbg 512); closing bench line of the call with the conv table and kernel stats.
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/tests_r3b.log 2>&1; tail -3 gpurun_out/tests_r3b.log | head -2
( for rep in 1 2; do for v in 0 16; do echo "== VQ_WGTILE=$v rep $rep"; VQ_ITERS=30 VQ_WGTILE=$v timeout 100 python tools/bench_conv.py bf16 16 0,1,2,3,6,7 2>&1 | grep -v amdgpu.ids | sed 's/.*| wgrad/wgrad/'; done; done
  for v in 0 16; do echo "== fp16 VQ_WGTILE=$v"; VQ_ITERS=30 VQ_WGTILE=$v timeout 100 python tools/bench_conv.py fp16 16 0,1,2,3 2>&1 | grep -v amdgpu.ids | sed 's/.*| wgrad/wgrad/'; done ) > gpurun_out/r3b_wgrad_ring_micro.txt 2>&1
cat gpurun_out/r3b_wgrad_ring_micro.txt
( for rep in 1 2; do for v in 0 8192; do VQ_TILE=$v timeout 100 python tools/bench_subpix.py 16 2>&1 | grep -v amdgpu.ids | grep "VQ_TILE\|^up"; done; done ) > gpurun_out/r3b_subpix_patch_micro.txt 2>&1
cat gpurun_out/r3b_subpix_patch_micro.txt
for rep in 1 2; do for k in "0 0" "16 0" "0 8192"; do set -- $k
  VQ_WGTILE=$1 VQ_TILE=$2 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 > gpurun_out/bench_r3b_$1_$2_$rep.json
  python - <<PY
import json
d = json.loads(open("gpurun_out/bench_r3b_$1_$2_$rep.json").read())
print("VQ_WGTILE=$1 VQ_TILE=$2 rep $rep:", d["value"], "img/s", d["ms_per_step"], "ms conv3x3", d["roofline"]["conv3x3"]["frac"], "wgrad", d["roofline"]["wgrad"]["frac"], "dropped", d["config"]["fp16_after_run"]["optimizer_steps_dropped"])
PY
done; done 2>&1 | tee gpurun_out/r3b_bench_ab.txt
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --conv-table gpurun_out/conv_table_r3b.txt > gpurun_out/bench_r3b.log 2>&1
tail -1 gpurun_out/bench_r3b.log | cut -c1-2500
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r3b -o p -- \
    python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > $GRAFT_REPO_ROOT/gpurun_out/prof_r3b_run.log 2>&1 )
db=$(find gpurun_out/prof_r3b -name "*.db" | head -1)
[ -n "$db" ] && python tools/rocpd_stats.py "$db" gpurun_out/kernel_stats_r3b.csv > gpurun_out/kernel_stats_r3b.txt 2>&1
rm -rf gpurun_out/prof_r3b
head -30 gpurun_out/kernel_stats_r3b.csv | cut -c1-150
