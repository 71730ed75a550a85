#!/bin/bash
# One short GPU call: hardware parity of the phase-decomposed resampling convs, then the bench with the path on / off.
# Results land in gpurun_out/ step by step so that a cut-off call still leaves what it finished.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 210 python -m pytest tests/test_kernels.py -m gpu -x -q \
    -k "subpixel or adjoint or large_shapes or (fwd_dgrad_wgrad and 3-1-1-2) or (fwd_dgrad_wgrad and 3-2-0-1) or groupnorm_properties" \
    > gpurun_out/sub_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/sub_tests.log ) 
tail -3 gpurun_out/sub_tests.log
( VQ_SUBPIXEL=1 timeout 100 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --conv-table gpurun_out/conv_table_sub_on.txt \
    > gpurun_out/bench_sub_on.log 2>&1; echo "rc=$?" >> gpurun_out/bench_sub_on.log )
tail -2 gpurun_out/bench_sub_on.log | cut -c1-400
( VQ_SUBPIXEL=0 timeout 100 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --conv-table gpurun_out/conv_table_sub_off.txt \
    > gpurun_out/bench_sub_off.log 2>&1; echo "rc=$?" >> gpurun_out/bench_sub_off.log )
tail -2 gpurun_out/bench_sub_off.log | cut -c1-400
( timeout 60 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "rc=$?" >> gpurun_out/smoke.log )
tail -2 gpurun_out/smoke.log
( cd /tmp && timeout 150 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -o r1v30 -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_run.log 2>&1
  db=$(ls $GRAFT_REPO_ROOT/gpurun_out/prof/*/*.db $GRAFT_REPO_ROOT/gpurun_out/prof/*.db 2>/dev/null | head -1)
  [ -n "$db" ] && python $GRAFT_REPO_ROOT/tools/rocpd_stats.py "$db" $GRAFT_REPO_ROOT/gpurun_out/kernel_stats_v30.csv > $GRAFT_REPO_ROOT/gpurun_out/kernel_stats_v30.txt 2>&1
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof )
head -12 gpurun_out/kernel_stats_v30.txt
