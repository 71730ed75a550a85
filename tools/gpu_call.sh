#!/bin/bash
# ONE parameterised GPU call (replaces the per-call one-shot scripts of rounds 1-3): a tag and a list of stages, run in order on the
# GPU box from the repo root; everything lands in gpurun_out/<tag>_*.  Typical use:
#   gpurun --timeout 1500 -- 'bash tools/gpu_call.sh r4a suite smoke bench stats'
# stages
#   suite            python -m pytest tests -m gpu -q            (no -x: every failure is listed)      -> <tag>_gpu_suite.log
#   suitex           the driver's form: -x -q                                                           -> <tag>_gpu_suite_x.log
#   k:<expr>         pytest -m gpu -k <expr>                                                            -> <tag>_tests.log
#   smoke / smoke!   __graft_entry__.smoke() (smoke!: stop the call when it fails)
#   bench            the driver's default line (python bench.py) + per-layer table                      -> <tag>_bench_c3.json.log, <tag>_conv_table_c3.txt
#   bench20          bench.py --steps 20 --warmup 5 (the driver's round-end flags)
#   quickp:<policy>  bench.py --steps 6 --warmup 2 --precision <policy> --no-cpu-baseline --no-secondary + table -> <tag>_quick_<policy>.json.log
#   quick            bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary + table            -> <tag>_quick.json.log
#   bench_c2 / bench_c5 / bench_l1 / bench_l2   the other workloads (l1, l2: the reference's own launch lines)
#   stats / stats_overlap   rocprofv3 --kernel-trace --stats of bench.py --steps 6 --warmup 2, one stream / as timed -> <tag>_kernel_stats[_overlap].csv / .txt
#   sq               tools/gpu_sq.sh: SQ counters (MFMA busy, LDS conflicts) of the top kernels, fp16 + f16x3 -> <tag>_sq_counters.txt, <tag>_sq_summary.json
#   traffic          tools/gpu_traffic.sh (separate --pmc passes) -> profiles/<tag>_traffic.json
#   ab:<ENVVAR>[=a,b]  bench quick with ENVVAR=1 / 0 (or the listed values), twice each, interleaved, same box ($BENCH_ARGS: extra bench.py flags) -> <tag>_ab_<ENVVAR>.txt
#   ablib:<path>     bench quick with VQ_BENCH_AB_LIB=<path> vs the in-tree library, twice each, interleaved
#   py:<script.py>[,arg,...]   python <script> args                                                     -> <tag>_<script>.txt
set -u
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG="$1"; shift
O=gpurun_out/$TAG
line() { grep -o '"value": [0-9.]*' "$1" | head -1; grep -o '"ms_per_step": [0-9.]*' "$1" | head -1; grep -o '"peak_hbm_allocated_GB": [0-9.]*' "$1" | head -1; }
for st in "$@"; do
  echo "=== stage $st"
  case "$st" in
    suite)   timeout 1500 python -m pytest tests -m gpu -q > ${O}_gpu_suite.log 2>&1; grep -n "passed\|failed\|error" ${O}_gpu_suite.log | tail -5 ;;
    suitex)  timeout 1500 python -m pytest tests -m gpu -x -q > ${O}_gpu_suite_x.log 2>&1; tail -3 ${O}_gpu_suite_x.log ;;
    k:*)     timeout 900 python -m pytest tests -m gpu -q -s -k "${st#k:}" > ${O}_tests.log 2>&1; grep -n "passed\|failed\|parity\|error" ${O}_tests.log | tail -12 ;;
    smoke|smoke!)   timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > ${O}_smoke.log 2>&1; tail -1 ${O}_smoke.log
             if [ "$st" = "smoke!" ] && ! grep -q "smoke ok" ${O}_smoke.log; then echo "smoke failed: stopping the call"; tail -20 ${O}_smoke.log; exit 1; fi ;;
    bench)   timeout 1500 python bench.py --conv-table ${O}_conv_table_c3.txt > ${O}_bench_c3.json.log 2>&1; tail -1 ${O}_bench_c3.json.log | cut -c1-500 ;;
    bench20) timeout 1500 python bench.py --steps 20 --warmup 5 --conv-table ${O}_conv_table_c3.txt > ${O}_bench_c3.json.log 2>&1; tail -1 ${O}_bench_c3.json.log | cut -c1-500 ;;
    quick)   timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --conv-table ${O}_conv_table_quick.txt > ${O}_quick.json.log 2>&1; tail -1 ${O}_quick.json.log | cut -c1-400 ;;
    quickp:*) P="${st#quickp:}"; timeout 900 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-secondary --precision $P --conv-table ${O}_conv_table_$P.txt > ${O}_quick_$P.json.log 2>&1; tail -1 ${O}_quick_$P.json.log | cut -c1-600 ;;
    bench_c2) timeout 600 python bench.py --workload c2 --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > ${O}_bench_c2.json.log 2>&1; tail -1 ${O}_bench_c2.json.log | cut -c1-200 ;;
    bench_l1|bench_l2) W=${st#bench_}; timeout 600 python bench.py --workload $W --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --conv-table ${O}_conv_table_$W.txt > ${O}_bench_$W.json.log 2>&1; tail -1 ${O}_bench_$W.json.log | cut -c1-400 ;;
    bench_c5) timeout 900 python bench.py --workload c5 --steps 6 --warmup 2 > ${O}_bench_c5.json.log 2>&1; tail -1 ${O}_bench_c5.json.log | cut -c1-300 ;;
    stats|stats_overlap)
             # stats: ONE stream (VQ_WGRAD_OVERLAP=0) — every kernel alone on the chip, the durations roofline.* must agree with;
             # stats_overlap: the step as it is timed (weight gradients on the side stream): totals only, per-kernel durations include co-runners
             if [ "$st" = stats ]; then OV=0; SUF=""; else OV=1; SUF="_overlap"; fi
             ( cd /tmp && env VQ_WGRAD_OVERLAP=$OV timeout 300 rocprofv3 --kernel-trace --stats -d $OLDPWD/gpurun_out/prof_$TAG -o p -- python $OLDPWD/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-secondary --no-serial-pass > $OLDPWD/${O}_prof_run$SUF.log 2>&1 )
             db=$(find gpurun_out/prof_$TAG -name "*.db" | head -1)
             [ -n "$db" ] && python tools/rocpd_stats.py "$db" ${O}_kernel_stats$SUF.csv > ${O}_kernel_stats$SUF.txt 2>&1
             [ -n "$db" ] && python tools/rocpd_busy.py "$db" > ${O}_gpu_busy$SUF.txt 2>&1
             rm -rf gpurun_out/prof_$TAG; head -16 ${O}_kernel_stats$SUF.txt; cat ${O}_gpu_busy$SUF.txt ;;
    sq)      bash tools/gpu_sq.sh $TAG > ${O}_sq_run.log 2>&1; tail -8 ${O}_sq_run.log | cut -c1-260 ;;
    traffic) bash tools/gpu_traffic.sh $TAG ref > ${O}_traffic_run.log 2>&1; tail -3 ${O}_traffic_run.log
             cp gpurun_out/traffic_$TAG.json profiles/${TAG}_traffic.json 2>/dev/null; cp gpurun_out/traffic_$TAG.json ${O}_traffic.json 2>/dev/null ;;
    ab:*)    V="${st#ab:}"; VALS="1 0"
             if [[ "$V" == *=* ]]; then VALS="${V#*=}"; VALS="${VALS//,/ }"; V="${V%%=*}"; fi     # ab:VAR=a,b: the two (or more) values to alternate
             : > ${O}_ab_$V.txt
             for rep in 1 2; do for v in $VALS; do
               env $V=$v timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary ${BENCH_ARGS:-} > ${O}_ab_${V}_${v}_$rep.log 2>&1
               echo "$V=$v rep $rep: $(line ${O}_ab_${V}_${v}_$rep.log | tr '\n' ' ') $(grep -o '"conv3x3": {[^}]*}' ${O}_ab_${V}_${v}_$rep.log | cut -c1-90) $(grep -o '"wgrad": {[^}]*}' ${O}_ab_${V}_${v}_$rep.log | cut -c1-70)" | tee -a ${O}_ab_$V.txt
             done; done ;;
    ablib:*) L="${st#ablib:}"; N=$(basename $L .so); : > ${O}_ablib_$N.txt
             for rep in 1 2; do for v in other tree; do
               if [ $v = other ]; then export VQ_BENCH_AB_LIB=$PWD/$L; else unset VQ_BENCH_AB_LIB; fi
               timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary ${BENCH_ARGS:-} > ${O}_ablib_${N}_${v}_$rep.log 2>&1
               echo "$v rep $rep: $(line ${O}_ablib_${N}_${v}_$rep.log | tr '\n' ' ') $(grep -o '"conv3x3": {[^}]*}' ${O}_ablib_${N}_${v}_$rep.log | cut -c1-90) $(grep -o '"wgrad": {[^}]*}' ${O}_ablib_${N}_${v}_$rep.log | cut -c1-70)" | tee -a ${O}_ablib_$N.txt
             done; done; unset VQ_BENCH_AB_LIB ;;
    py:*)    IFS=, read -r -a A <<< "${st#py:}"; N=$(basename ${A[0]} .py)
             timeout 900 python "${A[@]}" > ${O}_$N.txt 2>&1; tail -25 ${O}_$N.txt ;;
    *)       echo "unknown stage $st" ;;
  esac
done
