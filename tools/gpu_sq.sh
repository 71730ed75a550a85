#!/bin/bash
# SQ counters of the kernels that carry the step (patch-staged 256x256 tile, nine-tap 128-row kernel, three-tap weight gradient) on
# THIS build, binary16 and VQ_F16X2 instantiations: separate --pmc passes with --kernel-trace only (MI355X_MICROARCH.md; gpurun
# refuses --pmc together with sys / hip traces).  -> gpurun_out/<tag>_sq_counters.txt (raw) and <tag>_sq_summary.json (per kernel:
# duration, MFMA-busy share, LDS bank-conflict share).  usage on the GPU box: bash tools/gpu_sq.sh <tag>
set -u
cd "$GRAFT_REPO_ROOT"
TAG="${1:-r5}"
mkdir -p gpurun_out
export TMPDIR=/tmp
RAW=gpurun_out/${TAG}_sq_counters.txt
rm -f $RAW
for pr in fp16 f16x3; do
  for set in "SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS" "SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD"; do
    ( cd /tmp && VQ_ITERS=5 timeout 200 rocprofv3 --kernel-trace --pmc $set -d $GRAFT_REPO_ROOT/gpurun_out/pmc_sq -o p -- \
        python $GRAFT_REPO_ROOT/tools/bench_conv.py $pr 16 0,1,2 > $GRAFT_REPO_ROOT/gpurun_out/pmc_sq_run.log 2>&1 )
    db=$(find gpurun_out/pmc_sq -name "*.db" | head -1)
    echo "== $pr" >> $RAW
    [ -n "$db" ] && python tools/pmc_dump.py "$db" 2>&1 | grep -E "conv_igemm_p9|conv_igemm_tap9|conv_wgrad3" >> $RAW
    rm -rf gpurun_out/pmc_sq
  done
done
python - "$TAG" <<'PY'
import json, re, sys
tag = sys.argv[1]
cur, rows = None, {}
for line in open(f"gpurun_out/{tag}_sq_counters.txt"):
    if line.startswith("=="):
        cur = line.split()[1]; continue
    m = re.match(r"\s+void (\w+)<([^>]*?)[>(].*?\s(SQ_\w+)\s+([\d.]+)\s+n=(\d+) dur=([\d.]+)us", line)
    if m:
        k = f"{m.group(1)}<{m.group(2).strip()}>"
        rows.setdefault((cur, k), {})[m.group(3)] = float(m.group(4)); rows[(cur, k)]["dur_us"] = float(m.group(6))
out = []
for (pr, k), c in sorted(rows.items()):
    e = {"precision": pr, "kernel": k, "dur_us_under_profiler": c.get("dur_us")}
    # per counter INSTANCE averages (pmc_dump.py): an instance = one shader engine of one XCD = 8 CUs = 32 SIMDs on this part, so
    # MFMA-busy share of the instance's SIMD cycles = SQ_VALU_MFMA_BUSY_CYCLES / (32 * SQ_BUSY_CYCLES)
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "SQ_BUSY_CYCLES" in c and c["SQ_BUSY_CYCLES"] > 0:
        e["mfma_busy"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (32.0 * c["SQ_BUSY_CYCLES"]), 4)
    if "SQ_LDS_BANK_CONFLICT" in c and c.get("SQ_LDS_IDX_ACTIVE"):
        e["lds_bank_conflict_share"] = round(c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"], 4)
    if "SQ_WAIT_INST_LDS" in c and c.get("SQ_WAVE_CYCLES"):
        e["wave_cycles_waiting_on_lds"] = round(c["SQ_WAIT_INST_LDS"] / c["SQ_WAVE_CYCLES"], 4)
    e["counters"] = {n: v for n, v in c.items() if n.startswith("SQ_")}
    out.append(e)
json.dump({"method": "rocprofv3 --kernel-trace --pmc <8 SQ counters> in two passes per precision over tools/bench_conv.py <precision> 16 0,1,2 "
                     "(128->128 @256^2, 256->256 @128^2, 512->512 @64^2; forward, data gradient, weight gradient; 5 iterations); values = "
                     "averages per counter instance and dispatch (tools/pmc_dump.py)", "kernels": out},
          open(f"gpurun_out/{tag}_sq_summary.json", "w"), indent=1)
for e in out:
    print(e["precision"], e["kernel"][:60], {k: v for k, v in e.items() if k not in ("counters", "kernel", "precision")})
PY
