#!/bin/bash
# HBM traffic of the bench's dominant kernel family from the PMC counters, as MI355X_MICROARCH.md prescribes: separate
# --pmc passes (FETCH_SIZE, WRITE_SIZE), --kernel-trace only (gpurun refuses --pmc together with sys / hip / memory-copy traces).
# Writes gpurun_out/traffic_<tag>.txt: per kernel name launches, average FETCH (x2 gfx950 correction for 16 B/lane streaming
# reads) and WRITE bytes per launch, and the implicit-GEMM family's per-launch average to put beside roofline.achieved.
# usage (on the GPU box, from the repo root): bash tools/gpu_traffic.sh <tag>
set -u
TAG="${1:-r2}"
PREC="${2:-bf16}"
mkdir -p gpurun_out
export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 280 rocprofv3 --kernel-trace --pmc $c -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$c -o p -- \
      env VQ_WGRAD_OVERLAP=0 python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-calibrate --no-serial-pass --precision $PREC > $GRAFT_REPO_ROOT/gpurun_out/pmc_run_$c.log 2>&1 )
done
python - "$TAG" "$PREC" <<'PY'
import glob, json, os, sqlite3, sys
tag, prec = sys.argv[1], sys.argv[2]
root = os.environ.get("GRAFT_REPO_ROOT", ".")
sys.path.insert(0, root)
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    dbs = glob.glob(f"{root}/gpurun_out/pmc_{c}/**/*.db", recursive=True)
    if not dbs:
        print("no db for", c); continue
    cur = sqlite3.connect(dbs[0]).cursor()
    for name, val, cnt in cur.execute("select name, avg(counter_value), count(*) from pmc_events where counter_name = ? group by name", (c,)):
        out.setdefault(name, {})[c] = (val, cnt)
lines, fam = [], [0, 0.0, 0.0]
for name, d in sorted(out.items(), key=lambda kv: -kv[1].get("FETCH_SIZE", (0, 0))[0] * kv[1].get("FETCH_SIZE", (0, 0))[1]):
    f, n = d.get("FETCH_SIZE", (0.0, 0)); w, _ = d.get("WRITE_SIZE", (0.0, 0))
    fb, wb = f * 1024 * 2, w * 1024            # counters are KiB per dispatch; FETCH_SIZE x2 on gfx950 for 16 B/lane streaming reads
    lines.append(f"{name[:70]:70s} n={n:6d} fetch {fb / 1e6:10.2f} MB  write {wb / 1e6:10.2f} MB per launch")
    if "conv_igemm" in name:
        fam[0] += n; fam[1] += fb * n; fam[2] += wb * n
with open(f"{root}/gpurun_out/traffic_{tag}.txt", "w") as fh:
    fh.write("\n".join(lines) + "\n")
    if fam[0]:
        fh.write(f"# implicit-GEMM family: {fam[0]} launches, {(fam[1] + fam[2]) / fam[0] / 1e6:.2f} MB of HBM traffic per launch "
                 f"(fetch {fam[1] / fam[0] / 1e6:.2f} + write {fam[2] / fam[0] / 1e6:.2f})\n")
# HBM-bound families as bench.py names them (ops._launch "hbm:<family>") -> the kernels that serve them.  The run is
# bench.py --steps 3 --warmup 1 --no-calibrate: exactly 4 steps, nothing else on the GPU.
STEPS = 4
FAMILIES = {"gn_stats": ("gn_reduce_kernel<0, 0, 0>", "gn_reduce_kernel<2, 0, 0>", "gn_reduce_kernel<1, 0, 0>", "gn_stats_finalize"),
            "gn_apply": ("gn_apply_kernel",),
            "gn_bwd": ("gn_reduce_kernel<0, 1", "gn_reduce_kernel<2, 1", "gn_reduce_kernel<1, 1", "gn_bwd_apply_kernel", "gn_bwd_finalize", "gn_bwd_coef"),
            "maxpool": ("pool2_kernel",), "adamw": ("adamw_multi_kernel",), "lpips_tap": ("lpips_tap_kernel", "lpips_finalize"),
            "weight_pack": ("pack_weight_multi_kernel", "pack_amax_multi_kernel", "pack_zero"),
            "colsum": ("colsum_kernel", "colsum_finalize"), "layout": ("nchw_to_nhwc_kernel", "nhwc_to_nchw_kernel"),
            "gradnorm": ("sumsq_kernel", "l2norm"), "wgrad_reduce": ("wgrad_reduce",)}
fams = {}
for name, d in out.items():
    f, n = d.get("FETCH_SIZE", (0.0, 0)); w, _ = d.get("WRITE_SIZE", (0.0, 0))
    for famname, pats in FAMILIES.items():
        if any(pt in name for pt in pats):
            e = fams.setdefault(famname, {"bytes_per_step": 0.0, "launches_per_step": 0.0})
            e["bytes_per_step"] += (f * 1024 * 2 + w * 1024) * n / STEPS
            e["launches_per_step"] += n / STEPS
if fam[0]:      # what bench.py puts on the line as roofline.traffic (copy to profiles/<round>_traffic.json)
    with open(f"{root}/gpurun_out/traffic_{tag}.json", "w") as fh:
        json.dump({"igemm_family_bytes_per_launch": round((fam[1] + fam[2]) / fam[0]), "fetch_bytes_per_launch": round(fam[1] / fam[0]),
                   "write_bytes_per_launch": round(fam[2] / fam[0]), "launches": fam[0], "precision": prec, "workload": "c3", "steps_profiled": STEPS,
                   "sources_sha": __import__("importlib").import_module("bench").kernel_sources_sha(),   # bench.py only trusts a file measured on ITS kernel sources
                   "families": {k: {"bytes_per_step": round(v["bytes_per_step"]), "launches_per_step": round(v["launches_per_step"], 1)}
                                for k, v in sorted(fams.items())},
                   "method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE in separate passes over bench.py --steps 3 --warmup 1 "
                             "--no-calibrate (4 steps); FETCH_SIZE x2 (gfx950: 16 B/lane streaming reads are tallied at half, "
                             "MI355X_MICROARCH.md §HBM); WRITE_SIZE as reported; families = sums over the kernels serving each "
                             "HBM-bound call family of bench.py's `hbm` list, per step"}, fh, indent=1)
print(open(f"{root}/gpurun_out/traffic_{tag}.txt").read()[-1500:])
PY
rm -rf gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE
