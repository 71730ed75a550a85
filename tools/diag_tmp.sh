cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for v in 1 1 1; do
  VQ_WGRAD_OVERLAP=$v timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --parity-mode-steps 3 > gpurun_out/r4e_diag_$v.log 2>&1
  python - <<PY
import json
d=json.loads(open("gpurun_out/r4e_diag_$v.log").read().strip().splitlines()[-1])
print("overlap=$v value", d["value"], "bf16", d["bf16_mode"]["value"], {k:v["value"] for k,v in d["parity_mode"].items() if isinstance(v,dict)})
PY
done
