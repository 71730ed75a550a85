cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for wg in 0 4 260 132; do
  echo "== VQ_WGTILE=$wg" >> gpurun_out/r5k_wgrad_tiles_f16x3.txt
  VQ_WGTILE=$wg timeout 300 python tools/bench_conv.py f16x3 16 4 2>/dev/null | grep -v amdgpu >> gpurun_out/r5k_wgrad_tiles_f16x3.txt
done
cat gpurun_out/r5k_wgrad_tiles_f16x3.txt | sed 's/fwd.*| wgrad/wgrad/'
