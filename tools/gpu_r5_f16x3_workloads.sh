set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for w in l1 l2 c2 c5; do
  timeout 600 python bench.py --workload $w --precision f16x3 --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > gpurun_out/r5i_f16x3_$w.log 2>&1
  echo "$w f16x3: $(grep -o '"value": [0-9.]*' gpurun_out/r5i_f16x3_$w.log | head -1) $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/r5i_f16x3_$w.log | head -1) $(tail -1 gpurun_out/r5i_f16x3_$w.log | cut -c1-200 | grep -v '^{')"
done
timeout 600 python -m vqgan_training_amd.vae_trainer --vae_ch 64 --vae_ch_mult 1,2,4 --batch_size 4 --do_ganloss --disc_type hinge --synthetic True --max_steps 6 --precision f16x3 --vae_resolution 128 > gpurun_out/r5i_cli_f16x3.log 2>&1; echo "cli rc=$?"; tail -3 gpurun_out/r5i_cli_f16x3.log | cut -c1-300
