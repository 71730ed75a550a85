set -x
# the author's launch_hdr.sh configuration (wavelet front-end, HR decoder, crop invariance, lecam, hinge, clamp), synthetic data
timeout 900 python -m vqgan_training_amd.vae_trainer --learning_rate_vae 0.0078125 --vae_ch 128 --run_name hdr_smoke --max_steps 7 \
  --evaluate_every_n_steps 3 --learning_rate_disc 3e-5 --batch_size 4 --do_clamp --do_ganloss --decoder_also_perform_hr True \
  --crop_invariance True --flip_invariance True --augment_before_perceptual_loss True --use_wavelet True --vae_z_channels 64 \
  --vae_ch_mult 1,2,4,4,4 --use_lecam True --disc_type hinge 2>&1 | tail -12
ls -la ckpt/hdr_smoke/ | tail -3
# resume from the checkpoint it wrote (reference format, module. prefix) + attention on
timeout 900 python -m vqgan_training_amd.vae_trainer --vae_ch 128 --run_name hdr_smoke2 --max_steps 2 --evaluate_every_n_steps 0 \
  --batch_size 2 --do_clamp --decoder_also_perform_hr True --use_wavelet True --vae_z_channels 64 --vae_ch_mult 1,2,4,4,4 \
  --load_path ckpt/hdr_smoke/vae_epoch_0_step_4.pt 2>&1 | tail -4
timeout 600 python -m vqgan_training_amd.vae_trainer --vae_ch 64 --run_name attn_smoke --max_steps 2 --evaluate_every_n_steps 0 \
  --batch_size 2 --do_attn True --vae_ch_mult 1,2,4,4 2>&1 | tail -3
