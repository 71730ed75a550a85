import sys, os, logging
sys.path.insert(0, os.getcwd())
import torch, vqgan_training_amd as vq
logging.basicConfig(level=logging.WARNING)
# the reference's launcher.sh line (l1): vae_ch 64, HR decoder, bce GAN, clamp; 512x512 batches of 12
h = vq.vae_trainer.run_training(batch_size=12, do_ganloss=True, disc_type="bce", vae_resolution=256, vae_ch=64, vae_ch_mult="1,2,4,4",
                                vae_num_res_blocks=2, vae_z_channels=16, max_steps=40, evaluate_every_n_steps=20, precision=None,
                                log_every=5, run_name="soak", decoder_also_perform_hr=True, do_clamp=True, flip_invariance=True,
                                crop_invariance=False, augment_before_perceptual_loss=True)
print("l1 soak:", len(h), "log lines; dropped", sum(x.get("fp16/skipped_steps", 0) for x in h), "saturated", sum(x.get("fp16/saturated_waves", 0) for x in h),
      "final loss %.4f" % h[-1]["overall_vae_loss"])
# configs[4]-like: VQ quantizer, 512x512, 5 levels, batch 4
q = vq.quantizer.VectorQuantizer(16384, 32)
h = vq.vae_trainer.run_training(batch_size=4, do_ganloss=True, disc_type="hinge", vae_resolution=512, vae_ch=128, vae_ch_mult="1,2,4,4,4",
                                vae_num_res_blocks=2, vae_z_channels=32, max_steps=20, evaluate_every_n_steps=10, precision=None,
                                log_every=5, run_name="soak5", quantizer=q)
print("c5 soak:", len(h), "log lines; dropped", sum(x.get("fp16/skipped_steps", 0) for x in h), "saturated", sum(x.get("fp16/saturated_waves", 0) for x in h),
      "final loss %.4f" % h[-1]["overall_vae_loss"], "vq_loss %.4f" % h[-1].get("vq_loss", float("nan")))
