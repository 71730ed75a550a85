"""Round-6 evidence for VERDICT r5 weak 8: does a real `run_training` run drop optimizer steps where the reference never does?

configs[2] (vae_ch=128, 1,2,4,4, batch 16, 256x256, hinge GAN, synthetic batches), policy `ref`, N steps from a fresh discriminator —
the regime in which the generator's GAN gradient grows by orders of magnitude (bench.py's rehearsal exists because of it) — twice:
with the headroom-event path of round 6 (`relax_hot_scales` at every log line) and with it disabled.  Prints per log line the
dropped steps, saturated / headroom waves and loss scales, and the totals.

    python tools/train_probe.py [steps=60] [log_every=5]
"""
import logging
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import vqgan_training_amd as vq  # noqa: E402
from vqgan_training_amd import ops  # noqa: E402
from vqgan_training_amd.vae_trainer import VAETrainStep  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    log_every = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    logging.basicConfig(level=logging.WARNING)
    real_relax = VAETrainStep.relax_hot_scales
    for mode in ("headroom events on (round 6)", "headroom events ignored (rounds 3-5)"):
        relaxed_log = []
        if mode.startswith("headroom events on"):
            def spy(self, polled, _log=relaxed_log):
                moved = real_relax(self, polled)
                _log.append(moved)
                return moved
            VAETrainStep.relax_hot_scales = spy
        else:
            VAETrainStep.relax_hot_scales = lambda self, polled: []
        ops.clear_caches()
        torch.cuda.empty_cache()
        hist = vq.vae_trainer.run_training(batch_size=16, do_ganloss=True, disc_type="hinge", vae_resolution=256, vae_ch=128, vae_ch_mult="1,2,4,4",
                                           vae_num_res_blocks=2, vae_z_channels=16, max_steps=steps, evaluate_every_n_steps=0, precision="ref",
                                           log_every=log_every, run_name="probe")
        dropped = sum(h.get("fp16/skipped_steps", 0) for h in hist)
        sat = sum(h.get("fp16/saturated_waves", 0) for h in hist)
        print(f"== {mode}: {steps} steps, log every {log_every}: optimizer steps dropped on the device {dropped}, saturated gradient waves {sat}, "
              f"final overall_vae_loss {hist[-1]['overall_vae_loss']:.4f}, g_gan {hist[-1].get('gan/generator_gan_loss', float('nan')):.3f}")
        for i, h in enumerate(hist):
            moved = relaxed_log[i] if i < len(relaxed_log) else []
            print(f"   log {i:2d}: dropped {h.get('fp16/skipped_steps', 0)} saturated {h.get('fp16/saturated_waves', 0)} flushed {h.get('fp16/flushed_waves', 0)} "
                  f"g_gan {h.get('gan/generator_gan_loss', float('nan')):8.3f} d_loss {h.get('gan/discriminator_loss', float('nan')):6.3f}"
                  + (f"  scales lowered: {', '.join(f'{r}=2^{s:.0f}' for r, s in moved)}" if moved else ""))
    VAETrainStep.relax_hot_scales = real_relax


if __name__ == "__main__":
    main()
