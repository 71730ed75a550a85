"""The reference's own loop body and wrappers around THIS package's modules ("must survive DDP(vae, device_ids=[rank])").

/root/reference/vae_trainer.py:438,450 wraps the VAE and the PatchDiscriminator in torch.nn.parallel.DistributedDataParallel and then
 * calls `vae.module.encoder` / `vae.module.decoder` directly (:538,623-624) — the VAE's wrapper never sees a forward, so its
   gradients are NOT exchanged;
 * sends the discriminator through DDP.forward THREE times per step (:630-631 real / fake.detach(), :684 the generator term), with
   `total_d_loss.backward(retain_graph=True)` (:658) between the second and the third;
 * steps plain torch.optim.AdamW optimizers (:455-475) — no fused optimizer, no gradient sinks, no side stream.
`reference_step` restates exactly that sequence with the reference's function names from vqgan_training_amd.vae_trainer; tests run it
wrapped and unwrapped (tests/dist_worker.py mode "ddp": 2 ranks, gloo, emulator; tests/test_distributed.py: 1 rank, RCCL, MI355X).
"""
import torch

import vqgan_training_amd as vq
from oracle import weights as W


def build(device, res=16, ch=32):
    """Seeded modules + the reference's optimizers (vae_trainer.py:455-475: two VAE groups, wd 1e-3, betas .9/.95; D at 2e-4)."""
    vae = vq.ae.VAE(res, 3, ch, 3, [1, 2], 1, 4, False, False, False)
    vae.load_state_dict(W.randomize_state_dict(vae.state_dict(), 1))
    lp = vq.utils.LPIPS(pretrained_path=None)
    lp.load_state_dict(W.randomize_state_dict(lp.state_dict(), 2, relu_net=True))
    disc = vq.utils.PatchDiscriminator()
    disc.load_state_dict(W.randomize_state_dict(disc.state_dict(), 4, relu_net=True))
    disc.requires_grad_(True)                                           # vae_trainer.py:436
    return vae.to(device), lp.to(device).eval(), disc.to(device)


def optimizers(vae_w, disc_w, ch=32, lr_vae=1e-2, lr_disc=2e-4):
    opt_g = torch.optim.AdamW(
        [{"params": [p for n, p in vae_w.named_parameters() if "conv_in" not in n], "lr": lr_vae / ch},
         {"params": [p for n, p in vae_w.named_parameters() if "conv_in" in n], "lr": 1e-4}], weight_decay=1e-3, betas=(0.9, 0.95))
    opt_d = torch.optim.AdamW(disc_w.parameters(), lr=lr_disc, weight_decay=1e-3, betas=(0.9, 0.95))
    return opt_g, opt_d


def reference_step(vae_w, disc_w, lp, opt_g, opt_d, x, disc_type="hinge"):
    """One iteration of /root/reference/vae_trainer.py:538-703 (augmentations off).  `vae_w` / `disc_w`: DDP wrappers, or the bare
    modules (then `.module` is the module itself).  -> dict of losses and the gradients as the optimizers saw them."""
    T = vq.vae_trainer
    vae = vae_w.module if hasattr(vae_w, "module") else vae_w
    z = vae.encoder(x)                                                  # :538
    z_s = vae.reg(z)                                                    # :563
    reconstructed = vae.decoder(z_s)                                    # :623-624
    real_preds = disc_w(x)                                              # :630  (DDP.forward #1)
    fake_preds = disc_w(reconstructed.detach())                         # :631  (DDP.forward #2)
    d_loss, avg_real, avg_fake, acc = T.gan_disc_loss(real_preds, fake_preds, disc_type)     # :632-634
    avg_real = T.avg_scalar_over_nodes(avg_real, x.device)              # :636-637
    avg_fake = T.avg_scalar_over_nodes(avg_fake, x.device)
    total_d_loss = d_loss.mean()                                        # :647
    opt_d.zero_grad()                                                   # :657
    total_d_loss.backward(retain_graph=True)                            # :658
    d_grads = {n: p.grad.detach().clone() for n, p in (disc_w.module if hasattr(disc_w, "module") else disc_w).named_parameters()}
    opt_d.step()                                                        # :659
    recon_p = T.gradnorm(reconstructed)                                 # :662
    percep = lp(recon_p, x).mean()                                      # :676
    recon_for_mse = T.gradnorm(reconstructed, weight=0.001)             # :679
    vae_loss, _ = T.vae_loss_function(x, recon_for_mse, z)              # :680
    recon_for_gan = T.gradnorm(reconstructed, weight=1.0)               # :683
    fake2 = disc_w(recon_for_gan)                                       # :684  (DDP.forward #3)
    g_gan = -fake2.mean() if disc_type == "hinge" else torch.nn.functional.binary_cross_entropy_with_logits(fake2, torch.ones_like(fake2))
    overall = percep + g_gan + vae_loss                                 # :695
    overall.backward()                                                  # :701
    g_grads = {n: p.grad.detach().clone() for n, p in vae.named_parameters()}
    opt_g.step()                                                        # :702
    opt_g.zero_grad()                                                   # :703
    opt_d.zero_grad()                                                   # :708
    return {"d_loss": float(total_d_loss), "g_gan": float(g_gan), "overall": float(overall), "percep": float(percep),
            "avg_real": avg_real, "avg_fake": avg_fake, "d_grads": d_grads, "g_grads": g_grads}
