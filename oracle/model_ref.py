"""ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/ops_ref.py for the rules).

Functional CPU fp32 restatement of the reference's model + loss + optimizer step, driven by plain
state dicts (name -> tensor, the reference's own key names), so the same weights can be fed to the
reference modules (oracle/reference_import.py, build container only), to this restatement, and to the
HIP modules.  Citations are file:line in /root/reference.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

from . import ops_ref as R

VGG_IDX = ((0, 2), (5, 7), (10, 12, 14), (17, 19, 21), (24, 26, 28))   # torchvision vgg16.features conv indices
LPIPS_CHNS = (64, 128, 256, 512, 512)


# ------------------------------------------------------------------------------------------ VAE
def _levels(p, prefix):
    n = 0
    while f"{prefix}.{n}.block.0.norm1.weight" in p:
        n += 1
    return n


def _blocks(p, prefix):
    n = 0
    while f"{prefix}.{n}.norm1.weight" in p:
        n += 1
    return n


def encoder(p, x, pre="encoder."):
    """ae.py:239-257; the wavelet front-end (ae.py:189-194,240) is recognised by conv_in taking 4x the image channels."""
    if p[pre + "conv_in.weight"].shape[1] == 4 * x.shape[1]:
        x = R.wavelet_transform(x)
    h = R.conv2d(x, p[pre + "conv_in.weight"], p[pre + "conv_in.bias"], padding=1)
    nl = _levels(p, pre + "down")
    for lvl in range(nl):
        for b in range(_blocks(p, f"{pre}down.{lvl}.block")):
            h = R.resnet_block(h, p, f"{pre}down.{lvl}.block.{b}.")
        if f"{pre}down.{lvl}.downsample.conv.weight" in p:
            h = R.downsample(h, p[f"{pre}down.{lvl}.downsample.conv.weight"], p[f"{pre}down.{lvl}.downsample.conv.bias"])
    h = _middle(p, h, pre)
    h = R.swish(R.group_norm_fp32(h, p[pre + "norm_out.weight"], p[pre + "norm_out.bias"]))
    return R.conv2d(h, p[pre + "conv_out.weight"], p[pre + "conv_out.bias"], padding=1)


def _middle(p, h, pre):
    """mid.block_1 -> mid.attn_1 (AttnBlock when use_attn, else Identity: ae.py:224-226, 250-252) -> mid.block_2."""
    h = R.resnet_block(h, p, pre + "mid.block_1.")
    if pre + "mid.attn_1.qkv.weight" in p:
        h = R.attn_block(h, p, pre + "mid.attn_1.")
    return R.resnet_block(h, p, pre + "mid.block_2.")


def decoder(p, z, pre="decoder."):
    """ae.py:318-333."""
    h = R.conv2d(z, p[pre + "conv_in.weight"], p[pre + "conv_in.bias"], padding=1)
    h = _middle(p, h, pre)
    nl = _levels(p, pre + "up")
    for lvl in reversed(range(nl)):
        for b in range(_blocks(p, f"{pre}up.{lvl}.block")):
            h = R.resnet_block(h, p, f"{pre}up.{lvl}.block.{b}.")
        if f"{pre}up.{lvl}.upsample.conv.weight" in p:
            h = R.upsample(h, p[f"{pre}up.{lvl}.upsample.conv.weight"], p[f"{pre}up.{lvl}.upsample.conv.bias"])
    h = R.swish(R.group_norm_fp32(h, p[pre + "norm_out.weight"], p[pre + "norm_out.bias"]))
    return R.conv2d(h, p[pre + "conv_out.weight"], p[pre + "conv_out.bias"], padding=1)


def vae_forward(p, x):
    """ae.py:388-392; DiagonalGaussian (ae.py:342-346) has std = 0.00 => identity."""
    z = encoder(p, x)
    return decoder(p, z), z


# ------------------------------------------------------------------------------------------ VGG / LPIPS / D
def vgg_features(p, x, key):
    """utils.py:116-131 over torchvision's features[0:30]; `key(slice_no, idx)` -> state-dict prefix."""
    taps = []
    h = x
    for s, idxs in enumerate(VGG_IDX):
        if s > 0:
            h = F.max_pool2d(h, 2, 2)
        for idx in idxs:
            h = F.relu(R.conv2d(h, p[key(s + 1, idx) + ".weight"], p[key(s + 1, idx) + ".bias"], padding=1))
        taps.append(h)
    return taps


def lpips_forward(p, inp, tgt, masks=None):
    """utils.py:39-57 -> [B,1,1,1]."""
    shift, scale = p["scaling_layer.shift"].reshape(-1), p["scaling_layer.scale"].reshape(-1)
    key = lambda s, i: f"net.slice{s}.{i}"
    f0 = vgg_features(p, R.scaling_layer(inp, shift, scale), key)
    f1 = vgg_features(p, R.scaling_layer(tgt, shift, scale), key)
    val = 0
    for k in range(5):
        w = p.get(f"lin{k}.model.1.weight", p.get(f"lin{k}.model.0.weight"))
        val = val + R.lpips_tap(f0[k], f1[k], w.reshape(-1), None if masks is None else masks[k])
    return val


def disc_forward(p, x):
    """utils.py:187-203 -> [B, (H/16)*(W/16)]."""
    shift, scale = p["scaling_layer.shift"].reshape(-1), p["scaling_layer.scale"].reshape(-1)
    feats = vgg_features(p, R.scaling_layer(x, shift, scale), lambda s, i: f"slice{s}.0.{i}")
    strides = ((4, 4), (4, 2), (2, 2), (2, None), (1, None))
    out = 0
    for k, f in enumerate(feats):
        pre = f"binary_classifier{k + 1}."
        s1, s2 = strides[k]
        h = R.conv2d(f, p[pre + "0.weight"], p[pre + "0.bias"], stride=s1)
        if s2 is not None:
            h = R.conv2d(F.relu(h), p[pre + "2.weight"], p[pre + "2.bias"], stride=s2)
        out = out + h.flatten(1)
    return out


# ------------------------------------------------------------------------------------------ step
def cosine_with_warmup(step, warmup, total):
    """transformers.get_cosine_schedule_with_warmup lambda (vae_trainer.py:486-490)."""
    if step < warmup:
        return float(step) / float(max(1, warmup))
    progress = float(step - warmup) / float(max(1, total - warmup))
    return max(0.0, 0.5 * (1.0 + math.cos(math.pi * 0.5 * 2.0 * progress)))


# what the reference's CUDA path computes in (vae_trainer.py:18-19,453,538,623-624; utils.py:70-71) — for train_step_ref(arith=...)
REFERENCE_GPU_ARITH = {"encoder": "tf32", "decoder": "bf16", "lpips": "tf32", "disc": "tf32"}

VQ_KEY = "quantizer.embedding.weight"   # optional entry of RefState.vae: the config-5 codebook [K, D]


class RefState:
    """Parameters + AdamW moments of the restated trainer (plain tensors)."""

    def __init__(self, vae_p, lpips_p, disc_p=None, dtype=torch.float32):
        """dtype=torch.float64 runs the SAME restated step in double precision: the yardstick for how far two correct fp32
        evaluations of a GAN trajectory may drift apart (AdamW's sign-like first updates amplify round-off; tests/test_model.py)."""
        fl = lambda v: v.clone().to(dtype)                                                       # noqa: E731
        self.vae = {k: fl(v).requires_grad_() for k, v in vae_p.items()}
        self.lpips = {k: (fl(v) if v.dtype.is_floating_point else v.clone()) for k, v in lpips_p.items()}
        self.disc = None if disc_p is None else {k: (fl(v).requires_grad_() if v.dtype.is_floating_point and "scaling_layer" not in k
                                                     else (fl(v) if v.dtype.is_floating_point else v.clone())) for k, v in disc_p.items()}
        self.m_g = {k: torch.zeros_like(v) for k, v in self.vae.items()}
        self.v_g = {k: torch.zeros_like(v) for k, v in self.vae.items()}
        if self.disc is not None:
            self.m_d = {k: torch.zeros_like(v) for k, v in self.disc.items() if v.requires_grad}
            self.v_d = {k: torch.zeros_like(v) for k, v in self.disc.items() if v.requires_grad}
        self.step = 0
        self.lecam_real = 0.0
        self.lecam_fake = 0.0


def train_step_ref(st: RefState, x, *, do_ganloss=False, disc_type="hinge", use_lecam=False, learning_rate_vae=1e-5,
                   learning_rate_disc=2e-4, vae_ch=64, max_steps=1000, warmup_steps=200, lpips_masks=None,
                   rng=None, enc_size=None, flip_invariance=False, crop_invariance=False,
                   augment_before_perceptual_loss=False, decoder_also_perform_hr=False, downscale_factor=16,
                   do_clamp=False, clamp_th=8.0, vq_beta=0.25, arith=None):
    """vae_trainer.py:525-708 at world_size 1, LPIPS deterministic (masks given or eval mode).  Returns the
    logged scalars; mutates `st` like optimizer_G/D.step().  `rng` (random.Random-like) drives the augmentations
    in the reference's exact draw order (:534,:567,:572,:577-583,:665,:668); rng=None draws nothing and flips
    nothing (not even the unconditional 50 % flip of :534) for fixed-input parity tests.  `enc_size` is the
    encoder input size of the area resize (:531-533; the reference hard-codes 256).
    `arith` (error budgets only, see ops_ref.arith): {"encoder" | "decoder" | "lpips" | "disc": "fp32" | "tf32" | "bf16"} —
    REFERENCE_GPU_ARITH emulates what the reference's CUDA path computes in; None = the fp32 CPU path."""
    ar = lambda k: R.arith((arith or {}).get(k, "fp32"))          # noqa: E731
    out = {}
    x_hr = x
    x_enc = R.area_resize(x_hr, enc_size) if enc_size is not None and tuple(enc_size) != tuple(x.shape[-2:]) else x_hr
    if rng is not None and rng.random() < 0.5:                          # :534-536
        x_enc, x_hr = torch.flip(x_enc, [-1]), torch.flip(x_hr, [-1])
    with ar("encoder"):
        z = encoder(st.vae, x_enc)                                     # :538
    z_s = z.clamp(-clamp_th, clamp_th) if do_clamp else z              # :561-563 (reg = identity)
    if do_clamp:
        z = z_s
    vq_loss = None
    if VQ_KEY in st.vae:   # config 5 (not in the reference): the quantizer takes the place of `reg`
        from . import vq_oracle
        z_s, vq_loss, idx = vq_oracle.quantize(z_s, st.vae[VQ_KEY], vq_beta)
        out["indices"] = idx
    if rng is not None:
        nz = z_s.shape[1]
        if rng.random() < 0.5 and flip_invariance:                     # :567-570
            sign = torch.ones(nz); sign[nz - 4:nz - 2] = -1
            z_s = torch.flip(z_s, [-1]) * sign.view(1, -1, 1, 1)
            x_hr = torch.flip(x_hr, [-1])
        if rng.random() < 0.5 and flip_invariance:                     # :572-575
            sign = torch.ones(nz); sign[nz - 2:] = -1
            z_s = torch.flip(z_s, [-2]) * sign.view(1, -1, 1, 1)
            x_hr = torch.flip(x_hr, [-2])
        if rng.random() < 0.5 and crop_invariance:                     # :577-621
            z_h, z_w = z.shape[-2:]
            new_z_h, new_z_w = rng.randint(12, z_h - 1), rng.randint(12, z_w - 1)
            off_z_h, off_z_w = rng.randint(0, z_h - new_z_h - 1), rng.randint(0, z_w - new_z_w - 1)
            f = downscale_factor * (2 if decoder_also_perform_hr else 1)
            x_hr = x_hr[:, :, off_z_h * f:(off_z_h + new_z_h) * f, off_z_w * f:(off_z_w + new_z_w) * f]
            z_s = z_s[:, :, off_z_h:off_z_h + new_z_h, off_z_w:off_z_w + new_z_w]
    with ar("decoder"):
        recon = decoder(st.vae, z_s)                                   # :623-624
    x = x_hr
    out["target"] = x_hr
    if do_ganloss:                                                     # :629-659
        with ar("disc"):
            real_preds = disc_forward(st.disc, x)
            fake_preds = disc_forward(st.disc, recon.detach())
        d_loss, avg_r, avg_f, acc = R.gan_disc_loss(real_preds, fake_preds, disc_type)
        st.lecam_real = 0.9 * st.lecam_real + 0.1 * avg_r.item()
        st.lecam_fake = 0.9 * st.lecam_fake + 0.1 * avg_f.item()
        total = d_loss
        if use_lecam:
            total = total + 0.1 * ((real_preds - st.lecam_fake).pow(2).mean() + (fake_preds - st.lecam_real).pow(2).mean())
        names = [k for k, v in st.disc.items() if v.requires_grad]
        grads = torch.autograd.grad(total, [st.disc[k] for k in names], allow_unused=True)
        out["d_grads"] = {k: (g if g is not None else torch.zeros_like(st.disc[k])) for k, g in zip(names, grads)}
        st.step_d = getattr(st, "step_d", 0) + 1
        with torch.no_grad():
            for k in names:
                pn, mn, vn = R.adamw_step(st.disc[k], out["d_grads"][k], st.m_d[k], st.v_d[k], st.step_d, learning_rate_disc, 1e-3)
                st.disc[k].copy_(pn); st.m_d[k] = mn; st.v_d[k] = vn
        out.update(d_loss=d_loss.detach(), avg_real=avg_r.detach(), avg_fake=avg_f.detach(), disc_acc=acc)
    # GradNorm (vae_trainer.py:27-53): identity forward, normalised backward — restated via a hook
    rp = recon.clone()
    rp.register_hook(lambda g: R.gradnorm_backward(g, 1.0))
    x_aug = x
    if rng is not None and augment_before_perceptual_loss:             # :663-671
        if rng.random() < 0.5:
            rp, x_aug = torch.flip(rp, [-1]), torch.flip(x_aug, [-1])
        if rng.random() < 0.5:
            rp, x_aug = torch.flip(rp, [-2]), torch.flip(x_aug, [-2])
    with ar("lpips"):
        percep = lpips_forward(st.lpips, rp, x_aug, lpips_masks).mean()  # :676
    vae_loss = 0.1 * z.pow(2).mean()                                   # :202-209 (recon term x0.0)
    overall = percep + vae_loss
    if vq_loss is not None:
        overall = overall + vq_loss
        out["vq_loss"] = vq_loss.detach()
    if do_ganloss:                                                     # :682-696
        rg = recon.clone()
        rg.register_hook(lambda g: R.gradnorm_backward(g, 1.0))
        with ar("disc"):
            fake2 = disc_forward({k: v.detach() for k, v in st.disc.items()}, rg)
        g_gan = -fake2.mean() if disc_type == "hinge" else F.binary_cross_entropy_with_logits(fake2, torch.ones_like(fake2))
        overall = overall + g_gan
        out["g_gan_loss"] = g_gan.detach()
    names = list(st.vae.keys())
    grads = torch.autograd.grad(overall, [st.vae[k] for k in names])   # :701
    out["grads"] = dict(zip(names, grads))
    mult = cosine_with_warmup(st.step, warmup_steps, max_steps)
    st.step += 1
    with torch.no_grad():                                              # :455-468,:702
        for k, g in zip(names, grads):
            lr = (1e-4 if "conv_in" in k else learning_rate_vae / vae_ch) * mult
            pn, mn, vn = R.adamw_step(st.vae[k], g, st.m_g[k], st.v_g[k], st.step, lr, 1e-3)
            st.vae[k].copy_(pn); st.m_g[k] = mn; st.v_g[k] = vn
    out.update(overall_vae_loss=overall.detach(), perceptual_loss=percep.detach(), vae_loss=vae_loss.detach(),
               reconstructed=recon.detach(), z=z.detach())
    return out
