"""Pins the oracle (oracle/*.py restatement) — CPU only.

 1. against the committed golden fixtures, which were produced by the reference's own modules
    (tests/golden/make_golden.py) — runs everywhere;
 2. against the reference imported live from /root/reference — runs only where it is mounted
    (the build container), on additional configurations.
The reference ships no golden vectors or known-answer tests of its own.
"""
import os

import numpy as np
import pytest
import torch

from oracle import model_ref as M
from oracle import ops_ref as R
from oracle import reference_import as RI
from oracle import weights as W

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
from golden.make_golden import VAE_CFGS, cfg_fields  # noqa: E402


def _vae_state(res, ch, mult, nrb, zc, hr=False, wav=False):
    """Key names/shapes of the reference VAE state dict, built without the reference: via our module."""
    import vqgan_training_amd as vq
    vae = vq.ae.VAE(res, 3, ch, 3, list(mult), nrb, zc, False, hr, wav)
    return W.randomize_state_dict(vae.state_dict(), seed=1)


def close(a, b, tol):
    a = torch.as_tensor(a).double(); b = torch.as_tensor(b).double()
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item() < tol


@pytest.mark.parametrize("name", list(VAE_CFGS))
def test_vae_restatement_matches_golden(name):
    res, ch, mult, nrb, zc, b, hr, wav = cfg_fields(name)
    g = np.load(os.path.join(GOLD, name + ".npz"))
    p = {k: v.requires_grad_() for k, v in _vae_state(res, ch, mult, nrb, zc, hr, wav).items()}
    x = W.image_batch(b, res, seed=3)
    recon, z = M.vae_forward(p, x)
    assert close(recon, g["recon"], 1e-5) and close(z, g["z"], 1e-5)
    (recon * W.uniform_tensor(tuple(recon.shape), 99)).sum().backward()
    for k in g.files:
        if k.startswith("grad:"):
            assert close(p[k[5:]].grad, g[k], 1e-4), k


def test_frontend_restatements_match_golden():
    """Wavelet front-end (utils.py:206-247) and the area resize (vae_trainer.py:531-533)."""
    g = np.load(os.path.join(GOLD, "frontend.npz"))
    assert close(R.wavelet_transform(W.image_batch(2, 16, seed=12)), g["wavelet"], 1e-6)
    assert close(R.area_resize(W.image_batch(2, 32, seed=13), (16, 16)), g["area"], 1e-6)


def _attn_state():
    import vqgan_training_amd as vq
    return W.randomize_state_dict(vq.ae.AttnBlock(128).state_dict(), seed=9)


def test_attn_block_restatement_matches_golden():
    """ae.py:56-93 (AttnBlock alone: the reference cannot instantiate it inside Encoder/Decoder)."""
    g = np.load(os.path.join(GOLD, "attn_block.npz"))
    p = {"a." + k: v.requires_grad_() for k, v in _attn_state().items()}
    x = W.uniform_tensor((2, 128, 6, 5), 61, -1.5, 1.5).requires_grad_()
    y = R.attn_block(x, p, "a.")
    (y * W.uniform_tensor(tuple(y.shape), 62)).sum().backward()
    assert close(y, g["y"], 1e-5) and close(x.grad, g["grad:x"], 1e-4)
    for k in ("norm.weight", "norm.bias", "qkv.weight", "proj_out.weight"):
        assert close(p["a." + k].grad, g["grad:" + k], 1e-4), k


def test_loss_restatements_match_golden():
    import vqgan_training_amd as vq
    g = np.load(os.path.join(GOLD, "losses.npz"))
    lp = vq.utils.LPIPS(pretrained_path=None)
    p = W.randomize_state_dict(lp.state_dict(), seed=2, relu_net=True)
    a = W.image_batch(2, 32, seed=5).requires_grad_()
    val = M.lpips_forward(p, a, W.image_batch(2, 32, seed=6))
    val.sum().backward()
    assert close(val, g["lpips_val"], 1e-5) and close(a.grad, g["lpips_grad"], 1e-4)
    dp = {k: (v.requires_grad_() if v.dtype.is_floating_point else v) for k, v in
          W.randomize_state_dict(vq.utils.PatchDiscriminator().state_dict(), seed=4, relu_net=True).items()}
    c = W.image_batch(2, 32, seed=7).requires_grad_()
    logits = M.disc_forward(dp, c)
    (logits * W.uniform_tensor(tuple(logits.shape), 11)).sum().backward()
    assert close(logits, g["disc_logits"], 1e-5) and close(c.grad, g["disc_grad_x"], 1e-4)
    assert close(dp["slice2.0.7.weight"].grad, g["disc_grad_w"], 1e-4)
    assert close(dp["binary_classifier2.0.weight"].grad, g["disc_grad_head"], 1e-4)
    real, fake = W.uniform_tensor((3, 8), 21, -2, 2), W.uniform_tensor((3, 8), 22, -2, 2)
    for kind in ("hinge", "bce"):
        l, ar, af, acc = R.gan_disc_loss(real, fake, kind)
        assert close(torch.stack([l, ar, af, acc]), g[kind], 1e-6)
    zz = W.uniform_tensor((2, 4, 8, 8), 23, -3, 3)
    want = g["vae_loss"]
    assert close(0.1 * zz.pow(2).mean(), want[0], 1e-6) and close(zz.abs().std(), want[3], 1e-6)
    gg = W.uniform_tensor((2, 3, 8, 8), 24)
    assert close(R.gradnorm_backward(gg, 0.5), g["gradnorm"], 1e-6)


needs_ref = pytest.mark.skipif(not RI.available(), reason="/root/reference is only mounted in the build container")


@needs_ref
def test_state_dict_surface_matches_reference():
    """Same keys, shapes and (under the same seed) the same initial values as ae.VAE / utils.* ."""
    import vqgan_training_amd as vq
    ae, utils, vt = RI.load()
    for cfg in ((32, 32, [1, 2], 1, 4), (64, 64, [1, 2, 4, 4], 2, 16)):
        res, ch, mult, nrb, zc = cfg
        torch.manual_seed(42)
        ref = ae.VAE(res, 3, ch, 3, list(mult), nrb, zc, False, False, False).state_dict()
        torch.manual_seed(42)
        ours = vq.ae.VAE(res, 3, ch, 3, list(mult), nrb, zc, False, False, False).state_dict()
        assert list(ref) == list(ours)
        for k in ref:
            assert torch.equal(ref[k], ours[k]), k
    ref_l = RI.in_ref_cwd(lambda: utils.LPIPS()).state_dict()
    our_l = vq.utils.LPIPS(pretrained_path=None).state_dict()
    assert {k: tuple(v.shape) for k, v in ref_l.items()} == {k: tuple(v.shape) for k, v in our_l.items()}
    ref_d = utils.PatchDiscriminator().state_dict()
    our_d = vq.utils.PatchDiscriminator().state_dict()
    assert {k: tuple(v.shape) for k, v in ref_d.items()} == {k: tuple(v.shape) for k, v in our_d.items()}
    hr = ae.VAE(32, 3, 32, 3, [1, 2], 1, 4, False, True, False).state_dict()
    ours_hr = vq.ae.VAE(32, 3, 32, 3, [1, 2], 1, 4, False, True, False).state_dict()
    assert {k: tuple(v.shape) for k, v in hr.items()} == {k: tuple(v.shape) for k, v in ours_hr.items()}


@needs_ref
def test_train_step_restatement_matches_reference_modules():
    """One restated step (oracle) vs the same step assembled from the reference's own modules/functions
    (vae_trainer.py:525-708 with world_size 1, LPIPS eval, hinge GAN)."""
    ae, utils, vt = RI.load()
    res, ch, mult = 32, 32, [1, 2]
    torch.manual_seed(0)
    vae = ae.VAE(res, 3, ch, 3, list(mult), 1, 4, False, False, False)
    vae.load_state_dict(W.randomize_state_dict(vae.state_dict(), 1))
    lp = RI.in_ref_cwd(lambda: utils.LPIPS().eval())
    lp.load_state_dict(W.randomize_state_dict(lp.state_dict(), 2, relu_net=True))
    disc = utils.PatchDiscriminator()
    disc.load_state_dict(W.randomize_state_dict(disc.state_dict(), 4, relu_net=True))
    st = M.RefState(vae.state_dict(), lp.state_dict(), disc.state_dict())
    x = W.image_batch(2, res, seed=8)
    # --- reference modules, loop body restated literally
    opt_g = torch.optim.AdamW([{"params": [p for n, p in vae.named_parameters() if "conv_in" not in n], "lr": 1e-2 / ch},
                               {"params": [p for n, p in vae.named_parameters() if "conv_in" in n], "lr": 1e-4}],
                              weight_decay=1e-3, betas=(0.9, 0.95))
    opt_d = torch.optim.AdamW(disc.parameters(), lr=2e-4, weight_decay=1e-3, betas=(0.9, 0.95))
    from transformers import get_cosine_schedule_with_warmup
    sched = get_cosine_schedule_with_warmup(opt_g, 2, 10)
    outs_ref, outs = [], []
    for it in range(3):
        z = vae.encoder(x); zs = vae.reg(z); rec = vae.decoder(zs)
        rp, fp = disc(x), disc(rec.detach())
        d_loss, ar, af, acc = vt.gan_disc_loss(rp, fp, "hinge")
        opt_d.zero_grad(); d_loss.mean().backward(retain_graph=True); opt_d.step()
        percep = lp(vt.gradnorm(rec), x).mean()
        vloss, _ = vt.vae_loss_function(x, vt.gradnorm(rec, weight=0.001), z)
        g_gan = -disc(vt.gradnorm(rec, weight=1.0)).mean()
        overall = percep + g_gan + vloss
        overall.backward()
        ref_grads = {n: p.grad.clone() for n, p in vae.named_parameters()}
        opt_g.step(); opt_g.zero_grad(); sched.step(); opt_d.zero_grad()
        outs_ref.append((overall.item(), percep.item(), d_loss.item()))
        o = M.train_step_ref(st, x, do_ganloss=True, disc_type="hinge", learning_rate_vae=1e-2, vae_ch=ch,
                             max_steps=10, warmup_steps=2)
        outs.append((o["overall_vae_loss"].item(), o["perceptual_loss"].item(), o["d_loss"].item()))
        if it == 0:
            # gradients before Adam (Adam turns round-off on exactly-zero gradients, e.g. a conv bias in
            # front of a GroupNorm with one channel per group, into +-lr steps — compare those loosely)
            gmax = max(g.abs().max().item() for g in ref_grads.values())
            for k, g in ref_grads.items():
                assert (o["grads"][k] - g).abs().max().item() < 1e-4 * gmax, k
    # steps 2 and 3 run on the updated parameters: the losses pin optimizer + schedule + D update
    assert close(torch.tensor(outs), torch.tensor(outs_ref), 2e-5)
    for k, v in vae.state_dict().items():
        assert (st.vae[k] - v).abs().max().item() < 2.5e-3, k      # bounded by 3 Adam steps of <= lr


@needs_ref
def test_checkpoints_interchange_with_reference(tmp_path):
    """N4 wire formats both ways: our bf16 safetensors export loads strict into the reference VAE().bfloat16()
    (README.hf.md:38-40), and a reference-side `module.`-prefixed torch checkpoint (vae_trainer.py:903-907) loads into ours."""
    import vqgan_training_amd as vq
    from safetensors.torch import load_file
    ae, utils, vt = RI.load()
    args = (32, 3, 32, 3, [1, 2], 1, 4, False, False, False)
    ours = vq.ae.VAE(*args[:4], list(args[4]), *args[5:])
    ours.load_state_dict(W.randomize_state_dict(ours.state_dict(), 1))
    path = str(tmp_path / "x_bf16.pt")
    vq.vae_trainer.export_bf16_safetensors(ours, path)
    ref = ae.VAE(*args[:4], list(args[4]), *args[5:]).bfloat16()
    ref.load_state_dict(load_file(path))                                   # strict
    for k, v in ref.state_dict().items():
        assert torch.equal(v, ours.state_dict()[k].to(torch.bfloat16)), k
    ref32 = ae.VAE(*args[:4], list(args[4]), *args[5:])
    ref32.load_state_dict(W.randomize_state_dict(ref32.state_dict(), 7))
    p2 = str(tmp_path / "ref.pt")
    torch.save({"module." + k: v for k, v in ref32.state_dict().items()}, p2)
    vq.vae_trainer.load_checkpoint(ours, p2)
    for k, v in ours.state_dict().items():
        assert torch.equal(v, ref32.state_dict()[k]), k
