"""Generates tests/golden/*.npz by running the REAL reference (/root/reference) on CPU fp32.

Run in the build container only:   python tests/golden/make_golden.py
Inputs and weights are not stored: they are regenerated from oracle/weights.py (hash-based, machine
independent); the fixtures hold only what the reference's own modules computed from them.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import reference_import as RI  # noqa: E402
from oracle import weights as W            # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))

VAE_CFGS = {
    # name: (resolution, ch, ch_mult, num_res_blocks, z_channels, batch[, decoder_also_perform_hr, use_wavelet])
    "vae_ch32_m12_r16": (16, 32, [1, 2], 1, 4, 2),
    "vae_ch32_m124_r32": (32, 32, [1, 2, 4], 2, 8, 1),
    "vae_wavelet_hr_ch32_m12_r32": (32, 32, [1, 2], 1, 4, 1, True, True),   # launch_hdr.sh-style front/back ends
}


def cfg_fields(name):
    c = VAE_CFGS[name]
    return c[:6] + (c[6:] if len(c) > 6 else (False, False))


def vae_fixture(name):
    ae, utils, vt = RI.load()
    res, ch, mult, nrb, zc, b, hr, wav = cfg_fields(name)
    torch.manual_seed(0)
    vae = ae.VAE(resolution=res, in_channels=3, ch=ch, out_ch=3, ch_mult=list(mult), num_res_blocks=nrb, z_channels=zc,
                 use_attn=False, decoder_also_perform_hr=hr, use_wavelet=wav)
    vae.load_state_dict(W.randomize_state_dict(vae.state_dict(), seed=1), strict=True)
    x = W.image_batch(b, res, seed=3)
    recon, z = vae(x)
    gy = W.uniform_tensor(tuple(recon.shape), 99)
    (recon * gy).sum().backward()
    sd = dict(vae.named_parameters())
    pick = ["encoder.conv_in.weight", "encoder.down.0.block.0.norm1.weight", "encoder.down.0.block.0.conv2.weight",
            "encoder.mid.block_1.conv1.bias", "decoder.conv_out.weight", "decoder.up.0.block.0.norm2.bias",
            "decoder.up.1.upsample.conv.weight", "encoder.down.0.downsample.conv.weight"]
    out = {"recon": recon.detach().numpy(), "z": z.detach().numpy()}
    for k in pick:
        if k in sd:
            out["grad:" + k] = sd[k].grad.numpy()
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(name, "recon", tuple(recon.shape), "z", tuple(z.shape))


def loss_fixture():
    ae, utils, vt = RI.load()
    torch.manual_seed(0)
    lp = RI.in_ref_cwd(lambda: utils.LPIPS().eval())
    sd = W.randomize_state_dict(lp.state_dict(), seed=2, relu_net=True)
    lp.load_state_dict(sd, strict=True)
    a = W.image_batch(2, 32, seed=5).requires_grad_()
    b = W.image_batch(2, 32, seed=6)
    val = lp(a, b)
    val.sum().backward()
    disc = utils.PatchDiscriminator()
    disc.load_state_dict(W.randomize_state_dict(disc.state_dict(), seed=4, relu_net=True), strict=True)
    c = W.image_batch(2, 32, seed=7).requires_grad_()
    logits = disc(c)
    (logits * W.uniform_tensor(tuple(logits.shape), 11)).sum().backward()
    dparams = dict(disc.named_parameters())
    real, fake = W.uniform_tensor((3, 8), 21, -2, 2), W.uniform_tensor((3, 8), 22, -2, 2)
    hl = vt.gan_disc_loss(real, fake, "hinge")
    bl = vt.gan_disc_loss(real, fake, "bce")
    zz = W.uniform_tensor((2, 4, 8, 8), 23, -3, 3)
    vl, vd = vt.vae_loss_function(None, None, zz)
    g = W.uniform_tensor((2, 3, 8, 8), 24)
    gx = torch.zeros(2, 3, 8, 8, requires_grad=True)
    vt.gradnorm(gx, 0.5).backward(g)
    np.savez_compressed(
        os.path.join(OUT, "losses.npz"), lpips_val=val.detach().numpy(), lpips_grad=a.grad.numpy(),
        disc_logits=logits.detach().numpy(), disc_grad_x=c.grad.numpy(),
        disc_grad_w=dparams["slice2.0.7.weight"].grad.numpy(), disc_grad_head=dparams["binary_classifier2.0.weight"].grad.numpy(),
        hinge=np.array([hl[0].item(), hl[1], hl[2], hl[3]]), bce=np.array([bl[0].item(), bl[1], bl[2], bl[3]]),
        vae_loss=np.array([vl.item(), vd["kl_loss"], vd["average_of_abs_z"], vd["std_of_abs_z"]]),
        gradnorm=gx.grad.numpy())
    print("losses: lpips", val.flatten().tolist())


def frontend_fixture():
    """utils.wavelet_transform_multi_channel and the trainer's area resize (vae_trainer.py:531-533) on seeded images."""
    ae, utils, vt = RI.load()
    utils.prepare_filter("cpu")
    x = W.image_batch(2, 16, seed=12)
    wav = utils.wavelet_transform_multi_channel(x)
    area = torch.nn.functional.interpolate(W.image_batch(2, 32, seed=13), size=(16, 16), mode="area")
    np.savez_compressed(os.path.join(OUT, "frontend.npz"), wavelet=wav.numpy(), area=area.numpy())
    print("frontend: wavelet", tuple(wav.shape), "area", tuple(area.shape))


def attn_fixture():
    """ae.AttnBlock (ae.py:56-93) stand-alone: the reference cannot build an Encoder with use_attn=True,
    but the block itself runs."""
    ae, utils, vt = RI.load()
    torch.manual_seed(0)
    blk = ae.AttnBlock(128)
    blk.load_state_dict(W.randomize_state_dict(blk.state_dict(), seed=9), strict=True)
    x = W.uniform_tensor((2, 128, 6, 5), 61, -1.5, 1.5).requires_grad_()
    y = blk(x)
    (y * W.uniform_tensor(tuple(y.shape), 62)).sum().backward()
    out = {"y": y.detach().numpy(), "grad:x": x.grad.numpy()}
    for k, v in blk.named_parameters():
        out["grad:" + k] = v.grad.numpy()
    np.savez_compressed(os.path.join(OUT, "attn_block.npz"), **out)
    print("attn_block: y", tuple(y.shape))


TVAE_CFGS = {
    # name: (ch, ch_mult, num_res_blocks, z_channels, (B, T, H, W))
    "tvae_ch32_m12_t4": (32, [1, 2], 1, 4, (1, 4, 8, 8)),          # one Downsample / Upsample; attention with 8-channel heads
    "tvae_ch32_m124_t5": (32, [1, 2, 4], 1, 2, (2, 5, 8, 12)),      # odd frame count, non-square frames, 16-channel heads
}


def tvae_inputs(name):
    """-> (x, noise shape): seeded video batch in [-1, 1] and the latent shape the DiagonalGaussian draws for."""
    ch, mult, nrb, zc, (b, t, h, w) = TVAE_CFGS[name]
    x = W.uniform_tensor((b, 3, t, h, w), 31)
    for _ in range(len(mult) - 1):
        t, h, w = (t - 2) // 2 + 1, (h - 2) // 2 + 1, (w - 2) // 2 + 1
    return x, (b, zc, t, h, w)


def tvae_fixture(name):
    """tae.TVAE forward + backward on CPU fp32.  DiagonalGaussian draws randn_like(mean) from the global generator
    (tae.py:254): the fixture replaces that draw by the seeded hash noise the tests regenerate, through a patched
    torch.randn_like for the duration of the forward."""
    tae = RI.load_tae()
    ch, mult, nrb, zc, _ = TVAE_CFGS[name]
    torch.manual_seed(0)
    vae = tae.TVAE(resolution=8, in_channels=3, ch=ch, out_ch=3, ch_mult=list(mult), num_res_blocks=nrb, z_channels=zc)
    vae.load_state_dict(W.randomize_state_dict(vae.state_dict(), seed=5), strict=True)
    x, zshape = tvae_inputs(name)
    noise = W.uniform_tensor(zshape, 32, -1.5, 1.5)
    orig = torch.randn_like
    torch.randn_like = lambda t, *a, **k: noise.to(t.dtype)
    try:
        recon, z = vae(x)
    finally:
        torch.randn_like = orig
    assert tuple(z.shape) == (zshape[0], 2 * zc) + zshape[2:]
    gy = W.uniform_tensor(tuple(recon.shape), 33)
    (recon * gy).sum().backward()
    sd = dict(vae.named_parameters())
    # (biases that only feed one-channel GroupNorm groups — every 32-channel level — have an exactly zero gradient: not picked)
    pick = ["encoder.conv_in.weight", "encoder.down.1.block.0.conv1.bias", "encoder.down.0.block.0.conv2.weight", "encoder.down.0.downsample.conv.weight",
            "encoder.down.0.downsample.conv.bias", "encoder.down.1.block.0.nin_shortcut.weight", "encoder.mid.attn_1.qkv.weight",
            "encoder.mid.attn_1.norm.weight", "encoder.conv_out.weight", "decoder.conv_in.weight", "decoder.mid.attn_1.proj_out.weight",
            "decoder.up.1.upsample.conv.weight", "decoder.up.0.block.0.norm2.bias", "decoder.up.0.block.0.nin_shortcut.weight",
            "decoder.conv_out.weight", "decoder.conv_out.bias"]
    out = {"recon": recon.detach().numpy(), "z": z.detach().numpy()}
    for k in pick:
        if k in sd:
            out["grad:" + k] = sd[k].grad.numpy()
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(name, "recon", tuple(recon.shape), "z", tuple(z.shape), "grads", sum(k.startswith("grad:") for k in out))


PHOTOS = ("origin.png", "lavender.jpg", "chinatown.jpg", "cosplayers.jpg")     # /root/reference/contents (README.md's samples)


def photo_fixture():
    """tests/golden/photos_256.npz: four of the reference's own sample photographs (contents/) as the trainer would see
    them — shorter side resized to 512, centre crop 512 (vae_trainer.py:96-103), area-resized to 256 (vae_trainer.py:531-533) — stored as
    uint8 [4, 3, 256, 256]; `oracle.weights.photo_batch` maps them to [-1, 1] like ToTensor + Normalize(0.5, 0.5) (vae_trainer.py:98-99).
    Data, not source: smooth images with large per-region DC, what `uniform[-1, 1]` noise never shows the kernels."""
    from PIL import Image
    out = []
    for name in PHOTOS:
        im = Image.open(os.path.join("/root/reference/contents", name)).convert("RGB")
        w, h = im.size
        s = 512 / min(w, h)
        if s != 1.0:
            im = im.resize((max(512, round(w * s)), max(512, round(h * s))), Image.BICUBIC)
        w, h = im.size
        l, t = (w - 512) // 2, (h - 512) // 2
        a = np.asarray(im.crop((l, t, l + 512, t + 512)), dtype=np.float64)                # [512, 512, 3]
        a = a.reshape(256, 2, 256, 2, 3).mean(axis=(1, 3))                                  # area resize by 2
        out.append(np.clip(np.rint(a), 0, 255).astype(np.uint8).transpose(2, 0, 1))
    np.savez_compressed(os.path.join(OUT, "photos_256.npz"), images=np.stack(out), names=np.array(PHOTOS))
    print("photos_256:", np.stack(out).shape)


PHOTO_VAE_CFGS = {
    # name: (resolution, ch, ch_mult, num_res_blocks, z_channels, photo indices) — 2 / 4 / 8 channels per GroupNorm group
    "photo_small": (16, 64, [1, 2], 1, 4, [0]),               # emulator-sized
    "photo_large": (64, 64, [1, 2, 4], 2, 8, [0, 2]),          # GPU only
}


def photo_model_fixture():
    """The REAL reference's modules on photographs (area-resized to the model's resolution) with trained-like, biased weights:
    `W.randomize_state_dict(..., bias_scale=...)` puts a DC of up to +-30 behind every conv of the VAE, so every GroupNorm sees
    |mean| >> std (round-5 verdict, item 1): VAE forward + backward, LPIPS and the PatchDiscriminator (64 x 64)."""
    ae, utils, vt = RI.load()
    out = {}
    for name, (res, ch, mult, nrb, zc, idx) in PHOTO_VAE_CFGS.items():
        torch.manual_seed(0)
        vae = ae.VAE(resolution=res, in_channels=3, ch=ch, out_ch=3, ch_mult=list(mult), num_res_blocks=nrb, z_channels=zc,
                     use_attn=False, decoder_also_perform_hr=False, use_wavelet=False)
        vae.load_state_dict(W.randomize_state_dict(vae.state_dict(), seed=1, bias_scale=W.PHOTO_BIAS_SCALE), strict=True)
        x = W.photo_batch(idx, res)
        recon, z = vae(x)
        gy = W.uniform_tensor(tuple(recon.shape), 99)
        (recon * gy).sum().backward()
        sd = dict(vae.named_parameters())
        pick = ["encoder.conv_in.weight", "encoder.down.0.block.0.norm1.weight", "encoder.down.0.block.0.conv2.weight",
                "encoder.mid.block_1.conv1.bias", "decoder.conv_out.weight", "decoder.up.0.block.0.norm2.bias",
                "decoder.up.1.upsample.conv.weight", "encoder.down.0.downsample.conv.weight", "decoder.mid.block_2.norm1.weight",
                "decoder.mid.block_2.conv1.weight"]
        out[name + ":recon"], out[name + ":z"] = recon.detach().numpy(), z.detach().numpy()
        for k in pick:                                   # (large conv weights: the first 4 output channels — fixtures stay small)
            gk = sd[k].grad
            out[name + ":grad:" + k] = (gk[:4] if gk.dim() == 4 and gk.numel() > 40000 else gk).numpy()
        print(name, "recon", tuple(recon.shape), "|recon|max", float(recon.detach().abs().max()))
    lp = RI.in_ref_cwd(lambda: utils.LPIPS().eval())
    lp.load_state_dict(W.randomize_state_dict(lp.state_dict(), seed=2, relu_net=True, bias_scale=W.PHOTO_VGG_BIAS_SCALE), strict=True)
    a = W.photo_batch([1, 3], 64).requires_grad_()
    b = W.photo_batch([0, 2], 64)
    val = lp(a, b)
    val.sum().backward()
    disc = utils.PatchDiscriminator()
    disc.load_state_dict(W.randomize_state_dict(disc.state_dict(), seed=4, relu_net=True, bias_scale=W.PHOTO_VGG_BIAS_SCALE), strict=True)
    c = W.photo_batch([3, 1], 64).requires_grad_()
    logits = disc(c)
    (logits * W.uniform_tensor(tuple(logits.shape), 11)).sum().backward()
    out.update(lpips_val=val.detach().numpy(), lpips_grad=a.grad.numpy(), disc_logits=logits.detach().numpy(), disc_grad_x=c.grad.numpy())
    np.savez_compressed(os.path.join(OUT, "photo_models.npz"), **out)
    print("photo_models: lpips", val.flatten().tolist(), "logits", float(logits.abs().max()))


if __name__ == "__main__":
    only = sys.argv[1:]
    if "photos" in only:            # (only on request: needs /root/reference/contents and PIL)
        photo_fixture()
    if not only or "photo_models" in only:
        photo_model_fixture()
    for n in TVAE_CFGS:
        if not only or n in only:
            tvae_fixture(n)
    for n in VAE_CFGS:
        if not only or n in only:
            vae_fixture(n)
    if not only or "losses" in only:
        loss_fixture()
    if not only or "frontend" in only:
        frontend_fixture()
    if not only or "attn" in only:
        attn_fixture()
