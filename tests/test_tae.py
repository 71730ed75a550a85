"""The 3-D TVAE (reference tae.py) on the HIP path.

  * oracle (oracle/tae_ref.py) vs the golden fixtures the reference's own tae.TVAE produced (tests/golden/tvae_*.npz) and,
    where /root/reference is mounted, vs the reference module live — including the state-dict surface;
  * the HIP modules (vqgan-training_amd/tae.py) vs the same fixtures: forward and gradients in the fp32-class parity
    mode, forward in bf16 throughput mode;
  * ops.conv3d alone vs F.conv3d for the three layer kinds, with batch > 1 and odd frame counts (per-sample temporal
    taps, the dropped zero frame of Downsample, the frame copies of Upsample).
"""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import vqgan_training_amd as vq
from vqgan_training_amd import ops
from oracle import reference_import as RI
from oracle import tae_ref as T
from oracle import weights as W
from golden.make_golden import TVAE_CFGS, tvae_inputs

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def rel(a, b):
    a = torch.as_tensor(a).detach().double().cpu(); b = torch.as_tensor(b).detach().double().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()


def _make(name):
    ch, mult, nrb, zc, _ = TVAE_CFGS[name]
    vae = vq.tae.TVAE(resolution=8, in_channels=3, ch=ch, out_ch=3, ch_mult=list(mult), num_res_blocks=nrb, z_channels=zc)
    vae.load_state_dict(W.randomize_state_dict(vae.state_dict(), seed=5), strict=True)
    return vae


@pytest.mark.parametrize("name", list(TVAE_CFGS))
def test_oracle_matches_reference_golden(name):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    p = {k: v.clone().requires_grad_() for k, v in _make(name).state_dict().items()}
    x, zshape = tvae_inputs(name)
    recon, z = T.tvae_forward(p, x, W.uniform_tensor(zshape, 32, -1.5, 1.5))
    assert rel(recon, g["recon"]) < 1e-5 and rel(z, g["z"]) < 1e-5
    (recon * W.uniform_tensor(tuple(recon.shape), 33)).sum().backward()
    for k in g.files:
        if k.startswith("grad:"):
            assert rel(p[k[5:]].grad, g[k]) < 1e-4, k


@pytest.mark.skipif(not RI.available(), reason="/root/reference is only mounted in the build container")
def test_state_dict_surface_and_init_match_reference():
    """Same keys, shapes and — under the same seed — the same initial values as tae.TVAE (parameter creation order and
    initialisers, tae.py:13-23,60-81,123-170,189-224), and strict loading both ways."""
    tae = RI.load_tae()
    kw = dict(resolution=16, in_channels=3, ch=32, out_ch=3, ch_mult=[1, 2, 2], num_res_blocks=2, z_channels=4)
    torch.manual_seed(7)
    ref = tae.TVAE(**kw)
    torch.manual_seed(7)
    ours = vq.tae.TVAE(**kw)
    rs, os_ = ref.state_dict(), ours.state_dict()
    assert list(rs) == list(os_)
    for k in rs:
        assert rs[k].shape == os_[k].shape and torch.equal(rs[k], os_[k]), k
    ref.load_state_dict(os_, strict=True)
    ours.load_state_dict(rs, strict=True)
    assert ours.decoder.z_shape == ref.decoder.z_shape and ours.encoder.mid.attn_1.head_dim == ref.encoder.mid.attn_1.head_dim


@pytest.mark.parametrize("name", list(TVAE_CFGS))
def test_tvae_matches_reference_golden(backend, name):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    ops.set_default_precision("fp32x3")
    vae = _make(name).to(backend.device).set_precision("fp32x3")
    x, zshape = tvae_inputs(name)
    recon, z = vae(x.to(backend.device), noise=W.uniform_tensor(zshape, 32, -1.5, 1.5).to(backend.device))
    assert rel(recon, g["recon"]) < 2e-4 and rel(z, g["z"]) < 2e-4
    (recon * W.uniform_tensor(tuple(recon.shape), 33).to(backend.device)).sum().backward()
    params = dict(vae.named_parameters())
    for k in g.files:
        if k.startswith("grad:"):
            assert rel(params[k[5:]].grad, g[k]) < 5e-4, k


def test_tvae_bf16_mode_close_to_reference(backend):
    """Throughput mode: bf16 storage / MFMA operands, three roundings of the running sum per 3x3x3 conv."""
    name = "tvae_ch32_m12_t4"
    g = np.load(os.path.join(GOLD, name + ".npz"))
    ops.set_default_precision("bf16")
    vae = _make(name).to(backend.device).set_precision("bf16")
    x, zshape = tvae_inputs(name)
    recon, z = vae(x.to(backend.device), noise=W.uniform_tensor(zshape, 32, -1.5, 1.5).to(backend.device))
    assert rel(recon, g["recon"]) < 4e-2 and rel(z, g["z"]) < 4e-2
    recon.float().square().mean().backward()
    assert all(torch.isfinite(p.grad).all() for p in vae.parameters())


def test_tvae_samples_when_no_noise_is_given(backend):
    """DiagonalGaussian draws its own noise like tae.py:254: two forwards differ, the latent statistics do not."""
    name = "tvae_ch32_m12_t4"
    ops.set_default_precision("fp32x3")
    vae = _make(name).to(backend.device).set_precision("fp32x3")
    x, _ = tvae_inputs(name)
    with torch.no_grad():
        (r1, z1), (r2, z2) = vae(x.to(backend.device)), vae(x.to(backend.device))
    assert torch.equal(z1, z2) and not torch.equal(r1, r2)
    vae.reg.sample = False
    with torch.no_grad():
        r3, _ = vae(x.to(backend.device))
        r4, _ = vae(x.to(backend.device))
    assert torch.equal(r3, r4)


@pytest.mark.parametrize("mode,n,t,h,w,cin,cout", [
    ("same", 2, 3, 6, 5, 16, 24), ("same", 1, 1, 4, 4, 8, 8), ("same", 1, 4, 8, 8, 3, 16), ("same", 2, 2, 6, 6, 16, 3),
    ("down", 2, 4, 6, 6, 16, 16), ("down", 2, 5, 7, 6, 16, 16), ("down", 1, 2, 4, 4, 8, 8),
    ("up", 2, 2, 3, 4, 16, 16), ("up", 1, 3, 4, 4, 8, 8)])
def test_conv3d_kinds_match_torch(backend, mode, n, t, h, w, cin, cout):
    """ops.conv3d (temporal taps over frame runs, in-place accumulation, per-tap weight gradients) vs F.conv3d."""
    ops.set_default_precision("fp32x3")
    x = W.uniform_tensor((n, cin, t, h, w), 41)
    wt = W.uniform_tensor((cout, cin, 3, 3, 3), 42, -0.2, 0.2)
    b = W.uniform_tensor((cout,), 43, -0.5, 0.5)
    xr, wr, br = x.clone().requires_grad_(), wt.clone().requires_grad_(), b.clone().requires_grad_()
    if mode == "same":
        ref = F.conv3d(xr, wr, br, padding=1)
    elif mode == "down":
        ref = T.downsample(xr, wr, br)
    else:
        ref = T.upsample(xr, wr, br)
    res = W.uniform_tensor(tuple(ref.shape), 44) if mode == "same" else None
    g = W.uniform_tensor(tuple(ref.shape), 45)
    ((ref + res if res is not None else ref) * g).sum().backward()
    dev = backend.device
    xd, wd, bd = x.to(dev).requires_grad_(), wt.to(dev).requires_grad_(), b.to(dev).requires_grad_()
    cl = lambda u: vq.tae._to_video_cl(u, "fp32x3")                                   # noqa: E731
    y = ops.conv3d(cl(xd), wd, bd, residual=cl(res.to(dev)) if res is not None else None, mode=mode)
    y = vq.tae._from_video_cl(y, cout)
    assert tuple(y.shape) == tuple(ref.shape)
    assert rel(y, ref + res if res is not None else ref) < 2e-5
    (y * g.to(dev)).sum().backward()
    assert rel(xd.grad, xr.grad) < 2e-5 and rel(wd.grad, wr.grad) < 2e-5 and rel(bd.grad, br.grad) < 2e-5


def test_conv3d_sees_weight_updates(backend):
    """The tap-major weight copies and their packed operands follow in-place parameter updates (optimizer steps)."""
    ops.set_default_precision("fp32x3")
    dev = backend.device
    x = vq.tae._to_video_cl(W.uniform_tensor((1, 8, 2, 4, 4), 51).to(dev), "fp32x3")
    conv = vq.tae.Conv3d(8, 8, 3, 1, 1).to(dev)
    with torch.no_grad():
        y1 = conv(x).clone()
        conv.weight.mul_(2.0)
        conv.bias.zero_()
        y2 = conv(x)
        ref = F.conv3d(vq.tae._from_video_cl(x, 8), conv.weight, conv.bias, padding=1)
    assert rel(vq.tae._from_video_cl(y2, 8), ref) < 2e-5 and not torch.equal(y1, y2)


def test_tvae_training_steps_with_fused_adamw_match_oracle(backend):
    """Two optimizer steps on the flat-buffer AdamW (gradient sinks for the 1x1x1 / GroupNorm parameters, autograd
    accumulation for the tap-assembled 3x3x3 gradients, tap copies + packed operands refreshed after each update) vs
    torch.optim.AdamW on the oracle: the loss trajectory must agree."""
    name = "tvae_ch32_m12_t4"
    ops.set_default_precision("fp32x3")
    dev = backend.device
    vae = _make(name).to(dev).set_precision("fp32x3")
    p = {k: v.clone().requires_grad_() for k, v in _make(name).state_dict().items()}
    x, zshape = tvae_inputs(name)
    noise = W.uniform_tensor(zshape, 32, -1.5, 1.5)
    hp = dict(lr=2e-3, betas=(0.9, 0.95), eps=1e-8, weight_decay=1e-3)
    opt = vq.optim.FusedAdamW(vae.parameters(), **hp)
    ref_opt = torch.optim.AdamW(list(p.values()), **hp)
    got, want = [], []
    for _ in range(3):
        opt.zero_grad()
        recon, z = vae(x.to(dev), noise=noise.to(dev))
        loss = (recon - x.to(dev)).square().mean() + 1e-3 * z.square().mean()
        loss.backward()
        opt.step()
        got.append(loss.item())
        ref_opt.zero_grad()
        r, zz = T.tvae_forward(p, x, noise)
        ref_loss = (r - x).square().mean() + 1e-3 * zz.square().mean()
        ref_loss.backward()
        ref_opt.step()
        want.append(ref_loss.item())
    assert want[2] < want[0]                       # the steps do something
    for a, b in zip(got, want):
        assert abs(a - b) <= 2e-3 * abs(b), (got, want)


@pytest.mark.parametrize("mode,n,t,h,w,cin,cout", [("same", 2, 3, 8, 8, 64, 64), ("down", 1, 4, 8, 8, 64, 64), ("up", 1, 2, 4, 4, 64, 128)])
def test_conv3d_bf16_on_the_lds_dma_kernels(backend, mode, n, t, h, w, cin, cout):
    """bf16 storage with 64-channel multiples: the LDS-DMA implicit-GEMM kernels (in-place accumulation of the temporal taps
    through the LDS-transposed epilogue) and the bf16 weight-gradient kernels, vs F.conv3d on the bf16-rounded operands."""
    ops.set_default_precision("bf16")
    bf = lambda u: u.to(torch.bfloat16).float()                                      # noqa: E731
    x = bf(W.uniform_tensor((n, cin, t, h, w), 46))
    wt = W.uniform_tensor((cout, cin, 3, 3, 3), 47, -0.05, 0.05)
    b = W.uniform_tensor((cout,), 48, -0.5, 0.5)
    xr, wr, br = x.clone().requires_grad_(), bf(wt).requires_grad_(), b.clone().requires_grad_()
    ref = {"same": lambda: F.conv3d(xr, wr, br, padding=1), "down": lambda: T.downsample(xr, wr, br),
           "up": lambda: T.upsample(xr, wr, br)}[mode]()
    g = bf(W.uniform_tensor(tuple(ref.shape), 49))
    (ref * g).sum().backward()
    dev = backend.device
    xd, wd, bd = x.to(dev).requires_grad_(), wt.to(dev).requires_grad_(), b.to(dev).requires_grad_()
    y = vq.tae._from_video_cl(ops.conv3d(vq.tae._to_video_cl(xd, "bf16"), wd, bd, mode=mode), cout)
    assert rel(y, ref) < 2e-2
    (y * g.to(dev)).sum().backward()
    assert rel(xd.grad, xr.grad) < 2e-2 and rel(wd.grad, wr.grad) < 2e-2 and rel(bd.grad, br.grad) < 2e-2
