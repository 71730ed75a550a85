"""ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/ops_ref.py for the rules).

Functional CPU fp32 restatement of the reference's 3-D autoencoder (tae.py), driven by a plain state dict with the
reference's key names.  Tensors are NCTHW like the reference.  Pinned to the reference's own `tae.TVAE` (imported from
/root/reference in the build container) by tests/test_oracle.py and by tests/golden/tvae_*.npz
(tests/golden/make_golden.py).  Citations are file:line in /root/reference.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from .ops_ref import swish


def group_norm(x, gamma, beta):
    """nn.GroupNorm(32, C, eps=1e-6) on [B,C,T,H,W] (tae.py:19-21,66-67,72-73,165-167)."""
    return F.group_norm(x, 32, gamma, beta, 1e-6)


def resnet_block(x, p, pre):
    """tae.py:83-95."""
    h = F.conv3d(swish(group_norm(x, p[pre + "norm1.weight"], p[pre + "norm1.bias"])), p[pre + "conv1.weight"],
                 p[pre + "conv1.bias"], padding=1)
    h = F.conv3d(swish(group_norm(h, p[pre + "norm2.weight"], p[pre + "norm2.bias"])), p[pre + "conv2.weight"],
                 p[pre + "conv2.bias"], padding=1)
    if pre + "nin_shortcut.weight" in p:
        x = F.conv3d(x, p[pre + "nin_shortcut.weight"], p[pre + "nin_shortcut.bias"])
    return x + h


def attn_block(x, p, pre, num_heads=8):
    """tae.py:13-57: 8 heads of C/8 channels over the T*H*W tokens ("b (head d) t h w -> b head (t h w) d")."""
    qkv = F.conv3d(group_norm(x, p[pre + "norm.weight"], p[pre + "norm.bias"]), p[pre + "qkv.weight"])
    q, k, v = qkv.chunk(3, dim=1)
    b, c, t, h, w = q.shape
    d = c // num_heads
    tok = lambda u: u.reshape(b, num_heads, d, t * h * w).transpose(2, 3)          # noqa: E731
    s = torch.softmax(tok(q) @ tok(k).transpose(2, 3) / d ** 0.5, dim=-1)          # F.scaled_dot_product_attention
    o = (s @ tok(v)).transpose(2, 3).reshape(b, c, t, h, w)
    return x + F.conv3d(o, p[pre + "proj_out.weight"])


def downsample(x, w, b):
    """tae.py:102-107: one zero frame / row / column appended, then 3x3x3 stride 2 without padding."""
    return F.conv3d(F.pad(x, (0, 1, 0, 1, 0, 1)), w, b, stride=2)


def upsample(x, w, b):
    """tae.py:117-120: nearest 2x over (T, H, W), then 3x3x3 s1 p1."""
    return F.conv3d(F.interpolate(x, scale_factor=2.0, mode="nearest"), w, b, padding=1)


def _count(p, fmt):
    n = 0
    while fmt.format(n) in p:
        n += 1
    return n


def _middle(p, h, pre):
    h = resnet_block(h, p, pre + "mid.block_1.")
    h = attn_block(h, p, pre + "mid.attn_1.")
    return resnet_block(h, p, pre + "mid.block_2.")


def encoder(p, x, pre="encoder."):
    """tae.py:172-186."""
    h = F.conv3d(x, p[pre + "conv_in.weight"], p[pre + "conv_in.bias"], padding=1)
    for lvl in range(_count(p, pre + "down.{}.block.0.norm1.weight")):
        for b in range(_count(p, pre + f"down.{lvl}.block." + "{}.norm1.weight")):
            h = resnet_block(h, p, f"{pre}down.{lvl}.block.{b}.")
        if f"{pre}down.{lvl}.downsample.conv.weight" in p:
            h = downsample(h, p[f"{pre}down.{lvl}.downsample.conv.weight"], p[f"{pre}down.{lvl}.downsample.conv.bias"])
    h = _middle(p, h, pre)
    h = swish(group_norm(h, p[pre + "norm_out.weight"], p[pre + "norm_out.bias"]))
    return F.conv3d(h, p[pre + "conv_out.weight"], p[pre + "conv_out.bias"], padding=1)


def decoder(p, z, pre="decoder."):
    """tae.py:226-240."""
    h = F.conv3d(z, p[pre + "conv_in.weight"], p[pre + "conv_in.bias"], padding=1)
    h = _middle(p, h, pre)
    for lvl in reversed(range(_count(p, pre + "up.{}.block.0.norm1.weight"))):
        for b in range(_count(p, pre + f"up.{lvl}.block." + "{}.norm1.weight")):
            h = resnet_block(h, p, f"{pre}up.{lvl}.block.{b}.")
        if f"{pre}up.{lvl}.upsample.conv.weight" in p:
            h = upsample(h, p[f"{pre}up.{lvl}.upsample.conv.weight"], p[f"{pre}up.{lvl}.upsample.conv.bias"])
    h = swish(group_norm(h, p[pre + "norm_out.weight"], p[pre + "norm_out.bias"]))
    return F.conv3d(h, p[pre + "conv_out.weight"], p[pre + "conv_out.bias"], padding=1)


def diagonal_gaussian(z, noise):
    """tae.py:249-257 with the randn_like draw passed in."""
    mean, logvar = torch.chunk(z, 2, dim=1)
    return mean + torch.exp(0.5 * logvar.clamp(min=-3)) * noise


def tvae_forward(p, x, noise):
    """tae.py:283-287 -> (reconstruction, z)."""
    z = encoder(p, x)
    return decoder(p, diagonal_gaussian(z, noise)), z
