"""MI355X-native 3-D (video) autoencoder — drop-in for the module surface of the reference's tae.py.

Same contract as `ae.py` here: class names, constructor signatures, parameter names, nn.Conv3d / nn.GroupNorm
parameter shapes ([O,I,3,3,3] fp32, OIDHW) and creation order are the reference's (tae.py:13-273), so
`torch.manual_seed(s)` gives the same initial weights and `state_dict()`s are interchangeable; nothing here calls a
PyTorch compute kernel for the layers.  Activations are channels-last video tensors [N,T,H,W,C] (C padded to 8; bf16,
or fp32 in the parity mode) between the NCTHW-fp32 boundaries of Encoder / Decoder:

  * 3x3x3 convolutions = three temporal taps of the 2-D implicit-GEMM MFMA kernel over runs of frames, accumulated in
    place through the conv epilogue (ops.conv3d); 1x1x1 convolutions = the 2-D 1x1 kernel over all N*T frames;
  * GroupNorm(32)+swish = the 2-D kernel with T*H*W as the pixel axis (statistics per sample and group over all frames,
    like nn.GroupNorm on a 5-D tensor);
  * the bottleneck AttnBlock = ops.attention over the T*H*W tokens with 8 heads of C/8 channels (tae.py:17-53).

Unlike ae.py's, tae.py's DiagonalGaussian really samples (tae.py:243-252); the few elementwise operations on the tiny
latent are left to torch, with an optional `noise` argument so that tests can pin the draw.
"""
from __future__ import annotations

import math

import torch
from torch import Tensor, nn

from . import ops

_GN = dict(num_groups=32, eps=1e-6, affine=True)


class Conv3d(nn.Conv3d):
    """nn.Conv3d parameters; forward on a channels-last [N,T,H,W,C] activation through the HIP kernels.
    `mode`: "same" (k3 s1 p1), "down" (k3 s2 p0 after the (0,1) zero pad of tae.py:103-104), "up" (nearest 2x, then same)."""

    def forward(self, x: Tensor, residual: Tensor | None = None, mode: str = "same") -> Tensor:
        k = self.kernel_size
        if self.dilation != (1, 1, 1) or self.groups != 1 or self.padding_mode != "zeros" or k[0] != k[1] or k[1] != k[2]:
            raise NotImplementedError("only dense, undilated, cubic convolutions are on the HIP path")
        if k[0] == 1:
            if self.stride != (1, 1, 1) or self.padding != (0, 0, 0):
                raise NotImplementedError("1x1x1 convolutions: stride 1, no padding")
            n, t = x.shape[:2]
            res = residual.flatten(0, 1) if residual is not None else None
            o, i = self.weight.shape[:2]
            y = ops.conv2d(x.flatten(0, 1), self.weight.view(o, i, 1, 1), self.bias, residual=res, split=ops.split_for(x))
            return y.view((n, t) + y.shape[1:])
        if k[0] != 3:
            raise NotImplementedError("kernel sizes 1 and 3 only")
        want = {"same": ((1, 1, 1), (1, 1, 1)), "up": ((1, 1, 1), (1, 1, 1)), "down": ((2, 2, 2), (0, 0, 0))}[mode]
        if (self.stride, self.padding) != want:
            raise NotImplementedError(f"conv3d mode '{mode}' expects stride/padding {want}")
        return ops.conv3d(x, self.weight, self.bias, residual=residual, mode=mode)


class GroupNorm3d(nn.GroupNorm):
    """nn.GroupNorm on a 5-D tensor (+ swish tae.py:9-10 when silu=True), fp32 statistics over (C/32, T, H, W)."""

    def forward(self, x: Tensor, silu: bool = False, fork: bool = False):
        n, t, h, w, c = x.shape
        y = ops.group_norm_silu(x.reshape(n, t * h, w, c), self.weight, self.bias, self.num_groups, self.eps, silu, fork)
        if fork:                                               # (GN(x), x): see ops._GroupNormSiluFork
            return y[0].view(n, t, h, w, c), y[1].view(n, t, h, w, c)
        return y.view(n, t, h, w, c)


def swish(x: Tensor) -> Tensor:
    """tae.py:9-10 (kept for API parity; inside the model it is fused into the GroupNorm kernel)."""
    return x * torch.sigmoid(x)


class AttnBlock(nn.Module):
    """tae.py:13-57: x + proj_out(SDPA(qkv(GN(x)))) over the T*H*W tokens, 8 heads of C/8 channels."""

    def __init__(self, in_channels: int):
        super().__init__()
        self.in_channels = in_channels
        self.num_heads = 8
        self.head_dim = in_channels // self.num_heads
        self.norm = GroupNorm3d(num_channels=in_channels, **_GN)
        self.qkv = Conv3d(in_channels, in_channels * 3, kernel_size=1, bias=False)
        self.proj_out = Conv3d(in_channels, in_channels, kernel_size=1, bias=False)
        nn.init.normal_(self.proj_out.weight, std=0.2 / math.sqrt(in_channels))

    def attention(self, h_: Tensor) -> Tensor:
        return ops.attention(self.qkv(self.norm(h_)), self.head_dim)

    def forward(self, x: Tensor) -> Tensor:
        h, x = self.norm(x, fork=True)                             # (the skip gradient rejoins inside the GroupNorm backward kernel)
        return self.proj_out(ops.attention(self.qkv(h), self.head_dim), residual=x)    # the residual add rides in the conv epilogue


class ResnetBlock(nn.Module):
    """tae.py:60-95:  S(x) + conv2(swish(GN2(conv1(swish(GN1(x))))))  with S = identity or 1x1x1 conv."""

    def __init__(self, in_channels: int, out_channels: int = None):
        super().__init__()
        self.in_channels = in_channels
        out_channels = in_channels if out_channels is None else out_channels
        self.out_channels = out_channels
        self.norm1 = GroupNorm3d(num_channels=in_channels, **_GN)
        self.conv1 = Conv3d(in_channels, out_channels, kernel_size=3, stride=1, padding=1)
        self.norm2 = GroupNorm3d(num_channels=out_channels, **_GN)
        self.conv2 = Conv3d(out_channels, out_channels, kernel_size=3, stride=1, padding=1)
        if self.in_channels != self.out_channels:
            self.nin_shortcut = Conv3d(in_channels, out_channels, kernel_size=1, stride=1, padding=0)

    def forward(self, x):
        h, x = self.norm1(x, silu=True, fork=True)
        h = self.conv1(h)
        skip = self.nin_shortcut(x) if self.in_channels != self.out_channels else x
        return self.conv2(self.norm2(h, silu=True), residual=skip)


class Downsample(nn.Module):
    """tae.py:98-107.  The (0,1) pads are not materialised: the missing bottom / right taps fail the 2-D kernel's bounds
    check and the appended zero frame is simply not launched."""

    def __init__(self, in_channels: int):
        super().__init__()
        self.conv = Conv3d(in_channels, in_channels, kernel_size=3, stride=2, padding=0)

    def forward(self, x: Tensor):
        return self.conv(x, mode="down")


class Upsample(nn.Module):
    """tae.py:110-120: nearest 2x in T, H and W, then 3x3x3."""

    def __init__(self, in_channels: int):
        super().__init__()
        self.conv = Conv3d(in_channels, in_channels, kernel_size=3, stride=1, padding=1)

    def forward(self, x: Tensor):
        return self.conv(x, mode="up")


def _to_video_cl(x: Tensor, precision) -> Tensor:
    """[B,C,T,H,W] fp32 -> channels-last [B,T,H,W,pad8(C)] in the storage dtype (vq_nchw_to_nhwc with (T, H*W) as the image)."""
    b, c, t, h, w = x.shape
    y = ops.to_nhwc(x.reshape(b, c, t, h * w), precision)
    return y.view(b, t, h, w, y.shape[-1])


def _from_video_cl(x: Tensor, c: int) -> Tensor:
    b, t, h, w, cp = x.shape
    return ops.to_nchw(x.reshape(b, t, h * w, cp), c).view(b, c, t, h, w)


class _Level(nn.Module):
    def __init__(self, widths):
        super().__init__()
        self.block = nn.ModuleList(ResnetBlock(i, o) for i, o in widths)
        self.attn = nn.ModuleList()           # always empty in the reference (tae.py:145,212)

    def run(self, h):
        for k, blk in enumerate(self.block):
            h = blk(h)
            if len(self.attn) > 0:
                h = self.attn[k](h)
        return h


def _middle(width: int) -> nn.Module:
    mid = nn.Module()
    mid.block_1 = ResnetBlock(width, width)
    mid.attn_1 = AttnBlock(width)
    mid.block_2 = ResnetBlock(width, width)
    return mid


class Encoder(nn.Module):
    """tae.py:123-186.  [B,in_channels,T,H,W] fp32 -> [B,2*z_channels,T/f,H/f,W/f] fp32 (mean | logvar)."""

    def __init__(self, resolution: int, in_channels: int, ch: int, ch_mult: list[int], num_res_blocks: int, z_channels: int):
        super().__init__()
        self.ch, self.resolution, self.in_channels = ch, resolution, in_channels
        self.num_resolutions, self.num_res_blocks = len(ch_mult), num_res_blocks
        self.z_channels = z_channels
        self.conv_in = Conv3d(in_channels, ch, kernel_size=3, stride=1, padding=1)
        in_ch_mult = (1,) + tuple(ch_mult)
        self.down = nn.ModuleList()
        width = ch
        for lvl, mult in enumerate(ch_mult):
            w_in, w_out = ch * in_ch_mult[lvl], ch * mult
            stage = _Level([(w_in if k == 0 else w_out, w_out) for k in range(num_res_blocks)])
            if lvl != self.num_resolutions - 1:
                stage.downsample = Downsample(w_out)
            self.down.append(stage)
            width = w_out
        self.mid = _middle(width)
        self.norm_out = GroupNorm3d(num_channels=width, **_GN)
        self.conv_out = Conv3d(width, 2 * z_channels, kernel_size=3, stride=1, padding=1)
        self.precision = None

    def forward(self, x: Tensor) -> Tensor:
        h = self.conv_in(_to_video_cl(x, self.precision))
        for stage in self.down:
            h = stage.run(h)
            if hasattr(stage, "downsample"):
                h = stage.downsample(h)
        h = self.mid.block_2(self.mid.attn_1(self.mid.block_1(h)))
        h = self.conv_out(self.norm_out(h, silu=True))
        return _from_video_cl(h, 2 * self.z_channels)


class Decoder(nn.Module):
    """tae.py:189-240.  z [B,z_channels,t,h,w] fp32 -> [B,out_ch,t*f,h*f,w*f] fp32."""

    def __init__(self, ch: int, out_ch: int, ch_mult: list[int], num_res_blocks: int, in_channels: int, resolution: int,
                 z_channels: int):
        super().__init__()
        levels = len(ch_mult)
        self.ch, self.out_ch, self.resolution, self.in_channels = ch, out_ch, resolution, in_channels
        self.num_resolutions, self.num_res_blocks = levels, num_res_blocks
        self.ffactor = 2 ** (levels - 1)
        width = ch * ch_mult[-1]
        res = resolution // self.ffactor
        self.z_shape = (1, z_channels, res, res, res)
        self.conv_in = Conv3d(z_channels, width, kernel_size=3, stride=1, padding=1)
        self.mid = _middle(width)
        stages = []
        for lvl in range(levels - 1, -1, -1):           # deepest first: the reference's creation order (tae.py:210-224)
            w_out = ch * ch_mult[lvl]
            stage = _Level([(width if k == 0 else w_out, w_out) for k in range(num_res_blocks + 1)])
            if lvl != 0:
                stage.upsample = Upsample(w_out)
            stages.insert(0, stage)
            width = w_out
        self.up = nn.ModuleList(stages)
        self.norm_out = GroupNorm3d(num_channels=width, **_GN)
        self.conv_out = Conv3d(width, out_ch, kernel_size=3, stride=1, padding=1)
        self.precision = None

    def forward(self, z: Tensor) -> Tensor:
        h = self.conv_in(_to_video_cl(z, self.precision))
        h = self.mid.block_2(self.mid.attn_1(self.mid.block_1(h)))
        for stage in reversed(self.up):
            h = stage.run(h)
            if hasattr(stage, "upsample"):
                h = stage.upsample(h)
        h = self.conv_out(self.norm_out(h, silu=True))
        return _from_video_cl(h, self.out_ch)


class DiagonalGaussian(nn.Module):
    """tae.py:243-257: mean + exp(0.5 * max(logvar, -3)) * eps.  `noise` (optional) replaces the randn_like draw."""

    def __init__(self, sample: bool = True, chunk_dim: int = 1):
        super().__init__()
        self.sample, self.chunk_dim = sample, chunk_dim

    def forward(self, z: Tensor, noise: Tensor | None = None) -> Tensor:
        mean, logvar = torch.chunk(z, 2, dim=self.chunk_dim)
        if not self.sample:
            return mean
        std = torch.exp(0.5 * logvar.clamp(min=-3))
        return mean + std * (torch.randn_like(mean) if noise is None else noise)


class TVAE(nn.Module):
    """tae.py:260-288; forward(x) -> (reconstruction, z) with z the encoder output (mean | logvar)."""

    def __init__(self, resolution, in_channels, ch, out_ch, ch_mult, num_res_blocks, z_channels):
        super().__init__()
        self.encoder = Encoder(resolution=resolution, in_channels=in_channels, ch=ch, ch_mult=ch_mult,
                               num_res_blocks=num_res_blocks, z_channels=z_channels)
        self.decoder = Decoder(resolution=resolution, in_channels=in_channels, ch=ch, out_ch=out_ch, ch_mult=ch_mult,
                               num_res_blocks=num_res_blocks, z_channels=z_channels)
        self.reg = DiagonalGaussian()

    def set_precision(self, precision) -> "TVAE":
        self.encoder.precision = self.decoder.precision = ops.resolve_precision(precision)
        return self

    def forward(self, x: Tensor, noise: Tensor | None = None):
        z = self.encoder(x)
        return self.decoder(self.reg(z, noise)), z
