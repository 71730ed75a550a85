"""MI355X-native VAE — drop-in for the module surface of the reference's ae.py.

What is kept (because `vae_trainer.py`, checkpoints and the HF weights depend on it): class names,
constructor signatures, attribute / parameter names and OIHW fp32 shapes, and the order in which
parameters are created and initialised (ae.py:97-122, 171-237, 261-316), so that
`torch.manual_seed(s)` yields the same initial weights as the reference and `state_dict()`s are
interchangeable.  What is different: nothing here calls a PyTorch compute op.  Activations live in
NHWC (bf16, or fp32 for the parity mode) between the NCHW-fp32 boundaries of Encoder / Decoder and
every layer is a libvqhip kernel launched through `ops` (GroupNorm+swish fused, residual add in
the conv epilogue, Downsample's pad and Upsample's nearest-2x folded into the conv gather).
"""
from __future__ import annotations

import math

from torch import Tensor, nn

from . import ops

_GN = dict(num_groups=32, eps=1e-6, affine=True)


class StandardizedC2d(nn.Conv2d):
    """Holds nn.Conv2d parameters (reference: `StandardizedC2d = nn.Conv2d`, ae.py:38); forward is the
    implicit-GEMM MFMA kernel on an NHWC tensor."""

    def forward(self, x: Tensor, **kw) -> Tensor:
        if self.dilation != (1, 1) or self.groups != 1 or self.padding_mode != "zeros" or self.stride[0] != self.stride[1]:
            raise NotImplementedError("only dense, undilated, square-stride convolutions are on the HIP path")
        return ops.conv2d(x, self.weight, self.bias, stride=self.stride[0], pad=tuple(self.padding),
                          split=ops.split_for(x), **kw)


class FP32GroupNorm(nn.GroupNorm):
    """ae.py:41-53 (+ swish ae.py:13-14 when silu=True): fp32 statistics whatever the storage dtype."""

    def forward(self, x: Tensor, silu: bool = False, fork: bool = False) -> Tensor:
        return ops.group_norm_silu(x, self.weight, self.bias, self.num_groups, self.eps, silu, fork)


class AttnBlock(nn.Module):
    """ae.py:56-93: x + proj_out(SDPA(qkv(GN(x)))) over the H*W tokens, heads of 64 channels.  Unreachable in the
    reference at HEAD (`--do_attn True` raises in Encoder.__init__ because these convs have no bias);
    `_finish_init` guards that, so `use_attn=True` works here."""

    def __init__(self, in_channels: int):
        super().__init__()
        self.in_channels, self.head_dim = in_channels, 64
        self.num_heads = in_channels // self.head_dim
        self.norm = FP32GroupNorm(num_channels=in_channels, **_GN)
        self.qkv = StandardizedC2d(in_channels, 3 * in_channels, kernel_size=1, bias=False)
        self.proj_out = StandardizedC2d(in_channels, in_channels, kernel_size=1, bias=False)
        nn.init.normal_(self.proj_out.weight, std=0.2 / math.sqrt(in_channels))

    def attention(self, h_: Tensor) -> Tensor:
        return ops.attention(self.qkv(self.norm(h_)))          # GN without swish (ae.py:75)

    def forward(self, x):
        h, x = self.norm(x, fork=True)                         # (the skip gradient rejoins inside the GroupNorm backward kernel)
        return self.proj_out(ops.attention(self.qkv(h)), residual=x)    # the residual add rides in the conv epilogue


class ResnetBlock(nn.Module):
    """ae.py:96-140:  S(x) + conv2(swish(GN2(conv1(swish(GN1(x))))))  with S = identity or 1x1 conv."""

    def __init__(self, in_channels: int, out_channels: int):
        super().__init__()
        out_channels = out_channels or in_channels
        self.in_channels, self.out_channels = in_channels, out_channels
        self.norm1 = FP32GroupNorm(num_channels=in_channels, **_GN)
        self.conv1 = StandardizedC2d(in_channels, out_channels, 3, 1, 1)
        self.norm2 = FP32GroupNorm(num_channels=out_channels, **_GN)
        self.conv2 = StandardizedC2d(out_channels, out_channels, 3, 1, 1)
        if in_channels != out_channels:
            self.nin_shortcut = StandardizedC2d(in_channels, out_channels, 1, 1, 0)
        nn.init.normal_(self.conv2.weight, std=1e-4 / out_channels)     # ae.py:119-121
        nn.init.zeros_(self.conv2.bias)

    def forward(self, x):
        # one autograd node for the whole block: residual add in conv2's epilogue, skip gradient folded
        # into the GroupNorm backward kernel (ops._ResnetBlock)
        return ops.resnet_block(x, self.norm1, self.conv1, self.norm2, self.conv2, getattr(self, "nin_shortcut", None))


class Downsample(nn.Module):
    """ae.py:143-154.  F.pad(x,(0,1,0,1)) is not materialised: the missing bottom/right taps fail
    the kernel's bounds check."""

    def __init__(self, in_channels: int):
        super().__init__()
        self.conv = StandardizedC2d(in_channels, in_channels, 3, 2, 0)

    def forward(self, x):
        # the next level's first norm1 reads this output: let the conv epilogue reduce its GroupNorm statistics (ops.conv_fwd_raw)
        return self.conv(x, out_hw=((x.shape[1] - 2) // 2 + 1, (x.shape[2] - 2) // 2 + 1), gn=(_GN["num_groups"], _GN["eps"]))


class Upsample(nn.Module):
    """ae.py:157-167.  interpolate(2x, nearest) is not materialised: the gather reads (y>>1, x>>1)."""

    def __init__(self, in_channels: int):
        super().__init__()
        self.conv = StandardizedC2d(in_channels, in_channels, 3, 1, 1)

    def forward(self, x):
        return self.conv(x, up=2)


class _Level(nn.Module):
    """One resolution level: `.block` (ResnetBlocks), `.attn` (empty, as in the reference) and an
    optional `.downsample` / `.upsample` (names as in ae.py:207-222, 291-304)."""

    def __init__(self, widths: list[tuple[int, int]]):
        super().__init__()
        self.block = nn.ModuleList(ResnetBlock(i, o) for i, o in widths)
        self.attn = nn.ModuleList()

    def run(self, h):
        for blk in self.block:
            h = blk(h)
        return h


def _middle(width: int, use_attn: bool) -> nn.Module:
    mid = nn.Module()
    mid.block_1 = ResnetBlock(width, width)
    mid.attn_1 = AttnBlock(width) if use_attn else nn.Identity()
    mid.block_2 = ResnetBlock(width, width)
    return mid


def _finish_init(root: nn.Module):
    """ae.py:233-237 / 312-316: zero every conv and GroupNorm bias (guarded: fixes SURVEY F4)."""
    for m in root.modules():
        if isinstance(m, (StandardizedC2d, nn.GroupNorm)) and m.bias is not None:
            nn.init.zeros_(m.bias)


class Encoder(nn.Module):
    """ae.py:170-257.  [B,in_channels,H,W] fp32 -> z [B,z_channels,H/f,W/f] fp32 (mean only)."""

    def __init__(self, resolution: int, in_channels: int, ch: int, ch_mult: list[int], num_res_blocks: int,
                 z_channels: int, use_attn: bool = True, use_wavelet: bool = False):
        super().__init__()
        self.ch, self.resolution, self.in_channels, self.z_channels = ch, resolution, in_channels, z_channels
        self.num_resolutions, self.num_res_blocks, self.use_wavelet = len(ch_mult), num_res_blocks, use_wavelet
        if use_wavelet:
            # ae.py:189-194: the wavelet front-end halves the resolution and quadruples the channels; conv_in is twice
            # as wide and `ch_mult[0] *= 2` mutates the CALLER's list — VAE relies on it for the decoder.
            self.conv_in = StandardizedC2d(4 * in_channels, ch * 2, 3, 1, 1)
            ch_mult[0] *= 2
        else:
            self.conv_in = StandardizedC2d(in_channels, ch, 3, 1, 1)
        self.in_ch_mult = (2 if use_wavelet else 1,) + tuple(ch_mult)
        self.down = nn.ModuleList()
        width = ch
        for lvl, mult in enumerate(ch_mult):
            w_in, w_out = ch * self.in_ch_mult[lvl], ch * mult
            stage = _Level([(w_in if k == 0 else w_out, w_out) for k in range(num_res_blocks)])
            if lvl != len(ch_mult) - 1 and not (use_wavelet and lvl == 0):      # ae.py:217-219
                stage.downsample = Downsample(w_out)
            self.down.append(stage)
            width = w_out
        self.mid = _middle(width, use_attn)
        self.norm_out = FP32GroupNorm(num_channels=width, **_GN)
        self.conv_out = StandardizedC2d(width, z_channels, 3, 1, 1)
        _finish_init(self)
        self.precision = None   # None -> ops.default_precision() at call time

    def forward(self, x) -> Tensor:
        prec = ops.resolve_precision(self.precision)
        with ops.region(prec):                 # every op below belongs to this stack (its loss scale, in the fp16 mode)
            h = ops.wavelet_to_nhwc(x, prec) if self.use_wavelet else ops.to_nhwc(x, prec)
            h = self.conv_in(h, gn=(_GN["num_groups"], _GN["eps"]))
            for stage in self.down:
                h = stage.run(h)
                if hasattr(stage, "downsample"):
                    h = stage.downsample(h)
            h = self.mid.block_2(self.mid.attn_1(self.mid.block_1(h)))
            h = self.conv_out(self.norm_out(h, silu=True))
            return ops.to_nchw(h, self.z_channels)


class Decoder(nn.Module):
    """ae.py:260-333.  z [B,z_channels,h,w] fp32 -> [B,out_ch,h*f,w*f] fp32."""

    def __init__(self, ch: int, out_ch: int, ch_mult: list[int], num_res_blocks: int, in_channels: int,
                 resolution: int, z_channels: int, use_attn: bool = True):
        super().__init__()
        levels = len(ch_mult)
        self.ch, self.out_ch, self.resolution, self.in_channels = ch, out_ch, resolution, in_channels
        self.num_resolutions, self.num_res_blocks = levels, num_res_blocks
        self.ffactor = 2 ** (levels - 1)
        width = ch * ch_mult[-1]
        self.z_shape = (1, z_channels, resolution // self.ffactor, resolution // self.ffactor)
        self.conv_in = StandardizedC2d(z_channels, width, 3, 1, 1)
        self.mid = _middle(width, use_attn)
        stages = []
        for lvl in range(levels - 1, -1, -1):          # deepest first, exactly the reference's creation order
            w_out = ch * ch_mult[lvl]
            stage = _Level([(width if k == 0 else w_out, w_out) for k in range(num_res_blocks + 1)])
            if lvl != 0:
                stage.upsample = Upsample(w_out)
            stages.insert(0, stage)
            width = w_out
        self.up = nn.ModuleList(stages)
        self.norm_out = FP32GroupNorm(num_channels=width, **_GN)
        self.conv_out = StandardizedC2d(width, out_ch, 3, 1, 1)
        _finish_init(self)
        self.precision = None

    def forward(self, z) -> Tensor:
        prec = ops.resolve_precision(self.precision)
        with ops.region(prec):
            h = self.conv_in(ops.to_nhwc(z, prec), gn=(_GN["num_groups"], _GN["eps"]))
            h = self.mid.block_2(self.mid.attn_1(self.mid.block_1(h)))
            for stage in reversed(self.up):
                h = stage.run(h)
                if hasattr(stage, "upsample"):
                    h = stage.upsample(h)
            h = self.conv_out(self.norm_out(h, silu=True))
            return ops.to_nchw(h, self.out_ch)


class DiagonalGaussian(nn.Module):
    """ae.py:336-348 computes `mean * (1 + 0.00 * randn_like(mean))`: the std literal is 0.00, so the
    regulariser is the identity in value and gradient.  We return the mean and draw no noise."""

    def __init__(self, sample: bool = True, chunk_dim: int = 1):
        super().__init__()
        self.sample, self.chunk_dim = sample, chunk_dim

    def forward(self, z) -> Tensor:
        return z


class VAE(nn.Module):
    """ae.py:351-392: `.encoder`, `.reg`, `.decoder` are called individually by the trainer."""

    def __init__(self, resolution, in_channels, ch, out_ch, ch_mult, num_res_blocks, z_channels, use_attn,
                 decoder_also_perform_hr, use_wavelet):
        super().__init__()
        ch_mult = list(ch_mult)      # private copy; the Encoder doubles ch_mult[0] in place when use_wavelet (ae.py:194),
        self.encoder = Encoder(resolution, in_channels, ch, ch_mult, num_res_blocks, z_channels,   # and the decoder
                               use_attn=use_attn, use_wavelet=use_wavelet)                          # sees that (F13)
        dec_mult = ch_mult + [4] if decoder_also_perform_hr else ch_mult                            # ae.py:381
        self.decoder = Decoder(ch, out_ch, dec_mult, num_res_blocks, in_channels, resolution, z_channels,
                               use_attn=use_attn)
        self.reg = DiagonalGaussian()

    def set_precision(self, precision) -> "VAE":
        self.encoder.precision = self.decoder.precision = ops.resolve_precision(precision)
        return self

    def forward(self, x):
        z = self.encoder(x)
        return self.decoder(self.reg(z)), z
