"""ORACLE — TEST INFRASTRUCTURE ONLY (imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg; never by the product package).

Plain PyTorch CPU fp32 restatement of the per-op arithmetic on the reference's hot path.  Every
function cites the reference lines (/root/reference) whose behaviour it restates.  Tensors are
NCHW like the reference.  Pinning: oracle/check_against_reference.py and tests/test_oracle.py compare
these restatements (and oracle/model_ref.py built on them) with the reference's own modules imported
from /root/reference in the build container, and with the fixtures under tests/golden/ generated
from the reference by tests/golden/make_golden.py.
"""
from __future__ import annotations

import contextlib

import torch
import torch.nn.functional as F

# ---------------------------------------------------------------------------------------------------------------------
# Emulation of the arithmetic the reference's GPU path runs in, on the CPU, for ERROR BUDGETS only: how far
# the reference's own step is from its fp32 CPU path is the yardstick for the HIP path's "ref" precision mode.
#   "fp32"  plain fp32 (the reference's CPU path — the oracle proper)
#   "tf32"  `torch.backends.cudnn.allow_tf32 = True` (vae_trainer.py:18-19) on an fp32 module: conv operands (x, w and, in the
#           backward, dy) rounded to a 10-bit mantissa, fp32 accumulation, fp32 results — encoder (:538), LPIPS, discriminator
#   "bf16"  `torch.autocast("cuda", dtype=torch.bfloat16)` (vae_trainer.py:453,623-624): conv operands and results in bf16,
#           FP32GroupNorm computes in fp32 and casts back (ae.py:45-53), swish on bf16 tensors — the decoder
_ARITH = "fp32"


@contextlib.contextmanager
def arith(mode):
    global _ARITH
    assert mode in ("fp32", "tf32", "bf16")
    prev, _ARITH = _ARITH, mode
    try:
        yield
    finally:
        _ARITH = prev


def _round_mantissa(t, keep_bits):
    """fp32 -> fp32 with `keep_bits` explicit mantissa bits, round-to-nearest-even (tf32: 10, bf16: 7)."""
    drop = 23 - keep_bits
    u = t.detach().contiguous().view(torch.int32)
    bias = ((u >> drop) & 1) + ((1 << (drop - 1)) - 1)
    return ((u + bias) >> drop << drop).view(torch.float32)


def _q(t):
    return _round_mantissa(t, 10 if _ARITH == "tf32" else 7)


class _RoundST(torch.autograd.Function):
    """Value rounded to the emulated type, gradient passed through (and rounded the same way: autocast's backward tensors
    have the forward's dtype)."""

    @staticmethod
    def forward(ctx, t, bits):
        ctx.bits = bits
        return _round_mantissa(t, bits)

    @staticmethod
    def backward(ctx, g):
        return _round_mantissa(g, ctx.bits), None


def _act(t):
    """An activation tensor as the emulated module stores it: bf16 under autocast, fp32 otherwise."""
    return _RoundST.apply(t, 7) if _ARITH == "bf16" else t


class _QuantConv(torch.autograd.Function):
    """conv2d whose three GEMMs see operands rounded to the emulated matmul input type, with fp32 accumulation."""

    @staticmethod
    def forward(ctx, x, w, b, stride, padding):
        xq, wq = _q(x), _q(w)
        ctx.save_for_backward(xq, wq)
        ctx.cfg = (stride, padding, b is not None, _ARITH)
        return F.conv2d(xq, wq, b if (b is None or _ARITH != "bf16") else _q(b), stride=stride, padding=padding)

    @staticmethod
    def backward(ctx, dy):
        xq, wq = ctx.saved_tensors
        stride, padding, has_b, mode = ctx.cfg
        with arith(mode):
            dyq = _q(dy)
        dx = torch.nn.grad.conv2d_input(xq.shape, wq, dyq, stride=stride, padding=padding)
        dw = torch.nn.grad.conv2d_weight(xq, wq.shape, dyq, stride=stride, padding=padding)
        db = dyq.sum((0, 2, 3)) if has_b else None
        return dx, dw, db, None, None


def swish(x):
    """ae.py:13-14."""
    if _ARITH == "bf16":
        return _act(x * _act(torch.sigmoid(x)))
    return x * torch.sigmoid(x)


def group_norm_fp32(x, gamma, beta, groups=32, eps=1e-6):
    """ae.py:41-53 FP32GroupNorm.forward (fp32 math, cast back)."""
    return _act(F.group_norm(x.float(), groups, gamma.float(), beta.float(), eps).type_as(x))


def conv2d(x, w, b=None, stride=1, padding=0):
    """StandardizedC2d = nn.Conv2d (ae.py:38)."""
    if _ARITH != "fp32":
        return _act(_QuantConv.apply(x, w, b, stride, padding))
    return F.conv2d(x, w, b, stride=stride, padding=padding)


def downsample(x, w, b):
    """ae.py:150-154: zero pad right/bottom by one, then 3x3 stride-2 conv without padding."""
    return F.conv2d(F.pad(x, (0, 1, 0, 1), mode="constant", value=0), w, b, stride=2, padding=0)


def upsample(x, w, b):
    """ae.py:164-166: nearest 2x, then 3x3 s1 p1 conv."""
    return F.conv2d(F.interpolate(x, scale_factor=2.0, mode="nearest"), w, b, stride=1, padding=1)


def resnet_block(x, p, prefix):
    """ae.py:124-140 ResnetBlock.forward; `p` is a state-dict-like mapping."""
    h = swish(group_norm_fp32(x, p[prefix + "norm1.weight"], p[prefix + "norm1.bias"]))
    h = conv2d(h, p[prefix + "conv1.weight"], p[prefix + "conv1.bias"], padding=1)
    h = swish(group_norm_fp32(h, p[prefix + "norm2.weight"], p[prefix + "norm2.bias"]))
    h = conv2d(h, p[prefix + "conv2.weight"], p[prefix + "conv2.bias"], padding=1)
    if prefix + "nin_shortcut.weight" in p:
        x = conv2d(x, p[prefix + "nin_shortcut.weight"], p[prefix + "nin_shortcut.bias"])
    return _act(x + h)


def attn_block(x, p, prefix, head_dim=64):
    """ae.py:56-93 AttnBlock.forward: x + proj_out(SDPA(qkv(GN(x)))), heads "b (h d) x y -> b h (x y) d"."""
    h_ = group_norm_fp32(x, p[prefix + "norm.weight"], p[prefix + "norm.bias"])
    qkv = F.conv2d(h_, p[prefix + "qkv.weight"])
    q, k, v = qkv.chunk(3, dim=1)
    b, c, hh, ww = q.shape
    nh = c // head_dim
    split = lambda t: t.reshape(b, nh, head_dim, hh * ww).transpose(2, 3)          # noqa: E731
    o = F.scaled_dot_product_attention(split(q), split(k), split(v))
    o = o.transpose(2, 3).reshape(b, c, hh, ww)
    return x + F.conv2d(o, p[prefix + "proj_out.weight"])


_DEC_LO = (-0.1768, 0.3536, 1.0607, 0.3536, -0.1768, 0.0000)      # utils.py:206-209
_DEC_HI = (0.0000, -0.0000, 0.3536, -0.7071, 0.3536, -0.0000)


def wavelet_transform(x):
    """utils.py:206-247 wavelet_transform_multi_channel: zero-pad 2, the four outer-product 6x6 filters
    (lo x lo, lo x hi, hi x lo, hi x hi with `a.unsqueeze(0) * b.unsqueeze(1)`), stride 2, per channel;
    output channel c*4 + f."""
    lo, hi = torch.tensor(_DEC_LO), torch.tensor(_DEC_HI)
    bank = torch.stack([lo.unsqueeze(0) * lo.unsqueeze(1), lo.unsqueeze(0) * hi.unsqueeze(1),
                        hi.unsqueeze(0) * lo.unsqueeze(1), hi.unsqueeze(0) * hi.unsqueeze(1)], 0).unsqueeze(1)
    b, c, h, w = x.shape
    padded = F.pad(x, (2, 2, 2, 2))
    out = torch.cat([F.conv2d(padded[:, k:k + 1], bank, stride=2) for k in range(c)], dim=1)
    return out.view(b, 4 * c, out.shape[2], out.shape[3])


def area_resize(x, size):
    """vae_trainer.py:531-533."""
    return F.interpolate(x, size=size, mode="area")


def scaling_layer(x, shift, scale):
    """utils.py:60-71 ScalingLayer.forward."""
    return (x - shift.view(1, -1, 1, 1)) / scale.view(1, -1, 1, 1)


def normalize_tensor(x, eps=1e-10):
    """utils.py:134-136."""
    return x / (torch.sqrt(torch.sum(x ** 2, dim=1, keepdim=True)) + eps)


def lpips_tap(f0, f1, w, mask=None):
    """One term of utils.py:44-57: spatial mean of the 1x1 'lin' conv (utils.py:85-88) over the
    squared difference of channel-normalised features; `mask` stands for nn.Dropout's scaled keep
    mask (utils.py:79-83, live in train mode)."""
    d = (normalize_tensor(f0) - normalize_tensor(f1)) ** 2
    if mask is not None:
        d = d * mask
    return F.conv2d(d, w.view(1, -1, 1, 1)).mean([2, 3], keepdim=True)


def gradnorm_backward(g, weight=1.0, world_norms=None):
    """vae_trainer.py:34-48: g * w / (mean over ranks of ||g||_2 + 1e-8)."""
    n = torch.norm(g)
    if world_norms is not None:
        n = torch.stack(list(world_norms)).mean()
    return weight * g / (n + 1e-8)


def gan_disc_loss(real, fake, disc_type="bce"):
    """vae_trainer.py:63-90 (returns tensors instead of .item() floats)."""
    if disc_type == "bce":
        lr = F.binary_cross_entropy_with_logits(real, torch.ones_like(real))
        lf = F.binary_cross_entropy_with_logits(fake, torch.zeros_like(fake))
    else:
        lr = F.relu(1 - real).mean()
        lf = F.relu(1 + fake).mean()
    acc = ((real > 0).sum() + (fake < 0).sum()).float() / (real.numel() + fake.numel())
    return (lr + lf) * 0.5, real.mean(), fake.mean(), acc


def adamw_step(p, g, m, v, step, lr, wd, beta1=0.9, beta2=0.95, eps=1e-8):
    """torch.optim.AdamW single-tensor update (vae_trainer.py:455-475 hyper-parameters)."""
    p = p * (1 - lr * wd)
    m = m + (g - m) * (1 - beta1)
    v = v * beta2 + (1 - beta2) * g * g
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = v.sqrt() / (bc2 ** 0.5) + eps
    p = p - (lr / bc1) * (m / denom)
    return p, m, v
