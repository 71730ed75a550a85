"""torch.autograd wrappers over the libvqhip C ABI.

Internal activation layout is NHWC ("pixel-major") with channels padded to a multiple of 8, in
bf16 (throughput mode) or fp32 (parity mode, bf16x3-split MFMA).  Every Function below only moves
pointers: all arithmetic happens in the HIP kernels (vqgan-training_amd/csrc/*.hip).
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass

import weakref

import torch

import contextlib
import math
import threading

from . import _lib
from ._lib import VQ_BF16, VQ_F16, VQ_F16X2, VqConvDesc, VqGnBwdFuse, lib, ptr, stream_of, dtype_code, workspace


# ----------------------------------------------------------------------------- precision modes
@dataclass(eq=False)
class Precision:
    """Storage dtype of activations + MFMA operand split (include/vqhip.h VqConvDesc.split).

    fp16 (the reference-precision mode, include/vqhip.h VQ_F16): binary16 storage and MFMA operands — TF32's 10-bit mantissa —
    with fp32 accumulation.  Activations are stored as they are, packed weights times a per-tensor power of two measured on
    the device (vq_pack_weight_*), gradient tensors times `grad_scale` (a power of two: the loss scale of the module stack
    this object is attached to; every parameter gradient and every gradient leaving the stack is divided by it again, all
    exact).  An instance is the unit of scaling: apply_precision_policy gives the encoder, LPIPS and the discriminator one
    each (`fp16_region`), calibrated from the measured gradient maxima (GradMonitor)."""
    name: str
    dtype: torch.dtype
    split: int
    grad_scale: float = 1.0
    region: str = ""
    # fp16 stacks: the device counters the kernels report saturated / vanished binary16 stores to (include/vqhip.h "range
    # events"): int32 [>= 2] = {waves that clipped a value, waves whose values all flushed to zero} since the owner last cleared them.
    # None (direct op calls, tests): nothing is counted.  VAETrainStep owns the tensor and hands every stack a row of it.
    events: "torch.Tensor | None" = None

    def gs(self) -> float:
        return self.grad_scale if self.dtype in HALF_RANGE else 1.0

    def half_range(self) -> bool:
        """Stored values live in binary16's range: gradients carry the loss scale, stores report range events."""
        return self.dtype in HALF_RANGE


# Storage dtype of the "f16x3" arithmetic (include/vqhip.h VQ_F16X2): every value as TWO binary16 numbers hi + lo (22 significand
# bits), per 8 channels 16 bytes of hi then 16 bytes of lo.  torch.complex32 is only the CARRIER — 4 bytes per element, so shapes,
# strides and autograd plumbing are those of the other storage types; torch itself never computes on these tensors.
X2 = torch.complex32
HALF_RANGE = (torch.float16, X2)
import warnings                       # noqa: E402
with warnings.catch_warnings():       # torch warns ONCE per process at the first complex32 allocation ("ComplexHalf support is
    warnings.simplefilter("ignore")   # experimental"): spend that one warning here, on a carrier that is only allocated, viewed and
    torch.empty(0, dtype=X2)          # copied — no process-wide warning filter is installed


BF16 = Precision("bf16", torch.bfloat16, 1)        # bf16 storage, bf16 MFMA, fp32 accumulate
FP32 = Precision("fp32", torch.float32, 1)         # fp32 storage, operands rounded to bf16
FP32X3 = Precision("fp32x3", torch.float32, 3)     # fp32 storage, 3-term bf16 split: operands to 16 mantissa bits (~2^-16 per product)
# fp32 storage, three bf16 pieces per operand (all 24 mantissa bits) and the six products down to 2^-16: fp32-EXACT products,
# fp32 accumulation — the arithmetic of the CPU reference itself (north_star: "within stated fp32 tolerance"); twice fp32x3's MFMAs
FP32X6 = Precision("fp32x6", torch.float32, 6)
# binary16 storage + MFMA, scaled; this process-wide instance serves direct op calls (gradients of order one: 2^8 leaves the
# 4-sigma x fan-in gain of a data gradient inside 65504); module stacks get their own, calibrated, objects (fp16_region)
FP16 = Precision("fp16", torch.float16, 1, grad_scale=2.0 ** 8, region="default")
# two binary16 pieces per operand, three binary16 MFMAs per product (hi*hi + hi*lo + lo*hi, fp32 accumulate): ~2^-21 per product — the
# fp32-tolerance arithmetic of the TUNED kernels (LDS-DMA tiles, nine-tap / patch-staged kernels), where fp32x3 / fp32x6 run on the
# generic register-staged kernel.  Scaled like fp16 (weights times s_w, gradients times the stack's loss scale).
F16X3 = Precision("f16x3", X2, 1, grad_scale=2.0 ** 8, region="default")
_PRECISIONS = {p.name: p for p in (BF16, FP32, FP32X3, FP32X6, FP16, F16X3)}
_default_precision = BF16


def fp16_region(region: str, grad_scale: float = 2.0 ** 12) -> Precision:
    """A fresh fp16 precision object = one loss-scale domain (e.g. "encoder", "lpips", "disc")."""
    assert grad_scale > 0 and math.frexp(grad_scale)[0] == 0.5, "the loss scale must be a power of two"
    return Precision("fp16", torch.float16, 1, grad_scale=float(grad_scale), region=region)


def f16x3_region(region: str, grad_scale: float = 2.0 ** 12) -> Precision:
    """A fresh f16x3 precision object = one loss-scale domain, like fp16_region."""
    assert grad_scale > 0 and math.frexp(grad_scale)[0] == 0.5, "the loss scale must be a power of two"
    return Precision("f16x3", X2, 1, grad_scale=float(grad_scale), region=region)


# The stack a tensor belongs to is not visible from the tensor: the top-level modules (Encoder / Decoder / LPIPS /
# PatchDiscriminator forward) declare it for the ops they call; autograd nodes remember it for their backward.
_tls = threading.local()


@contextlib.contextmanager
def region(prec, backward: bool = False):
    """Declare the stack of the ops called inside.  `backward=True` (the autograd nodes' backward passes): range events of binary16
    stores go to the stack's GRADIENT counters — the ones the optimizers' device-side skip looks at — instead of the forward ones."""
    prev = (getattr(_tls, "prec", None), getattr(_tls, "bwd", False))
    _tls.prec, _tls.bwd = prec, backward
    try:
        yield
    finally:
        _tls.prec, _tls.bwd = prev


def precision_of(x: torch.Tensor) -> Precision:
    """Precision object of the stack `x` flows through: the declared region's when the storage dtype matches, else the
    process-wide instance of that dtype."""
    cur = getattr(_tls, "prec", None)
    if cur is not None and cur.dtype == x.dtype:
        return cur
    if x.dtype == torch.float16:
        return FP16
    if x.dtype == X2:
        return F16X3
    if x.dtype == torch.bfloat16:
        return BF16
    return {3: FP32X3, 6: FP32X6}.get(_fp32_split, FP32)


def _events():
    """Device pointer of the range-event counters of the stack whose ops are running (forward: the region the module declared;
    backward: the region the autograd node re-declares), or None."""
    cur = getattr(_tls, "prec", None)
    if cur is None or cur.dtype not in HALF_RANGE or cur.events is None:
        return None
    # a stack's counter row (VAETrainStep.bind_range_events): [0:6] = gradient stores (window: saturated, flushed, headroom; then their
    # totals — a clipped one drops the optimizer step and a lower loss scale fixes it), [6:12] = forward stores (activations are stored
    # unscaled: no loss scale can help, so they must not gate the optimizer — they are logged and escalated instead).  Shorter rows
    # (probes: >= 3 counters) share one set.
    fwd = (not getattr(_tls, "bwd", False)) and cur.events.numel() >= 12
    return C.c_void_p(cur.events.data_ptr() + (24 if fwd else 0))


def _op(x: torch.Tensor) -> int:
    """MFMA operand type of the kernels that consume `x` (include/vqhip.h: vq_pack_weight_* op_dtype)."""
    return VQ_F16 if x.dtype == torch.float16 else (VQ_F16X2 if x.dtype == X2 else VQ_BF16)


def _adev(scale):
    """VqConvDesc.alpha_dev of a launch that consumes a VQ_F16 packed weight: the 1/s_w slot of its scale record."""
    return None if scale is None else C.c_void_p(scale.data_ptr() + 8)


# ---- gradient-magnitude monitor of the fp16 stacks (calibration of Precision.grad_scale; off on ordinary steps)
class GradMonitor:
    """While active, every gradient tensor an fp16 stack produces gets one extra vq_absmax pass; `report()` syncs once and
    returns per precision object the largest and the smallest non-zero per-tensor maximum IN STORED (scaled) UNITS."""

    def __init__(self):
        self.slots = []       # (precision object, device scalar)

    def watch(self, prec, t):
        if prec.dtype not in HALF_RANGE or t is None or t.numel() % 8:
            return
        out = torch.zeros(1, dtype=torch.float32, device=t.device)
        lib().call("vq_absmax", ptr(t), t.numel(), dtype_code(t), ptr(out), stream_of(t))
        self.slots.append((prec, out))

    def report(self):
        res = {}
        if not self.slots:
            return res
        vals = torch.cat([s for _, s in self.slots]).tolist()
        for (prec, _), v in zip(self.slots, vals):
            r = res.setdefault(id(prec), {"prec": prec, "max": 0.0, "min": float("inf"), "tensors": 0, "zero": 0})
            r["tensors"] += 1
            if v > 0:
                r["max"], r["min"] = max(r["max"], v), min(r["min"], v)
            else:
                r["zero"] += 1
        return res


_monitor = None


@contextlib.contextmanager
def monitor_gradients():
    global _monitor
    mon = GradMonitor()
    _monitor = mon
    try:
        yield mon
    finally:
        _monitor = None


def _watch(prec, t):
    if _monitor is not None:
        _monitor.watch(prec, t)
    return t


def set_default_precision(p) -> None:
    global _default_precision, _fp32_split
    _default_precision = _PRECISIONS[p] if isinstance(p, str) else p
    if _default_precision.dtype == torch.float32:
        _fp32_split = _default_precision.split


def default_precision() -> Precision:
    return _default_precision


def resolve_precision(p) -> Precision:
    if p is None:
        return _default_precision
    return _PRECISIONS[p] if isinstance(p, str) else p


_fp32_split = 3


def set_fp32_split(split: int) -> None:
    """MFMA operand split used for fp32-storage tensors: 3 (default), 6 (fp32-exact products) or 1."""
    global _fp32_split
    assert split in (1, 3, 6)
    _fp32_split = split


def split_for(x: torch.Tensor) -> int:
    """MFMA operand split of the kernels that consume `x`: 1 for 16-bit storage; for fp32 storage what the REGION's precision
    object says (policy "fp32" = single bf16 product, "fp32x3" = the 3-term split) — the process-wide default only serves ops
    called outside any module stack."""
    return precision_of(x).split if x.dtype == torch.float32 else 1


def pad8(c: int) -> int:
    return (c + 7) // 8 * 8


# ----------------------------------------------------------------------------- packed weights
_pack_cache: dict = {}
_generation: dict = {}


def bump_generation(data_ptrs) -> None:
    """Called by the fused optimizer: its kernel rewrites parameters behind autograd's back (no
    `_version` bump), so packed copies keyed on these storages must be invalidated explicitly."""
    for p in data_ptrs:
        _generation[p] = _generation.get(p, 0) + 1


def _packed(weight: torch.Tensor, kind: str, cout_pad: int, cin_pad: int, split: int, desc=None, op: int = VQ_BF16):
    """16-bit GEMM operand of an OIHW fp32 master weight, cached on (storage, version, generation) -> (operand, scale).
    `desc` is the VqConvDesc of the launch that will consume it: the library picks the packed layout (row-major, or
    MFMA-fragment order for the direct-to-register kernels) per descriptor.  op = VQ_BF16: bf16 values, scale None;
    op = VQ_F16: binary16 values of w * s_w and the device record {|w|max, s_w, 1/s_w, 0} the pack kernels fill."""
    L = lib()
    layout = L.dll.vq_conv_weight_layout(C.byref(desc)) if desc is not None else 0
    key = (weight.data_ptr(), weight._version, _generation.get(weight.data_ptr(), 0), kind, cout_pad, cin_pad, split,
           layout, str(weight.device), tuple(weight.shape), op)
    # one entry per (weight, direction, operand type, VARIANT): the discriminator runs at two batch sizes per step (real + fake,
    # then fake alone) and the larger one selects kernels with another operand layout for the 512-channel layers — keyed without
    # the variant, the two evicted each other every step (24 single-weight pack launches / step at config 3)
    ck = (weight.data_ptr(), kind, op) + key[4:8]
    hit = _pack_cache.get(ck)
    if hit is not None and hit[0] == key:
        return hit[1], hit[2]
    co, ci, r, s = weight.shape
    rows, kch = (cout_pad, cin_pad) if kind == "fwd" else (cin_pad, cout_pad)
    # (VQ_F16X2 operands: the reduction runs over virtual channels — hi and lo pieces — twice the real count)
    n = L.size("vq_packed_weight_elems", rows, r, s, kch * (2 if op == VQ_F16X2 else 1), split, layout)
    static = hit is not None and hit[0][3:] == key[3:] and hit[1].numel() == n     # same weight, new values only
    if pack_stats is not None:
        pack_stats[(tuple(weight.shape), kind, op, "new" if hit is None else ("values" if static else "variant"))] += 1
    scaled = op in (VQ_F16, VQ_F16X2)
    buf = hit[1] if static else torch.empty(n, dtype=torch.float16 if scaled else torch.bfloat16, device=weight.device)
    scale = hit[2] if static else (torch.zeros(4, dtype=torch.float32, device=weight.device) if scaled else None)
    w = weight.detach()
    if not w.is_contiguous():
        w = w.contiguous()
    global _pack_launches
    _pack_launches += 1                # (monotonic: run_on_side_stream notices in-place re-packs too)
    L.call("vq_pack_weight_fwd" if kind == "fwd" else "vq_pack_weight_dgrad", ptr(w), co, ci, r, s, cout_pad,
           cin_pad, split, layout, op, ptr(scale), ptr(buf), stream_of(w))
    _pack_cache[ck] = (key, buf, scale)
    if not static:
        global _pack_epoch
        _pack_epoch += 1           # a new (weight, operand) pair: device job tables built before it are stale
        _pack_index.setdefault(ck[:3], [])
        if ck not in _pack_index[ck[:3]]:
            _pack_index[ck[:3]].append(ck)
        # The key is an ADDRESS: a temporary weight that dies hands its address — and this entry — to the next tensor of the
        # same shape the allocator puts there (seen as a flaky parity test).  Module parameters live as long as their module;
        # for anything else the entry dies with the tensor object.
        # (views — the temporal taps of a Conv3d weight — share their base's memory and are left alone)
        if not isinstance(weight, torch.nn.Parameter) and weight._base is None:
            weakref.finalize(weight, _evict_packed, weight.data_ptr())
    return buf, scale


def _evict_packed(data_ptr: int) -> None:
    global _pack_epoch
    hit = False
    for k3 in [k for k in _pack_index if k[0] == data_ptr]:
        for ck in _pack_index.pop(k3):
            _pack_cache.pop(ck, None)
            hit = True
    if hit:
        _pack_epoch += 1


def packed_scale(weight: torch.Tensor, kind: str, op: int):
    """The device scale record {|w|max, s_w, 1/s_w, 0} of a packed binary16 operand of `weight` (any variant: same values)."""
    cks = _pack_index.get((weight.data_ptr(), kind, op))
    if not cks:
        raise KeyError("no packed operand for this weight yet")
    cur = (weight._version, _generation.get(weight.data_ptr(), 0))
    for ck in reversed(cks):               # a variant packed from the weight's current values
        if _pack_cache[ck][0][1:3] == cur:
            return _pack_cache[ck][2]
    raise KeyError("the packed operands of this weight are stale")


_pack_launches = 0         # single-weight pack launches issued so far
_pack_index: dict = {}     # (ptr, kind, op) -> [cache keys of its variants]
_pack_epoch = 0
pack_stats = None          # debugging: set to collections.Counter() to count single-weight pack launches by (shape, kind, op, reason)


class PackPlan:
    """All cached packed operands of one optimizer's weights as a device job table: `run()` re-packs them in ONE
    launch right after the optimizer step (vq_pack_weights_multi) and re-validates the cache entries, instead of one
    pack launch per conv and direction on first use (176 launches / step at config 3).  Rebuilt lazily whenever a
    new operand joined the cache (first steps, a new input shape choosing another kernel / layout)."""

    def __init__(self, params):
        self.params = [p for p in params if p.dim() == 4]
        self.epoch = -1
        self.entries, self.table, self.blocks, self.with_scales = [], None, 0, 0

    def _build(self):
        L = lib()
        self.entries, jobs, blocks, self.with_scales = [], [], 0, 0
        for p in self.params:
            for kind in ("fwd", "dgrad"):
                for op in (VQ_BF16, VQ_F16, VQ_F16X2):
                    if not p.is_contiguous():
                        continue
                    for ck in _pack_index.get((p.data_ptr(), kind, op), ()):
                        key, buf, scale = _pack_cache[ck]
                        _, _, _, _, cout_pad, cin_pad, split, layout, _, shape, _ = key
                        co, ci, r, s = shape
                        job = _lib.VqPackJob()
                        L.call("vq_pack_job", C.byref(job), ptr(p), co, ci, r, s, cout_pad, cin_pad, split, layout,
                               1 if kind == "dgrad" else 0, op, ptr(scale), ptr(buf))
                        job.block_start = blocks
                        blocks += L.size("vq_pack_job_blocks", C.byref(job))
                        jobs.append(job)
                        self.entries.append((p, ck))
                        self.with_scales |= int(op in (VQ_F16, VQ_F16X2))
        self.blocks = blocks
        if jobs:
            raw = b"".join(bytes(memoryview(j)) for j in jobs)
            self.table = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(self.params[0].device)
        self.epoch = _pack_epoch

    def prepare(self):
        """Re-build the job table if operands joined or left the cache since the last build (clear_pack_cache after a checkpoint
        load, an evicted temporary, a new input shape)."""
        if self.params and self.epoch != _pack_epoch:
            self._build()

    def algorithmic_bytes(self) -> float:
        """One fp32 read per weight and one 2-byte write per packed copy of the CURRENT table (call prepare() first)."""
        tot = 0.0
        for p, ck in self.entries:
            hit = _pack_cache.get(ck)
            if hit is not None:
                tot += 4.0 * p.numel() * (2 if ck[2] in (VQ_F16, VQ_F16X2) else 1) + hit[1].numel() * 2.0
        return tot      # (binary16 operands read the master weight twice: |w|max, then the scaled conversion)

    def run(self):
        if not self.params:
            return
        self.prepare()
        if not self.entries:
            return
        lib().call("vq_pack_weights_multi", ptr(self.table), len(self.entries), self.blocks, self.with_scales,
                   stream_of(self.params[0]))
        for p, ck in self.entries:                 # the operands now hold the current values: refresh the cache keys
            key, buf, scale = _pack_cache[ck]
            _pack_cache[ck] = ((key[0], p._version, _generation.get(p.data_ptr(), 0)) + key[3:], buf, scale)


def clear_pack_cache() -> None:
    """Drop the packed bf16 weight copies (after parameters were rewritten behind the cache's back)."""
    global _pack_epoch
    _pack_cache.clear()
    _pack_index.clear()
    _derived_cache.clear()
    _pack_epoch += 1


def clear_caches() -> None:
    clear_pack_cache()
    _tap_cache.clear()
    _grad_sinks.clear()


# ----------------------------------------------------------------------------- layout boundary
class _ToNHWC(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, prec: Precision, shift, scale):
        n, c, h, w = x.shape
        x = x.contiguous().float()
        cp = pad8(c)
        y = torch.empty((n, h, w, cp), dtype=prec.dtype, device=x.device)
        _launch("hbm:layout", _nbytes(x, y), lambda: lib().call("vq_nchw_to_nhwc", ptr(x), ptr(y), n, c, h, w, cp, dtype_code(y),
                                                              ptr(shift), ptr(scale), 1.0, None, stream_of(x)))
        ctx.c = c
        ctx.scale = scale
        ctx.prec = prec
        return y

    @staticmethod
    def backward(ctx, dy):
        n, h, w, cp = dy.shape
        dy = dy.contiguous()
        dx = torch.empty((n, ctx.c, h, w), dtype=torch.float32, device=dy.device)
        # the gradient leaves the stack: its loss scale comes off here (exact: a power of two)
        _launch("hbm:layout", _nbytes(dy, dx), lambda: lib().call("vq_nhwc_to_nchw", ptr(dy), ptr(dx), n, ctx.c, h, w, cp,
                                                                dtype_code(dy), ptr(ctx.scale), 1.0 / ctx.prec.gs(), stream_of(dy)))
        return dx, None, None, None


class _ToNCHW(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, c: int):
        n, h, w, cp = x.shape
        x = x.contiguous()
        y = torch.empty((n, c, h, w), dtype=torch.float32, device=x.device)
        _launch("hbm:layout", _nbytes(x, y), lambda: lib().call("vq_nhwc_to_nchw", ptr(x), ptr(y), n, c, h, w, cp, dtype_code(x),
                                                              None, 1.0, stream_of(x)))
        ctx.cp = cp
        ctx.dt = x.dtype
        ctx.prec = precision_of(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        n, c, h, w = dy.shape
        dy = dy.contiguous().float()
        dx = torch.empty((n, h, w, ctx.cp), dtype=ctx.dt, device=dy.device)
        # the gradient enters the stack: times its loss scale (fp16 stacks; 1 otherwise)
        with region(ctx.prec, backward=True):
            ev = _events()
        _launch("hbm:layout", _nbytes(dy, dx), lambda: lib().call("vq_nchw_to_nhwc", ptr(dy), ptr(dx), n, c, h, w, ctx.cp,
                                                                dtype_code(dx), None, None, ctx.prec.gs(), ev, stream_of(dy)))
        return _watch(ctx.prec, dx), None


def to_nhwc(x: torch.Tensor, prec: Precision | None = None, shift=None, scale=None) -> torch.Tensor:
    """[N,C,H,W] fp32 -> [N,H,W,pad8(C)] in the precision's storage dtype (optionally ScalingLayer)."""
    return _ToNHWC.apply(x, resolve_precision(prec), shift, scale)


def to_nchw(x: torch.Tensor, c: int) -> torch.Tensor:
    return _ToNCHW.apply(x, c)


# ----------------------------------------------------------------------------- AttnBlock self-attention
class _Attention(torch.autograd.Function):
    """F.scaled_dot_product_attention over the tokens of a channels-last qkv tensor [N, ..., 3C] (the output of the
    1x1 qkv conv); heads of `head_dim` channels: 64 over H*W tokens in ae.py:74-90, C/8 over T*H*W tokens in
    tae.py:24-53.  Returns [N, ..., C]."""

    @staticmethod
    def forward(ctx, qkv, head_dim):
        n, c3 = qkv.shape[0], qkv.shape[-1]
        c, t = c3 // 3, qkv[0].numel() // c3
        qkv = qkv.contiguous()
        out = torch.empty(qkv.shape[:-1] + (c,), dtype=qkv.dtype, device=qkv.device)
        lse = torch.empty((n * (c // head_dim), t), dtype=torch.float32, device=qkv.device)
        lib().call("vq_attention_fwd", ptr(qkv), ptr(out), ptr(lse), n, t, c, head_dim, dtype_code(qkv), stream_of(qkv))
        ctx.save_for_backward(qkv, out, lse)
        ctx.head_dim = head_dim
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, out, lse = ctx.saved_tensors
        n, c3 = qkv.shape[0], qkv.shape[-1]
        c, t = c3 // 3, qkv[0].numel() // c3
        L = lib()
        dout = dout.contiguous()
        dqkv = torch.empty_like(qkv)
        ws = workspace(qkv.device, L.size("vq_attention_workspace", n, t, c, ctx.head_dim))
        L.call("vq_attention_bwd", ptr(qkv), ptr(out), ptr(dout), ptr(lse), ptr(dqkv), n, t, c, ctx.head_dim,
               dtype_code(qkv), ptr(ws), ws.numel(), stream_of(dout))
        return dqkv, None


def attention(qkv: torch.Tensor, head_dim: int = 64) -> torch.Tensor:
    return _Attention.apply(qkv, head_dim)


# ----------------------------------------------------------------------------- input preparation
def wavelet_to_nhwc(x: torch.Tensor, precision=None) -> torch.Tensor:
    """utils.py:229-247 on an NCHW fp32 image, written as the NHWC (padded C) activation encoder.conv_in reads.
    The images carry no gradient in the trainer (vae_trainer.py:529-538), so this is forward-only."""
    if x.requires_grad:
        raise NotImplementedError("the wavelet front-end is forward-only (its input is the image batch)")
    prec = resolve_precision(precision)
    n, c, h, w = x.shape
    x = x.contiguous().float()
    cp = pad8(4 * c)
    y = torch.empty((n, h // 2, w // 2, cp), dtype=prec.dtype, device=x.device)
    lib().call("vq_wavelet_fwd", ptr(x), ptr(y), n, c, h, w, cp, dtype_code(y), 1, stream_of(x))
    return y


def wavelet_nchw(x: torch.Tensor) -> torch.Tensor:
    """utils.wavelet_transform_multi_channel with the reference's own layout: NCHW fp32 [B,4C,H/2,W/2]."""
    n, c, h, w = x.shape
    x = x.detach().contiguous().float()
    y = torch.empty((n, 4 * c, h // 2, w // 2), dtype=torch.float32, device=x.device)
    lib().call("vq_wavelet_fwd", ptr(x), ptr(y), n, c, h, w, 4 * c, 1, 0, stream_of(x))
    return y


class _Flip(torch.autograd.Function):
    """torch.flip along H and/or W of an NCHW fp32 tensor, channels [neg0, neg1) negated; its own backward."""

    @staticmethod
    def forward(ctx, x, flip_h, flip_w, neg0, neg1):
        ctx.args = (flip_h, flip_w, neg0, neg1)
        return _flip_raw(x, flip_h, flip_w, neg0, neg1)

    @staticmethod
    def backward(ctx, g):
        return _flip_raw(g, *ctx.args), None, None, None, None


def _flip_raw(x, flip_h, flip_w, neg0, neg1):
    n, c, h, w = x.shape
    x = x.contiguous().float()
    y = torch.empty_like(x)
    lib().call("vq_flip_nchw", ptr(x), ptr(y), n, c, h, w, int(flip_h), int(flip_w), neg0, neg1, stream_of(x))
    return y


def flip_nchw(x: torch.Tensor, flip_h=False, flip_w=False, negate_channels=(0, 0)) -> torch.Tensor:
    """vae_trainer.py:534-536,567-575,664-671: `torch.flip(x, [-1])` = flip_w, `[-2]` = flip_h; the latent sign flips
    `z[:, a:b] = -z[:, a:b]` are passed as negate_channels=(a, b) (non-negative indices)."""
    return _Flip.apply(x, bool(flip_h), bool(flip_w), int(negate_channels[0]), int(negate_channels[1]))


def area_downsample(x: torch.Tensor, size) -> torch.Tensor:
    """F.interpolate(x, size=size, mode="area") for integer ratios (vae_trainer.py:531-533); images carry no gradient."""
    n, c, h, w = x.shape
    ho, wo = size
    if (h, w) == (ho, wo):
        return x
    if h % ho or w % wo or h // ho != w // wo:
        raise NotImplementedError(f"area resize {h}x{w} -> {ho}x{wo}: only equal integer ratios are on the HIP path")
    x = x.detach().contiguous().float()
    y = torch.empty((n, c, ho, wo), dtype=torch.float32, device=x.device)
    lib().call("vq_area_downsample_nchw", ptr(x), ptr(y), n, c, h, w, h // ho, stream_of(x))
    return y


# ----------------------------------------------------------------------------- convolution
_launch_hook = None


def set_launch_hook(hook) -> None:
    """bench.py installs hook(kind, algorithmic_flops, launch_fn, tag) to bracket the conv launches with HIP
    events on the launch stream (tag = layer shape, for the per-shape table); None (default) = plain launch."""
    global _launch_hook
    _launch_hook = hook


def _launch(kind, flops, fn, tag=""):
    """kind "conv_igemm" / "conv_wgrad": `flops` = algorithmic FLOPs of the launch; kind "hbm:<family>": `flops` carries the
    ALGORITHMIC BYTES of the call (the minimum traffic of the op as the reference states it)."""
    if _launch_hook is None:
        fn()
    else:
        _launch_hook(kind, flops, fn, tag)


def _nbytes(*tensors) -> float:
    return float(sum(t.numel() * t.element_size() for t in tensors if t is not None))


def _tag(what, n, h, w, cin, cout, r, stride, up):
    return f"{what} {cin}->{cout} in {n}x{h}x{w} k{r} s{stride} up{up}"


# ---- weight gradients on a side stream ----------------------------------------------------------------------------------------
# Nothing in a backward pass waits for a weight gradient: the chain is data gradient -> GroupNorm backward -> data gradient ..., and
# dW is only read by the optimizer (or the gradient exchange).  The weight-gradient GEMMs (MFMA-bound, one block per CU) therefore go
# to a second HIP stream, where they run UNDER the HBM-bound kernels of the chain (GroupNorm backward: two passes over x and dy, max-
# pool / LPIPS-tap backward, layout) instead of between them.  Only launches that write into gradient SINKS (the optimizer's flat
# buffers) move: a freshly allocated gradient tensor would be handed to autograd on the main stream.  Ordering: the side stream
# waits for the main stream at every launch (its inputs were produced there); the main stream waits for the side stream once, at
# the end of the backward pass (autograd engine callback) — and before a gradient bucket goes on the wire (distributed._launch).
_wgrad_overlap = os.environ.get("VQ_WGRAD_OVERLAP", "1") != "0"
_side_streams: dict = {}
_side_dirty: dict = {}          # device index -> the side stream has work the main stream has not waited for


def set_wgrad_overlap(on: bool) -> None:
    global _wgrad_overlap
    join_side_stream()
    _wgrad_overlap = bool(on)


# Inputs of side-stream launches, held until the side stream is done with them: [(event recorded behind the launch, tensors)].
# Entries whose event has completed are dropped at the next launch (held until the single join at the end of the backward
# pass, the dy of every conv layer stayed resident for the whole pass); everything goes at a join.
_side_keep: list = []
side_keep_peak = 0               # (diagnostics: the largest number of tensors held at once since import)


def _prune_side_keep() -> None:
    global side_keep_peak
    side_keep_peak = max(side_keep_peak, sum(len(t) for _, t in _side_keep))
    while _side_prune and _side_keep and _side_keep[0][0].query():       # (launch order = completion order on one stream)
        _side_keep.pop(0)


_side_prune = os.environ.get("VQ_SIDE_KEEP_PRUNE", "1") != "0"      # A/B knob: 0 = hold every input until the join (rounds 1-4)


def join_side_stream(device=None) -> None:
    """The current stream waits for everything the weight-gradient stream holds (no host sync)."""
    for idx in list(_side_dirty):
        if _side_dirty.get(idx) and (device is None or torch.device(device).index in (None, idx)):
            torch.cuda.current_stream(idx).wait_stream(_side_streams[idx])
            _side_dirty[idx] = False
    if not any(_side_dirty.values()):
        # Everything the side stream read is now ordered before whatever this stream enqueues next: the inputs can go back to the
        # allocator the ordinary way.  (They are kept alive by reference rather than `record_stream`ed: with record_stream the
        # allocator defers each block's reuse until an event of the OTHER stream has passed, so which blocks are free at a given
        # allocation depends on how far the side stream happens to be — the pool kept growing by fresh hipMallocs for many steps
        # after every reset, 2-3x slower steps in the short secondary legs of bench.py: profiles/r4e_*.)
        _side_keep.clear()


@contextlib.contextmanager
def collective_after_side_stream(device):
    """Issue a gradient collective so that it starts after BOTH streams' work so far, without stalling the backward chain: the
    side stream waits for the chain's stream (it is the one that lags: no stall in practice), and the collective is issued with the
    side stream current — the communicator's stream waits for that — so the chain's stream never waits for the weight gradients
    here (a `join_side_stream()` per bucket would make it wait ~one weight-gradient GEMM ten times per step)."""
    idx = torch.device(device).index
    if idx is None:
        idx = torch.cuda.current_device()
    side = _side_streams.get(idx)
    if side is None or not _wgrad_overlap:
        yield
        return
    side.wait_stream(torch.cuda.current_stream(idx))
    with torch.cuda.stream(side):
        yield


# A/B knob (round 5): the backward chain's MFMA-bound launches (forward / data-gradient convs) wait for the weight-gradient stream, so a
# weight-gradient GEMM only ever shares the chip with the HBM-bound kernels that follow the data gradient it was issued behind
# (GroupNorm backward, pools, LPIPS taps) and never with another GEMM.  1 = on.
_side_mfma_barrier = os.environ.get("VQ_SIDE_MFMA_BARRIER", "0") == "1"


def _mfma_waits_for_side(t) -> None:
    if _side_mfma_barrier and t.is_cuda and _side_dirty.get(t.device.index):
        side = _side_streams.get(t.device.index)
        if side is not None and torch.cuda.current_stream(t.device.index) != side:      # (not from inside a side-stream context)
            join_side_stream(t.device.index)


_join_queued = -1                # id of the backward pass (autograd graph task) whose end-of-pass join is already queued.  Module-wide, not
                                 # thread-local: backward nodes run on the engine's device thread, the callback on the caller's.  Graph-task
                                 # ids are unique per backward call, so a pass that raised (its callback never ran) leaves nothing stale, and
                                 # joins in the MIDDLE of a pass (run_on_side_stream after a re-pack, the reducer, VQ_SIDE_MFMA_BARRIER) do
                                 # not make every later launch of that pass queue another callback (round-5 advice)


_defer_join = False


def _end_of_pass_join():
    if not _defer_join:
        join_side_stream()


@contextlib.contextmanager
def deferred_side_join():
    """A backward pass run inside this context does NOT make the caller's stream wait for the weight-gradient stream at its end: the
    caller promises to call join_side_stream() before anything reads those gradients (VAETrainStep: the discriminator's backward —
    its last weight gradients then run under the LPIPS forward that follows instead of in front of it)."""
    global _defer_join
    prev, _defer_join = _defer_join, True
    try:
        yield
    finally:
        _defer_join = prev


def _queue_join() -> bool:
    """Inside a backward pass: have the engine call join_side_stream() when the pass is over (once per pass).  -> queued?"""
    global _join_queued
    tid = torch._C._current_graph_task_id()
    if tid < 0:                     # not inside a backward pass (a direct call from a test / tool): the caller joins right away
        return False
    if _join_queued == tid:
        return True
    try:
        torch.autograd.Variable._execution_engine.queue_callback(_end_of_pass_join)
        _join_queued = tid
    except RuntimeError:
        return False
    return True


# A/B knob (round 5): VQ_SIDE_CU_MASK=<n> creates the weight-gradient stream with a CU mask of n compute units
# (hipExtStreamCreateWithCUMask; the driver deals mask bits round-robin over the 8 XCDs, so the first n bits are n / 8 CUs of each), the
# idea being that the chain's HBM-bound kernels then keep CUs the GEMMs cannot occupy.  0 / unset = an ordinary stream.
_side_cu_mask = int(os.environ.get("VQ_SIDE_CU_MASK", "0") or 0)


def _new_side_stream(idx: int):
    if _side_cu_mask <= 0:
        return torch.cuda.Stream(device=idx)
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    words = (_side_cu_mask + 31) // 32
    mask = (ctypes.c_uint32 * words)(*[(0xFFFFFFFF if (w + 1) * 32 <= _side_cu_mask else (1 << (_side_cu_mask - w * 32)) - 1) for w in range(words)])
    handle = ctypes.c_void_p()
    with torch.cuda.device(idx):
        rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(handle), ctypes.c_uint32(words), mask)
    if rc != 0 or not handle.value:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask({_side_cu_mask} CUs) failed: {rc}")
    return torch.cuda.ExternalStream(handle.value, device=idx)


@contextlib.contextmanager
def _on_side_stream(use: bool, *inputs, join: bool = True):
    if not (use and _wgrad_overlap and inputs[0].is_cuda):
        yield False
        return
    idx = inputs[0].device.index
    side = _side_streams.get(idx)
    if side is None:
        side = _side_streams[idx] = _new_side_stream(idx)             # (stream priority -1 / 0 / +1 measured: no difference, profiles/r4i_*)
    side.wait_stream(torch.cuda.current_stream(idx))
    _lib._ws_slot.v = 1                 # scratch buffers of their own (the main stream's launches keep using slot 0 meanwhile)
    ev = None
    try:
        with torch.cuda.stream(side):
            yield True
            ev = torch.cuda.Event()
            ev.record()
    finally:
        _lib._ws_slot.v = 0
    if ev is not None:           # alive until the side stream has passed this launch (or a join): no reuse while it reads
        _side_keep.append((ev, [t for t in inputs if t is not None]))
        _prune_side_keep()
    _side_dirty[idx] = True
    if join and not _queue_join():      # (join=False: the caller waits for an event of its own where it consumes the results)
        join_side_stream(idx)


class SideResult:
    """Tensors produced by `run_on_side_stream` + the event that marks them complete: `wait()` before the first use on the caller's
    stream."""

    def __init__(self, value, event, stream):
        self.value, self.event, self.stream = value, event, stream

    def wait(self):
        if self.event is not None:
            torch.cuda.current_stream().wait_event(self.event)
            main = torch.cuda.current_stream()
            for t in (self.value if isinstance(self.value, (list, tuple)) else [self.value]):
                if isinstance(t, torch.Tensor):
                    t.record_stream(main)          # allocated in the side stream's pool, consumed here
            self.event = None
        return self.value


def run_on_side_stream(fn, *inputs) -> SideResult:
    """Forward-only work that nothing on the main stream waits for yet (LPIPS' features of the TARGET image: they depend on the input
    batch alone) under whatever the main stream runs meanwhile.  -> SideResult (call .wait() where the values are consumed)."""
    if not (_wgrad_overlap and inputs and inputs[0].is_cuda):
        return SideResult(fn(), None, None)
    n_packed = _pack_launches
    with _on_side_stream(True, *inputs, join=False):
        value = fn()
        ev = torch.cuda.Event()
        ev.record()
    res = SideResult(value, ev, None)
    if _pack_launches != n_packed:      # a weight was (re-)packed over there — new entry or in place: nobody else may read it before that
        join_side_stream(inputs[0].device)
    return res


# VqConvDesc.kernel_hint (include/vqhip.h): 0 in the product.  tests/ and tools/ force shipped kernels at shapes the library's own
# choice would route elsewhere (conv: forward / data-gradient launches; wgrad: weight-gradient launches).
_hint_conv = 0
_hint_wgrad = 0


@contextlib.contextmanager
def kernel_hints(conv: int = 0, wgrad: int = 0):
    """Test / A-B helper: descriptors built inside carry these hints.  Packed weights depend on the kernel (layout): the cache is
    keyed on the layout the hinted descriptor asks for, so no clearing is needed."""
    global _hint_conv, _hint_wgrad
    prev = (_hint_conv, _hint_wgrad)
    _hint_conv, _hint_wgrad = int(conv), int(wgrad)
    try:
        yield
    finally:
        _hint_conv, _hint_wgrad = prev


def _desc(n, h, w, cin, ho, wo, cout, cin_w, cout_w, r, s, stride, dil_in, up, pad_t, pad_l, dtype, split, relu, subpix=0,
          alpha=1.0, wgrad=False):
    d = VqConvDesc()
    (d.N, d.H, d.W, d.Cin, d.Ho, d.Wo, d.Cout, d.Cin_w, d.Cout_w, d.R, d.S, d.stride, d.dil_in, d.up, d.pad_t,
     d.pad_l, d.dtype, d.split, d.relu, d.subpix) = (n, h, w, cin, ho, wo, cout, cin_w, cout_w, r, s, stride, dil_in, up,
                                                     pad_t, pad_l, dtype, split, int(relu), subpix)
    d.alpha, d.kernel_hint, d.alpha_dev = float(alpha), (_hint_wgrad if wgrad else _hint_conv), None
    d.range_events = None if wgrad else _events()       # (weight gradients are fp32: nothing to clip)
    return d


# Phase-decomposed ("sub-pixel") forms of the two resampling convolutions (include/vqhip.h, VqConvDesc.subpix):
#   * Upsample = nearest-2x + 3x3 conv (ae.py:164-166): output pixel (2i+a, 2j+b) only ever sees a 2x2 window of the
#     LOW-resolution input, with the 3x3 taps that land on the same input pixel summed beforehand — four 2x2 convs
#     (16 multiply-accumulates per input pixel and channel pair) instead of one 3x3 conv at the high resolution (36).
#     Its data gradient is a plain 4x4 / stride-2 / pad-1 conv over dy with the same tap sums (again 16 instead of 36,
#     and no 2x2 sum-pool pass over a high-resolution intermediate).  Its weight gradient is that 4x4 / stride-2 conv's
#     weight gradient with the roles swapped (x := dy, dy := the conv's input), folded from 16 onto the 9 taps
#     (vq_subpixel_wgrad_fold); the bias gradient becomes a column sum of dy.  VQ_SUBPIXEL_WGRAD=0 keeps the `up=2` gather.
#   * Downsample = 3x3 / stride-2 conv (ae.py:150-154): its data gradient per output parity is a 2x2 conv over dy
#     (16 executed, 9 useful) instead of nine taps over the zero-dilated dy (36 executed, 9 useful).
# VQ_SUBPIXEL=0 restores the single-conv forms (A/B runs).
_subpixel = os.environ.get("VQ_SUBPIXEL", "1") != "0"
_subpixel_wgrad = os.environ.get("VQ_SUBPIXEL_WGRAD", "1") != "0"      # the Upsample weight gradient too (A/B knob)
_derived_cache: dict = {}


def set_subpixel(on: bool) -> None:
    global _subpixel
    _subpixel = bool(on)


def _derived_weight(weight: torch.Tensor, mode: int) -> torch.Tensor:
    """fp32 tap sums of a 3x3 OIHW weight (vq_subpixel_weights; mode 0 Upsample fwd, 1 Upsample dgrad, 2 Downsample
    dgrad), cached on (storage, version, optimizer generation).  The buffer is re-filled in place, so the packed-operand
    cache sees one stable weight whose generation is bumped on every refresh."""
    o, i = weight.shape[:2]
    key = (weight.data_ptr(), weight._version, _generation.get(weight.data_ptr(), 0), str(weight.device), tuple(weight.shape))
    hit = _derived_cache.get((weight.data_ptr(), mode))
    if hit is not None and hit[0] == key:
        return hit[1]
    shape = ((4 * o, i, 2, 2), (i, o, 4, 4), (4 * i, o, 2, 2))[mode]
    reuse = hit is not None and tuple(hit[1].shape) == shape and hit[1].device == weight.device
    buf = hit[1] if reuse else torch.empty(shape, dtype=torch.float32, device=weight.device)
    w = weight.detach()
    if not w.is_contiguous():
        w = w.contiguous()
    lib().call("vq_subpixel_weights", ptr(w), ptr(buf), o, i, mode, stream_of(w))
    bump_generation([buf.data_ptr()])
    _derived_cache[(weight.data_ptr(), mode)] = (key, buf)
    return buf


def _subpixel_up_shape(wshape, stride, pad_t, pad_l, up) -> bool:
    co, ci, r, s = wshape
    return (_subpixel and up == 2 and r == 3 and s == 3 and stride == 1 and pad_t == 1 and pad_l == 1 and co % 32 == 0
            and ci % 8 == 0)


def _subpixel_up(weight, stride, pad_t, pad_l, up) -> bool:
    return _subpixel_up_shape(weight.shape, stride, pad_t, pad_l, up) and weight.dtype == torch.float32


def _subpixel_wgrad_into(x, dy, co_w, ci_w, dw, acc, split, gs=1.0):
    """Weight gradient of an Upsample conv (x [N,H,W,Cin] low resolution, dy [N,2H,2W,Cout]) through its transposed form:
    the 4x4 / stride-2 / pad-1 wgrad with the roles swapped, then the 16 -> 9 tap fold into `dw` (`acc` = 1 adds).
    gs: loss scale carried by dy (fp16 stacks), removed from the result."""
    n, h, w, cin = x.shape
    _, ho, wo, cout = dy.shape
    L = lib()
    st = stream_of(dy)
    dw4 = torch.empty((ci_w, co_w, 4, 4), dtype=torch.float32, device=dy.device)
    d4 = _desc(n, ho, wo, cout, h, w, cin, co_w, ci_w, 4, 4, 2, 1, 1, 1, 1, dtype_code(dy), split, False, alpha=1.0 / gs, wgrad=True)
    ws = workspace(dy.device, L.size("vq_conv2d_wgrad_workspace", C.byref(d4)))
    flops = 2.0 * n * h * w * co_w * ci_w * 16
    _launch("conv_wgrad", flops, lambda: L.call("vq_conv2d_wgrad", C.byref(d4), ptr(dy), ptr(x), ptr(dw4), None, 0, ptr(ws),
                                                ws.numel(), st),
            _tag("wgrad", n, h, w, ci_w, co_w, 3, 1, "2sub") if _launch_hook else "")
    L.call("vq_subpixel_wgrad_fold", ptr(dw4), ptr(dw), co_w, ci_w, acc, st)


def subpixel_up_eligible(weight) -> bool:
    """True when an `up=2` 3x3 / stride-1 / pad-1 conv with this weight runs in the phase-decomposed form."""
    return _subpixel_up(weight, 1, 1, 1, 2)


def _subpixel_down(weight, stride, pad_t, pad_l, up, h, w, ho, wo) -> bool:
    co, ci, r, s = weight.shape
    return (_subpixel and up == 1 and r == 3 and s == 3 and stride == 2 and pad_t == 0 and pad_l == 0 and ci % 32 == 0
            and co % 8 == 0 and h == 2 * ho and w == 2 * wo and weight.dtype == torch.float32)


# Gradient sinks: the fused optimizer re-homes every parameter's gradient into a flat fp32 buffer
# (optim.FlatGroup).  When a sink is registered for a parameter storage, the backward kernels accumulate
# straight into it (the buffer is zeroed by zero_grad) and hand `None` to autograd — no per-parameter `add`
# launches — then fire the optional "gradient ready" callback (the DP reducer's bucket countdown).
_grad_sinks: dict = {}


def register_grad_sink(param: torch.Tensor, view: torch.Tensor) -> None:
    """(Re-)register the flat-buffer view the backward kernels accumulate into; a "gradient ready" callback that is already
    attached to this parameter (the DP reducer's) survives a re-registration (FlatGroup.rebind_grads)."""
    prev = _grad_sinks.get(param.data_ptr())
    _grad_sinks[param.data_ptr()] = [view, prev[1] if prev is not None else None]


def set_grad_ready_callback(param: torch.Tensor, cb) -> bool:
    ent = _grad_sinks.get(param.data_ptr())
    if ent is None:
        return False
    ent[1] = cb
    return True


def clear_grad_sinks() -> None:
    _grad_sinks.clear()


def unregister_grad_sinks(data_ptrs) -> None:
    for p in data_ptrs:
        _grad_sinks.pop(p, None)


def _sink_of(param):
    return None if param is None else _grad_sinks.get(param.data_ptr())


def _conv_out_hw(h, w, r, s, stride, pad_t, pad_l, up, out_hw):
    if out_hw is not None:
        return out_hw
    # PyTorch conv arithmetic with symmetric padding; asymmetric bottom/right padding (Downsample,
    # ae.py:150-154) is requested through out_hw.
    return (h * up + 2 * pad_t - r) // stride + 1, (w * up + 2 * pad_l - s) // stride + 1


def conv_fwd_raw(x, weight, bias, residual, stride, pad_t, pad_l, up, relu, split, out_hw, out=None, gn=None):
    """`out` (optional): the contiguous [N,Ho,Wo,Cout] tensor to write; it may be `residual` itself (in-place accumulate:
    every output element is read and written by the same lane).
    gn = (groups, eps): the output feeds an FP32GroupNorm — where the kernel can, its epilogue also reduces the GroupNorm
    statistics of y (vq_conv2d_fwd gn_partials) and the result rides on the tensor (`y._vq_gn`), so gn_fwd_raw skips its own
    statistics pass over it."""
    n, h, w, cin = x.shape
    co_w, ci_w, r, s = weight.shape
    assert pad8(ci_w) == cin, f"input has {cin} channels, weight expects pad8({ci_w})"
    cout = pad8(co_w)
    ho, wo = _conv_out_hw(h, w, r, s, stride, pad_t, pad_l, up, out_hw)
    x = x.contiguous()
    _mfma_waits_for_side(x)
    y = torch.empty((n, ho, wo, cout), dtype=x.dtype, device=x.device) if out is None else out
    assert y.is_contiguous() and tuple(y.shape) == (n, ho, wo, cout) and y.dtype == x.dtype
    res = residual.contiguous() if residual is not None else None
    if (ho, wo) == (2 * h, 2 * w) and _subpixel_up(weight, stride, pad_t, pad_l, up):
        # four 2x2 convs of the low-resolution input, one launch: rows = (phase, cout), depth-to-space store
        wd = _derived_weight(weight, 0)
        d = _desc(n, h, w, cin, h, w, 4 * cout, ci_w, 4 * co_w, 2, 2, 1, 1, 1, 1, 1, dtype_code(x), split, relu, subpix=2)
        wp, sc = _packed(wd, "fwd", 4 * cout, cin, split, d, _op(x))
        d.alpha_dev = _adev(sc)
        flops = 2.0 * n * h * w * 4 * co_w * ci_w * 4
        _launch("conv_igemm", flops, lambda: lib().call("vq_conv2d_fwd", C.byref(d), ptr(x), ptr(wp), ptr(bias), ptr(res),
                                                        None, ptr(y), None, 0, stream_of(x)),
                _tag("fwd", n, h, w, ci_w, co_w, r, stride, "2sub") if _launch_hook else "")
        return y
    d = _desc(n, h, w, cin, ho, wo, cout, ci_w, co_w, r, s, stride, 1, up, pad_t, pad_l, dtype_code(x), split, relu)
    wp, sc = _packed(weight, "fwd", cout, cin, split, d, _op(x))
    d.alpha_dev = _adev(sc)
    flops = 2.0 * n * ho * wo * co_w * ci_w * r * s
    part, bp = None, 0
    if gn is not None and _gn_fusion and out is None:
        bp = lib().dll.vq_conv2d_gn_tile(C.byref(d), int(gn[0]))
        if bp > 0:
            part = torch.empty((n, (ho * wo) // bp, int(gn[0]), 2), dtype=torch.float32, device=x.device)
    _launch("conv_igemm", flops, lambda: lib().call("vq_conv2d_fwd", C.byref(d), ptr(x), ptr(wp), ptr(bias), ptr(res),
                                                    None, ptr(y), ptr(part), int(gn[0]) if part is not None else 0, stream_of(x)),
            _tag("fwd", n, h, w, ci_w, co_w, r, stride, up) if _launch_hook else "")
    if part is not None:
        stats = torch.empty((2, n * int(gn[0])), dtype=torch.float32, device=x.device)
        lib().call("vq_gn_stats_finalize", ptr(part), n, (ho * wo) // bp, ho * wo, cout, int(gn[0]), float(gn[1]), ptr(stats[0]),
                   ptr(stats[1]), stream_of(x))
        y._vq_gn = (stats, int(gn[0]), float(gn[1]))
    return y


# GroupNorm statistics from the producing convolution's epilogue (VQ_GN_FUSED=0 / set_gn_fusion(False): always the separate pass)
_gn_fusion = os.environ.get("VQ_GN_FUSED", "1") != "0"


def set_gn_fusion(on: bool) -> None:
    global _gn_fusion
    _gn_fusion = bool(on)


# GroupNorm backward with its sums formed in the epilogue of the data-gradient conv that produces dy (VqGnBwdFuse): four tensor passes
# instead of five.  OFF by default: per layer the fused sums cost less than the reduction pass they replace (profiles/r3ab_*: 128 ch
# @256^2 +58 us on the conv against a 108 us pass), in the STEP they lose 0.9 % (247.3 vs 249.5 img/s, profiles/r3ac_bench_ab.txt:
# the implicit-GEMM family slows down by more than the GroupNorm family gains) — and the mere presence of the path in the shared conv
# epilogue cost the default step 1.5 % (profiles/r3ad_bench_ab.txt), so a release library does not carry it: vq_conv2d_gnb_rows
# answers 0 there and this switch does nothing; `make ABLATE=1` libraries and the emulator do.  VQ_GN_BWD_FUSED=1 /
# set_gn_bwd_fusion(True) turns it on (A/B runs, tests).
_gn_bwd_fused = os.environ.get("VQ_GN_BWD_FUSED", "0") == "1"


def set_gn_bwd_fusion(on: bool) -> None:
    global _gn_bwd_fused
    _gn_bwd_fused = bool(on)


def conv_dgrad_raw(dy, x, weight, stride, pad_t, pad_l, up, split, mask_input_grad, add=None, out=None, keep_up=False,
                   alpha=None, gn_bwd=None):
    """dx of the conv whose input was `x` (data gradient = conv over the zero-dilated dy with rotated weights);
    `add` (same shape as dx) is summed in the epilogue.  `out`: tensor to write (may alias `add`).  keep_up (up == 2
    only): return the gradient at the up-sampled resolution [N,2H,2W,C] (add / out at that resolution) and leave the
    2x2 sum to the caller, so that several launches can accumulate before one vq_sumpool2 (callers that accumulate check
    `subpixel_up_eligible` first: the phase-decomposed form writes — and accumulates — at the low resolution directly).
    alpha (plain 3x3 path only): a host factor applied to the accumulator INSTEAD of the packed weight's 1/s_w — the result
    then carries the extra scale alpha * s_w (how _ResnetBlock keeps its branch gradient inside binary16's range).
    gn_bwd = (x_gn, stats, gamma, beta, groups, silu): dx is the dy of that GroupNorm (x_gn = its input, same shape as dx); where the
    library can (vq_conv2d_gnb_rows), the conv forms the GroupNorm-backward sums in its epilogue and the call returns (dx, part) with
    part = [N][rows][C][2] for gn_bwd_raw(part=...); (dx, None) otherwise."""
    n, h, w, cin = x.shape
    co_w, ci_w, r, s = weight.shape
    _, ho, wo, cout = dy.shape
    L = lib()
    _mfma_waits_for_side(dy)
    st = stream_of(dy)
    dt = dtype_code(dy)
    hv, wv = h * up, w * up
    if not keep_up and (ho, wo) == (hv, wv) and _subpixel_up(weight, stride, pad_t, pad_l, up):
        # Upsample: a plain 4x4 / stride-2 / pad-1 conv over dy with the summed taps; writes dx at the low resolution
        assert not mask_input_grad
        w4 = _derived_weight(weight, 1)
        dx = torch.empty((n, h, w, cin), dtype=dy.dtype, device=dy.device) if out is None else out
        assert dx.is_contiguous() and tuple(dx.shape) == (n, h, w, cin)
        d4 = _desc(n, ho, wo, cout, h, w, cin, co_w, ci_w, 4, 4, 2, 1, 1, 1, 1, dt, split, False)
        wp, sc = _packed(w4, "fwd", cin, cout, split, d4, _op(dy))
        d4.alpha_dev = _adev(sc)
        flops = 2.0 * n * h * w * co_w * ci_w * 16
        _launch("conv_igemm", flops, lambda: L.call("vq_conv2d_fwd", C.byref(d4), ptr(dy), ptr(wp), None, ptr(add), None,
                                                    ptr(dx), None, 0, st),
                _tag("dgrad", n, h, w, ci_w, co_w, r, stride, "2sub") if _launch_hook else "")
        return dx
    if _subpixel_down(weight, stride, pad_t, pad_l, up, h, w, ho, wo):
        # Downsample: per output parity a 2x2 conv over dy; rows = (phase, cin), depth-to-space store
        wd = _derived_weight(weight, 2)
        dx = torch.empty((n, h, w, cin), dtype=dy.dtype, device=dy.device) if out is None else out
        assert dx.is_contiguous() and tuple(dx.shape) == (n, h, w, cin)
        ds = _desc(n, ho, wo, cout, ho, wo, 4 * cin, co_w, 4 * ci_w, 2, 2, 1, 1, 1, 1, 1, dt, split, False, subpix=2)
        wp, sc = _packed(wd, "fwd", 4 * cin, cout, split, ds, _op(dy))
        ds.alpha_dev = _adev(sc)
        mask = x if mask_input_grad else None
        flops = 2.0 * n * ho * wo * co_w * ci_w * r * s
        _launch("conv_igemm", flops, lambda: L.call("vq_conv2d_fwd", C.byref(ds), ptr(dy), ptr(wp), None, ptr(add), ptr(mask),
                                                    ptr(dx), None, 0, st),
                _tag("dgrad", n, h, w, ci_w, co_w, r, stride, "1sub") if _launch_hook else "")
        return dx
    dd = _desc(n, ho, wo, cout, hv, wv, cin, co_w, ci_w, r, s, 1, stride, 1, r - 1 - pad_t, s - 1 - pad_l, dt, split, False)
    wp, sc = _packed(weight, "dgrad", cout, cin, split, dd, _op(dy))
    dd.alpha_dev = _adev(sc)
    if alpha is not None:
        dd.alpha, dd.alpha_dev = float(alpha), None
    direct = up == 1 or keep_up
    du = out if (direct and out is not None) else torch.empty((n, hv, wv, cin), dtype=dy.dtype, device=dy.device)
    assert du.is_contiguous() and tuple(du.shape) == (n, hv, wv, cin)
    mask = x if (mask_input_grad and up == 1) else None
    res = add if direct else None
    part = fuse = None
    if gn_bwd is not None and _gn_bwd_fused and direct and mask is None and res is None:
        rows = L.dll.vq_conv2d_gnb_rows(C.byref(dd))
        if rows > 0:
            xg, stats, gamma, beta, groups, silu = gn_bwd
            assert tuple(xg.shape) == tuple(du.shape) and xg.dtype == du.dtype and xg.is_contiguous()
            part = torch.empty((n, rows, cin, 2), dtype=torch.float32, device=dy.device)
            fuse = VqGnBwdFuse(ptr(xg), ptr(stats[0]), ptr(stats[1]), ptr(gamma), ptr(beta), ptr(part), int(groups), int(bool(silu)))
            dd.gn_bwd = C.addressof(fuse)
    flops = 2.0 * n * ho * wo * co_w * ci_w * r * s
    _launch("conv_igemm", flops, lambda: L.call("vq_conv2d_fwd", C.byref(dd), ptr(dy), ptr(wp), None, ptr(res), ptr(mask),
                                                ptr(du), None, 0, st),
            _tag("dgrad", n, h, w, ci_w, co_w, r, stride, up) if _launch_hook else "")
    if gn_bwd is not None:
        assert not (up == 2 and not keep_up)
        return du, part
    if up == 2 and not keep_up:
        assert not mask_input_grad and add is None
        dx = torch.empty_like(x) if out is None else out
        L.call("vq_sumpool2", ptr(du), ptr(dx), n, hv, wv, cin, dt, st)
        return dx
    return du


def sumpool2(du):
    """[N,2H,2W,C] -> [N,H,W,C]: the adjoint of the nearest-2x gather folded into an `up=2` conv."""
    n, hv, wv, c = du.shape
    dx = torch.empty((n, hv // 2, wv // 2, c), dtype=du.dtype, device=du.device)
    lib().call("vq_sumpool2", ptr(du), ptr(dx), n, hv, wv, c, dtype_code(du), stream_of(du))
    return dx


def conv_wgrad_into(x, dy, wshape, dw, db, acc, stride, pad_t, pad_l, up, split, gs=1.0):
    """Weight (and, when `db` is given, bias) gradient of one conv launch written into caller-owned fp32 tensors
    (`acc` = 1 adds to their contents).  Used where one parameter's gradient is assembled from several launches
    (the temporal taps of a 3-D conv); never touches the gradient sinks.  gs: loss scale carried by dy, removed here."""
    n, h, w, cin = x.shape
    co_w, ci_w, r, s = wshape
    _, ho, wo, cout = dy.shape
    L = lib()
    assert dw.is_contiguous() and dw.dtype == torch.float32 and tuple(dw.shape) == tuple(wshape)
    if _subpixel_wgrad and (ho, wo) == (2 * h, 2 * w) and _subpixel_up_shape(wshape, stride, pad_t, pad_l, up):
        _subpixel_wgrad_into(x.contiguous(), dy.contiguous(), co_w, ci_w, dw, acc, split, gs)
        if db is not None:
            _colsum(dy.contiguous(), db, co_w, acc, gs)
        return
    d = _desc(n, h, w, cin, ho, wo, cout, ci_w, co_w, r, s, stride, 1, up, pad_t, pad_l, dtype_code(dy), split, False,
              alpha=1.0 / gs, wgrad=True)
    ws = workspace(dy.device, L.size("vq_conv2d_wgrad_workspace", C.byref(d)))
    flops = 2.0 * n * ho * wo * co_w * ci_w * r * s
    _launch("conv_wgrad", flops, lambda: L.call("vq_conv2d_wgrad", C.byref(d), ptr(x), ptr(dy), ptr(dw), ptr(db), acc,
                                                ptr(ws), ws.numel(), stream_of(dy)),
            _tag("wgrad", n, h, w, ci_w, co_w, r, stride, up) if _launch_hook else "")


def conv_wgrad_raw(x, dy, weight, bias, stride, pad_t, pad_l, up, split, want_dw=True, want_db=True, gs=1.0, gs_dev=None):
    """-> (dw, db); an entry is None when it was not wanted or went into the parameter's gradient sink.
    gs: loss scale carried by dy (fp16 stacks), removed from both gradients in the kernels' epilogues; gs_dev: a device
    scalar the gradients are additionally MULTIPLIED by (VqConvDesc.alpha_dev; plain path only)."""
    want_db = want_db and bias is not None
    wsink, bsink = _sink_of(weight) if want_dw else None, _sink_of(bias) if want_db else None
    # every output goes into a gradient sink: the launches can leave the backward chain's stream (see _on_side_stream)
    sunk = (want_dw or want_db) and (wsink is not None or not want_dw) and (bsink is not None or not want_db)
    with _on_side_stream(sunk, dy, x, gs_dev if isinstance(gs_dev, torch.Tensor) else None) as side:
        out = _conv_wgrad_launches(x, dy, weight, bias, stride, pad_t, pad_l, up, split, want_dw, want_db, gs, gs_dev, wsink, bsink)
    for sink in (wsink, bsink):          # (on the chain's stream: a bucket that is now complete waits for the side stream itself)
        if sink is not None and sink[1] is not None:
            sink[1]()
    return out


def _conv_wgrad_launches(x, dy, weight, bias, stride, pad_t, pad_l, up, split, want_dw, want_db, gs, gs_dev, wsink, bsink):
    n, h, w, cin = x.shape
    co_w, ci_w, r, s = weight.shape
    _, ho, wo, cout = dy.shape
    L = lib()
    st = stream_of(dy)
    dt = dtype_code(dy)
    dw = db = None
    if want_db:
        db = bsink[0] if bsink else torch.empty(co_w, dtype=torch.float32, device=dy.device)
    if want_dw and _subpixel_wgrad and (ho, wo) == (2 * h, 2 * w) and _subpixel_up(weight, stride, pad_t, pad_l, up):
        # Upsample: weight gradient of the transposed form (4x4 / stride-2 conv over dy, roles swapped), folded onto the 3x3 taps
        dw = wsink[0] if wsink else torch.empty_like(weight, dtype=torch.float32)
        _subpixel_wgrad_into(x, dy, co_w, ci_w, dw, 1 if wsink else 0, split, gs)
        if want_db:
            _colsum(dy, db, co_w, 1 if bsink else 0, gs)
    elif want_dw:
        dw = wsink[0] if wsink else torch.empty_like(weight, dtype=torch.float32)
        # one accumulate flag per call: sinks accumulate, fresh tensors are overwritten
        acc = 1 if wsink else 0
        if want_db and (bsink is None) != (wsink is None):
            # mixed case (one sunk, one not): do the bias separately below
            db_here = None
        else:
            db_here = db
        d = _desc(n, h, w, cin, ho, wo, cout, ci_w, co_w, r, s, stride, 1, up, pad_t, pad_l, dt, split, False, alpha=1.0 / gs,
                  wgrad=True)
        d.alpha_dev = gs_dev
        ws = workspace(dy.device, L.size("vq_conv2d_wgrad_workspace", C.byref(d)))
        flops = 2.0 * n * ho * wo * co_w * ci_w * r * s
        _launch("conv_wgrad", flops, lambda: L.call("vq_conv2d_wgrad", C.byref(d), ptr(x), ptr(dy), ptr(dw), ptr(db_here), acc,
                                                    ptr(ws), ws.numel(), st),
                _tag("wgrad", n, h, w, ci_w, co_w, r, stride, up) if _launch_hook else "")
        if want_db and db_here is None:
            _colsum(dy, db, co_w, 1 if bsink else 0, gs, gs_dev)
    elif want_db:
        _colsum(dy, db, co_w, 1 if bsink else 0, gs, gs_dev)
    return (None if wsink else dw), (None if bsink else db)


def _colsum(dy, db, co_w, acc, gs=1.0, gs_dev=None):
    n, ho, wo, cout = dy.shape
    L = lib()
    pixels = n * ho * wo
    ws = workspace(dy.device, L.size("vq_colsum_workspace", pixels, cout))
    _launch("hbm:colsum", _nbytes(dy), lambda: L.call("vq_colsum", ptr(dy), pixels, cout, dtype_code(dy), ptr(db), co_w, acc, 1.0 / gs,
                                                      gs_dev, ptr(ws), ws.numel(), stream_of(dy)))


class _Conv2d(torch.autograd.Function):
    """y = conv(x, W) + b [+ residual] [relu].

    Contract for ReLU: a conv with relu=True returns post-ReLU y and expects the incoming dy to be
    already masked by (y > 0); consumers do that through `mask_input_grad` (their dx is zeroed
    where their input x <= 0) — see vq_conv2d_fwd's `relu_mask`.
    """

    @staticmethod
    def forward(ctx, x, weight, bias, residual, stride, pad_t, pad_l, up, relu, mask_input_grad, split, out_hw, gn=None):
        y = conv_fwd_raw(x, weight, bias, residual, stride, pad_t, pad_l, up, relu, split, out_hw, gn=gn)
        ctx.save_for_backward(x, weight, bias)
        ctx.cfg = (stride, pad_t, pad_l, up, mask_input_grad, split, residual is not None)
        ctx.prec = precision_of(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, bias = ctx.saved_tensors
        stride, pad_t, pad_l, up, mask_input_grad, split, has_res = ctx.cfg
        dy = dy.contiguous()
        dx = None
        with region(ctx.prec, backward=True):               # (the backward runs on the autograd thread: re-declare the stack, cf. _events)
            if ctx.needs_input_grad[0]:
                dx = _watch(ctx.prec, conv_dgrad_raw(dy, x, weight, stride, pad_t, pad_l, up, split, mask_input_grad))
            dw, db = conv_wgrad_raw(x, dy, weight, bias, stride, pad_t, pad_l, up, split, ctx.needs_input_grad[1],
                                    bias is not None and ctx.needs_input_grad[2], gs=ctx.prec.gs())
        dres = dy if (has_res and ctx.needs_input_grad[3]) else None
        return dx, dw, db, dres, None, None, None, None, None, None, None, None, None


def conv2d(x, weight, bias=None, *, residual=None, stride=1, pad=(0, 0), up=1, relu=False, mask_input_grad=False,
           split=1, out_hw=None, gn=None):
    """gn = (groups, eps): the output is the input of an FP32GroupNorm (see conv_fwd_raw)."""
    return _Conv2d.apply(x, weight, bias, residual, stride, pad[0], pad[1], up, relu, mask_input_grad, split, out_hw, gn)


# ----------------------------------------------------------------------------- 3-D convolution (tae.py)
# A 3x3x3 convolution over channels-last video activations [N,T,H,W,C] is run as its three TEMPORAL taps: each tap is
# a 3x3 implicit-GEMM conv over a run of frames (frames = the batch dimension of the 2-D kernels, so every tap keeps
# the K = 9*Cin shapes the MFMA kernels are tuned for) that accumulates into the output in place through the
# residual operand of the conv epilogue.  The centre tap covers every frame of every sample in ONE launch (it also
# carries the bias and the block residual); the two outer taps are shifted by one frame and therefore run per sample
# (T-1 frames each), which is what implements the zero padding in time.  The weight gradient of tap dt is likewise the
# 2-D wgrad over the same frame runs, accumulated over the samples; the data gradient is the mirrored chain.
_tap_cache: dict = {}


def _temporal_taps(weight: torch.Tensor) -> torch.Tensor:
    """[O,I,3,R,S] fp32 master -> tap-major copy [3,O,I,R,S] (each tap an OIHW weight the 2-D pack / conv entry points
    take), cached on (storage, version, optimizer generation); refreshed in place so the packed operands stay static."""
    key = (weight.data_ptr(), weight._version, _generation.get(weight.data_ptr(), 0), tuple(weight.shape), str(weight.device))
    hit = _tap_cache.get(weight.data_ptr())
    if hit is not None and hit[0] == key:
        return hit[1]
    w = weight.detach().permute(2, 0, 1, 3, 4)
    if hit is not None and hit[0][3:] == key[3:]:
        buf = hit[1]
        buf.copy_(w)
    else:
        buf = w.contiguous()
    _tap_cache[weight.data_ptr()] = (key, buf)
    return buf


def _frame_runs(a5, a_lo, b5, b_lo, cnt):
    """Frames [a_lo, a_lo+cnt) of every sample of a5 paired with frames [b_lo, b_lo+cnt) of b5, as 4-D [frames,H,W,C]
    views: one pair over all samples when both ranges are whole samples, else one pair per sample."""
    if cnt <= 0:
        return []
    if a_lo == 0 and b_lo == 0 and cnt == a5.shape[1] == b5.shape[1]:
        return [(a5.flatten(0, 1), b5.flatten(0, 1))]
    return [(a5[n, a_lo:a_lo + cnt], b5[n, b_lo:b_lo + cnt]) for n in range(a5.shape[0])]


def _conv3d_plan(mode, t):
    """-> (T_out, [(tap, first input frame, first output frame, count)]), the whole-sample tap first.
    same: zero padding 1 in time, stride 1 (tae.py:68-75).  down: F.pad (0,1) in time + stride 2, i.e. output frame j
    reads frames 2j+dt and the appended zero frame contributes nothing (tae.py:97-106)."""
    if mode == "same":
        return t, [(1, 0, 0, t), (0, 0, 1, t - 1), (2, 1, 0, t - 1)]
    if t < 2:
        raise ValueError("the 3-D Downsample needs at least two frames")
    t_out = (t - 2) // 2 + 1
    return t_out, [(dt, 0, 0, min(t_out, (t - 1 - dt) // 2 + 1)) for dt in range(3)]


class _Conv3d(torch.autograd.Function):
    """y = conv3d(x, W) + b [+ residual] for the three 3x3x3 layer kinds of tae.py: "same" (ResnetBlock / conv_in /
    conv_out, tae.py:68-75,137-139,167-169), "down" (tae.py:92-106) and "up" (nearest 2x in T, H and W, then "same":
    tae.py:109-120 — the spatial 2x is folded into the conv gather, the temporal 2x is a frame copy of the small
    pre-upsample tensor)."""

    @staticmethod
    def _sources(x, mode, plan):
        """Per tap the 5-D tensor whose frame runs feed the 2-D conv."""
        if mode == "down":      # frames dt, dt+2, ... gathered into a dense batch (the 2-D kernels take one batch stride)
            return {dt: x[:, dt:dt + 2 * cnt:2].contiguous() for dt, _, _, cnt in plan if cnt > 0}
        src = x.repeat_interleave(2, dim=1) if mode == "up" else x
        return {dt: src for dt, _, _, _ in plan}

    @staticmethod
    def forward(ctx, x, weight, bias, residual, mode, split):
        n, t, h, w, cin = x.shape
        co_w, ci_w, kt, r, s = weight.shape
        assert (kt, r, s) == (3, 3, 3) and mode in ("same", "down", "up")
        if x.dtype in HALF_RANGE:
            raise NotImplementedError("the 3-D convolutions run in bf16 / fp32 storage (fp16 loss-scale plumbing is 2-D only)")
        x = x.contiguous()
        taps = _temporal_taps(weight)
        up, stride, pad = (2 if mode == "up" else 1), (2 if mode == "down" else 1), (0 if mode == "down" else 1)
        t_out, plan = _conv3d_plan("down" if mode == "down" else "same", t * up)
        ho, wo = ((h - 2) // 2 + 1, (w - 2) // 2 + 1) if mode == "down" else (h * up, w * up)
        y = torch.empty((n, t_out, ho, wo, pad8(co_w)), dtype=x.dtype, device=x.device)
        res = residual.contiguous() if residual is not None else None
        src = _Conv3d._sources(x, mode, plan)
        for i, (dt, a_lo, y_lo, cnt) in enumerate(plan):
            for xa, ya in _frame_runs(src.get(dt, x), a_lo, y, y_lo, cnt):
                first = i == 0          # one run over every output frame: bias, block residual, initialises y
                conv_fwd_raw(xa, taps[dt], bias if first else None, (res.flatten(0, 1) if res is not None else None) if first
                             else ya, stride, pad, pad, up, False, split, (ho, wo), out=ya)
        ctx.save_for_backward(x, weight, bias)
        ctx.cfg = (mode, split, residual is not None, t_out, ho, wo)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, bias = ctx.saved_tensors
        mode, split, has_res, t_out, ho, wo = ctx.cfg
        n, t, h, w, cin = x.shape
        co_w, ci_w = weight.shape[:2]
        dy = dy.contiguous()
        taps = _temporal_taps(weight)
        up, stride, pad = (2 if mode == "up" else 1), (2 if mode == "down" else 1), (0 if mode == "down" else 1)
        _, plan = _conv3d_plan("down" if mode == "down" else "same", t * up)
        src = _Conv3d._sources(x, mode, plan)
        # ---- weight / bias gradient: tap dt over the same frame runs as the forward
        dw = db = None
        if ctx.needs_input_grad[1] or (bias is not None and ctx.needs_input_grad[2]):
            dwt = torch.empty((3, co_w, ci_w, 3, 3), dtype=torch.float32, device=x.device)
            db = torch.empty(co_w, dtype=torch.float32, device=x.device) if bias is not None else None
            for i, (dt, a_lo, y_lo, cnt) in enumerate(plan):
                runs = _frame_runs(src.get(dt, x), a_lo, dy, y_lo, cnt)
                if not runs:
                    dwt[dt].zero_()
                for k, (xa, ga) in enumerate(runs):
                    conv_wgrad_into(xa, ga, (co_w, ci_w, 3, 3), dwt[dt], db if i == 0 else None, 1 if k else 0, stride, pad, pad,
                                    up, split)
            dw = dwt.permute(1, 2, 0, 3, 4).contiguous()
        # ---- data gradient: the mirrored chain, accumulated in place through the epilogue's residual operand
        dx = None
        if ctx.needs_input_grad[0]:
            if mode == "down":
                dx = torch.empty_like(x)
                parts = {0: torch.zeros((n, (t + 1) // 2, h, w, cin), dtype=x.dtype, device=x.device),     # even frames
                         1: torch.zeros((n, t // 2, h, w, cin), dtype=x.dtype, device=x.device)}            # odd frames
                for dt, _, _, cnt in plan:
                    tgt, lo = parts[dt & 1], dt >> 1        # input frame 2j+dt is frame j + (dt >> 1) of its parity class
                    for ga, da in _frame_runs(dy, 0, tgt, lo, cnt):
                        conv_dgrad_raw(ga, da, taps[dt], 2, 0, 0, 1, split, False, add=da, out=da)
                dx[:, 0::2] = parts[0]
                dx[:, 1::2] = parts[1]
            else:
                ts = t * up
                # phase-decomposed Upsample: every tap's 4x4 / stride-2 conv accumulates at the LOW spatial resolution
                low = up == 2 and subpixel_up_eligible(taps[0])
                du = torch.empty((n, ts, h, w, cin) if low else (n, ts, h * up, w * up, cin), dtype=x.dtype, device=x.device)
                for i, (dt, a_lo, y_lo, cnt) in enumerate(plan):
                    for ga, da in _frame_runs(dy, y_lo, du, a_lo, cnt):
                        conv_dgrad_raw(ga, _ShapeOnly(ga.shape[0], h, w, cin), taps[dt], 1, 1, 1, up, split, False, add=None if i == 0 else da, out=da,
                                       keep_up=up == 2 and not low)
                if low:         # the two frame copies of the temporal nearest-2x
                    du = du.view(n, t, 2, h, w, cin).sum(2)
                elif up == 2:   # 2x2 spatial sum in the HIP kernel, then the temporal pair sum
                    du = sumpool2(du.flatten(0, 1)).view(n, t, 2, h, w, cin).sum(2)
                dx = du
        dres = dy if (has_res and ctx.needs_input_grad[3]) else None
        return dx, dw, (db if ctx.needs_input_grad[2] else None), dres, None, None


class _ShapeOnly:
    """Stands in for the forward input where conv_dgrad_raw only needs its shape."""

    def __init__(self, *shape):
        self.shape = shape


def conv3d(x, weight, bias=None, *, residual=None, mode="same"):
    """x: [N,T,H,W,pad8(Cin)] channels-last; weight [Cout,Cin,3,3,3] fp32 (nn.Conv3d layout)."""
    return _Conv3d.apply(x, weight, bias, residual, mode, split_for(x))


def clear_tap_cache() -> None:
    _tap_cache.clear()


# ----------------------------------------------------------------------------- GroupNorm + swish
def gn_fwd_raw(x, gamma, beta, groups, eps, silu):
    n, h, w, c = x.shape
    if n * (c // groups) * h * w == 1:      # F.group_norm's own refusal (torch/nn/functional.py _verify_batch_size), behind ae.py:41-53
        raise ValueError(f"Expected more than 1 value per channel when training, got input size {[n * c // groups, groups, h, w]}")
    x = x.contiguous()
    L = lib()
    st = stream_of(x)
    hw = h * w
    pre = getattr(x, "_vq_gn", None)          # statistics reduced by the epilogue of the convolution that wrote x
    if pre is not None and pre[1] == groups and pre[2] == float(eps) and pre[0].shape[1] == n * groups:
        stats = pre[0]
    else:
        ws = workspace(x.device, L.size("vq_gn_workspace", n, hw, c))
        stats = torch.empty((2, n * groups), dtype=torch.float32, device=x.device)
        _launch("hbm:gn_stats", _nbytes(x), lambda: L.call("vq_gn_stats", ptr(x), n, hw, c, groups, float(eps), dtype_code(x),
                                                           ptr(stats[0]), ptr(stats[1]), ptr(ws), ws.numel(), st))
    y = torch.empty_like(x)
    _launch("hbm:gn_apply", _nbytes(x, y), lambda: L.call("vq_gn_silu_fwd", ptr(x), ptr(stats[0]), ptr(stats[1]), ptr(gamma),
                                                          ptr(beta), n, hw, c, groups, c, dtype_code(x), int(silu), ptr(y), st))
    return y, stats


def gn_bwd_raw(x, dy, stats, gamma, beta, groups, silu, add=None, want_param_grads=True, gs=1.0, dx_scale=(1.0, None),
               pg_dev=None, part=None):
    """-> (dx, dgamma, dbeta); dx = dx_scale * d(silu∘gn)·dy (+ add).  Parameter grads go to their sinks when registered.
    gs: loss scale carried by dy — removed from dgamma / dbeta (times the device scalar pg_dev when given);
    dx_scale = (host factor, device scalar or None): see vq_gn_silu_bwd.
    part: the per-channel sums formed by the conv that produced dy (conv_dgrad_raw(gn_bwd=...)): the reduction pass is skipped."""
    n, h, w, c = x.shape
    L = lib()
    st = stream_of(dy)
    hw = h * w
    ws = workspace(x.device, L.size("vq_gn_workspace", n, hw, c))
    gsink, bsink = (_sink_of(gamma), _sink_of(beta)) if want_param_grads else (None, None)
    sunk = gsink is not None and bsink is not None
    dx = torch.empty_like(x)
    dg = gsink[0] if sunk else torch.empty(c, dtype=torch.float32, device=x.device)
    db = bsink[0] if sunk else torch.empty(c, dtype=torch.float32, device=x.device)
    # algorithmic bytes: reads of x and dy, one write of dx (+ the skip gradient where it is folded in)
    _launch("hbm:gn_bwd", _nbytes(x, dy, dx, add),
            lambda: L.call("vq_gn_silu_bwd", ptr(x), ptr(dy), ptr(stats[0]), ptr(stats[1]), ptr(gamma), ptr(beta), ptr(add), n, hw, c,
                           groups, c, dtype_code(x), int(silu), ptr(dx), ptr(dg), ptr(db), 1 if sunk else 0, float(dx_scale[0]),
                           dx_scale[1], 1.0 / gs, pg_dev, _events(), ptr(part), 0 if part is None else int(part.shape[1]),
                           ptr(ws), ws.numel(), st))
    if sunk:
        for sink in (gsink, bsink):
            if sink[1] is not None:
                sink[1]()
        return dx, None, None
    return dx, dg, db


class _GroupNormSilu(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, groups, eps, silu):
        y, stats = gn_fwd_raw(x, gamma, beta, groups, eps, silu)
        ctx.save_for_backward(x, stats, gamma, beta)
        ctx.cfg = (groups, silu)
        ctx.prec = precision_of(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, stats, gamma, beta = ctx.saved_tensors
        groups, silu = ctx.cfg
        with region(ctx.prec, backward=True):
            dx, dg, db = gn_bwd_raw(x, dy.contiguous(), stats, gamma, beta, groups, silu, gs=ctx.prec.gs())
        return _watch(ctx.prec, dx), dg, db, None, None, None


class _GroupNormSiluFork(torch.autograd.Function):
    """GN(x) and x itself, for the blocks whose input fans out (AttnBlock: ae.py:92 `x + proj_out(...)`; tae.py's ResnetBlock): the
    backward folds the skip gradient into the GroupNorm backward kernel (`add`) like _ResnetBlock does, instead of leaving the sum at
    the fan-out to autograd — torch would add two VQ_F16X2 carriers as complex32 numbers, piece by piece in binary16 (3e-4 on dx)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, groups, eps, silu):
        y, stats = gn_fwd_raw(x, gamma, beta, groups, eps, silu)
        ctx.save_for_backward(x, stats, gamma, beta)
        ctx.cfg = (groups, silu)
        ctx.prec = precision_of(x)
        return y, x.view_as(x)

    @staticmethod
    def backward(ctx, dy, dskip):
        x, stats, gamma, beta = ctx.saved_tensors
        groups, silu = ctx.cfg
        if dy is None:
            return dskip, None, None, None, None, None
        with region(ctx.prec, backward=True):
            dx, dg, db = gn_bwd_raw(x, dy.contiguous(), stats, gamma, beta, groups, silu, gs=ctx.prec.gs(),
                                    add=None if dskip is None else dskip.contiguous())
        return _watch(ctx.prec, dx), dg, db, None, None, None


def group_norm_silu(x, gamma, beta, groups=32, eps=1e-6, silu=True, fork=False):
    """fork=True: -> (GN(x), x): use the second output wherever the block reads x again (see _GroupNormSiluFork)."""
    assert gamma.dtype == torch.float32 and beta.dtype == torch.float32
    if fork:
        return _GroupNormSiluFork.apply(x, gamma, beta, groups, eps, silu)
    return _GroupNormSilu.apply(x, gamma, beta, groups, eps, silu)


class _ResnetBlock(torch.autograd.Function):
    """ae.py:124-140 as ONE autograd node: x(+1x1) + conv2(swish(GN2(conv1(swish(GN1(x)))))).
    The hand-sequenced backward folds the skip gradient into the GroupNorm backward kernel (`add`) instead
    of letting autograd launch a separate elementwise add at the fan-out of x."""

    @staticmethod
    def forward(ctx, x, n1w, n1b, c1w, c1b, n2w, n2b, c2w, c2b, sw, sb, groups, eps, split):
        a1, st1 = gn_fwd_raw(x, n1w, n1b, groups, eps, True)
        h1 = conv_fwd_raw(a1, c1w, c1b, None, 1, 1, 1, 1, False, split, None, gn=(groups, eps))      # -> norm2
        a2, st2 = gn_fwd_raw(h1, n2w, n2b, groups, eps, True)
        skip = x if sw is None else conv_fwd_raw(x, sw, sb, None, 1, 0, 0, 1, False, split, None)
        # the block output is the input of the next block's norm1 (or norm_out) wherever a GroupNorm follows: ae.py:131,254,330
        out = conv_fwd_raw(a2, c2w, c2b, skip, 1, 1, 1, 1, False, split, None, gn=(groups, eps))
        ctx.save_for_backward(x, a1, st1, h1, a2, st2, n1w, n1b, c1w, c1b, n2w, n2b, c2w, c2b, sw, sb)
        ctx.cfg = (groups, split)
        ctx.prec = precision_of(x)
        return out

    @staticmethod
    def backward(ctx, dout):
        with region(ctx.prec, backward=True):               # the autograd thread: re-declare the stack (loss scale, range-event counters)
            return _ResnetBlock._backward(ctx, dout)

    @staticmethod
    def _backward(ctx, dout):
        x, a1, st1, h1, a2, st2, n1w, n1b, c1w, c1b, n2w, n2b, c2w, c2b, sw, sb = ctx.saved_tensors
        groups, split = ctx.cfg
        dout = dout.contiguous()
        ng = ctx.needs_input_grad
        prec = ctx.prec
        gs = prec.gs()                       # loss scale of dout / dx (1 outside the fp16 stacks)
        # fp16 stacks: the gradient of the conv branch is kept in the units of a conv2 whose weights are normalised to
        # |w|max ~ 1 — da2 = _BRANCH_GAIN * (accumulator over the s_w-scaled operand) instead of acc / s_w, i.e. the branch
        # tensors da2, dh1, da1 carry the extra factor rho = _BRANCH_GAIN * s_w(conv2).  The reference initialises conv2 with
        # std 1e-4 / out_ch (ae.py:119-121): in natural units the branch gradient would sit 2^-20 below the skip gradient and
        # fall out of binary16's range, taking conv1's and norm2's parameter gradients with it.  rho comes back out (exactly:
        # powers of two) in the parameter gradients of the branch and where the branch rejoins the skip gradient.
        rho_inv, bg = (1.0, None), None
        if prec.half_range() and _branch_rebase:
            da2, part2 = conv_dgrad_raw(dout, a2, c2w, 1, 1, 1, 1, split, False, alpha=_BRANCH_GAIN, gn_bwd=(h1, st2, n2w, n2b, groups, True))
            bg = _adev(packed_scale(c2w, "dgrad", _op(dout)))     # device scalar 1 / s_w(conv2)
            rho_inv = (1.0 / _BRANCH_GAIN, bg)
        else:
            da2, part2 = conv_dgrad_raw(dout, a2, c2w, 1, 1, 1, 1, split, False, gn_bwd=(h1, st2, n2w, n2b, groups, True))
        gb = gs * _BRANCH_GAIN if bg is not None else gs         # host part of the branch tensors' scale
        _watch(prec, da2)
        # (the weight gradient goes to the side stream AFTER the data gradient that shares its dy: the GroupNorm backward that
        # follows then has it for company; issued before the data gradient — the two GEMMs side by side, the GroupNorm pass alone —
        # measured 271.0 vs 271.8 img/s, profiles/r4j_*)
        dc2w, dc2b = conv_wgrad_raw(a2, dout, c2w, c2b, 1, 1, 1, 1, split, ng[7], ng[8], gs=gs)
        dh1, dn2w, dn2b = gn_bwd_raw(h1, da2, st2, n2w, n2b, groups, True, gs=gb, pg_dev=bg, part=part2)
        da1, part1 = conv_dgrad_raw(dh1, a1, c1w, 1, 1, 1, 1, split, False, gn_bwd=(x, st1, n1w, n1b, groups, True))
        _watch(prec, dh1)
        _watch(prec, da1)
        dc1w, dc1b = conv_wgrad_raw(a1, dh1, c1w, c1b, 1, 1, 1, 1, split, ng[3], ng[4], gs=gb, gs_dev=bg)
        dsw = dsb = None
        if sw is None:
            dskip = dout
        else:
            dskip = conv_dgrad_raw(dout, x, sw, 1, 0, 0, 1, split, False) if ng[0] else None
            dsw, dsb = conv_wgrad_raw(x, dout, sw, sb, 1, 0, 0, 1, split, ng[9], ng[10], gs=gs)
        dx, dn1w, dn1b = gn_bwd_raw(x, da1, st1, n1w, n1b, groups, True, add=dskip, gs=gb, pg_dev=bg, dx_scale=rho_inv, part=part1)
        return _watch(prec, dx), dn1w, dn1b, dc1w, dc1b, dn2w, dn2b, dc2w, dc2b, dsw, dsb, None, None, None


# accumulator -> stored units of the conv-branch gradient in fp16 stacks: the s_w-scaled conv2 operand has |w|max in [2^14, 2^15)
# and a 3x3 conv over 128..512 channels sums ~2^5 coherent-ish terms, so 2^-18 keeps da2 at the magnitude of dout
_BRANCH_GAIN = 2.0 ** -18
_branch_rebase = True


def set_branch_rebase(on: bool) -> None:
    """Test knob: False keeps the conv-branch gradient of fp16 ResnetBlocks in natural units (what the re-basing is there for
    shows with the reference's own initialisation, tests/test_model.py)."""
    global _branch_rebase
    _branch_rebase = bool(on)


def resnet_block(x, norm1, conv1, norm2, conv2, shortcut=None):
    sw, sb = (shortcut.weight, shortcut.bias) if shortcut is not None else (None, None)
    return _ResnetBlock.apply(x, norm1.weight, norm1.bias, conv1.weight, conv1.bias, norm2.weight, norm2.bias,
                              conv2.weight, conv2.bias, sw, sb, norm1.num_groups, norm1.eps, split_for(x))


# ----------------------------------------------------------------------------- pooling
def _maxpool_bwd(x, dy, add, prec):
    n, h, w, c = x.shape
    dx = torch.empty_like(x)
    with region(prec, backward=True):
        ev = _events()
    _launch("hbm:maxpool", _nbytes(x, dy, dx, add), lambda: lib().call("vq_maxpool2_bwd", ptr(x), ptr(dy), ptr(add), ptr(dx), n, h, w, c,
                                                                     dtype_code(x), ev, stream_of(x)))
    return dx


class _MaxPool2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        n, h, w, c = x.shape
        x = x.contiguous()
        y = torch.empty((n, h // 2, w // 2, c), dtype=x.dtype, device=x.device)
        _launch("hbm:maxpool", _nbytes(x, y), lambda: lib().call("vq_maxpool2_fwd", ptr(x), ptr(y), n, h, w, c, dtype_code(x), stream_of(x)))
        ctx.save_for_backward(x)
        ctx.prec = precision_of(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return _maxpool_bwd(x, dy.contiguous(), None, ctx.prec)


def max_pool2(x):
    return _MaxPool2.apply(x)


class _PoolWithTap(torch.autograd.Function):
    """(tap, pooled) = (x, maxpool2(x)) for a feature map with TWO consumers: every VGG slice output feeds the next slice through
    the pool AND an LPIPS tap / a discriminator head (utils.py:116-131,187-203).  As two autograd edges out of `x`, autograd sums
    the two gradients with an elementwise kernel of its own (12 launches per step at configs[2], 0.7 % of the GPU time, and in
    binary16 an UNSATURATED add); as one node the sum rides in the pool's backward kernel: dx = route(d_pooled) + d_tap."""

    @staticmethod
    def forward(ctx, x):
        n, h, w, c = x.shape
        x = x.contiguous()
        y = torch.empty((n, h // 2, w // 2, c), dtype=x.dtype, device=x.device)
        _launch("hbm:maxpool", _nbytes(x, y), lambda: lib().call("vq_maxpool2_fwd", ptr(x), ptr(y), n, h, w, c, dtype_code(x), stream_of(x)))
        ctx.save_for_backward(x)
        ctx.prec = precision_of(x)
        ctx.set_materialize_grads(False)
        return x.view_as(x), y

    @staticmethod
    def backward(ctx, d_tap, d_pool):
        (x,) = ctx.saved_tensors
        if d_pool is None:
            return d_tap
        add = d_tap.contiguous() if d_tap is not None else None
        return _watch(ctx.prec, _maxpool_bwd(x, d_pool.contiguous(), add, ctx.prec))


def pool_with_tap(x):
    """-> (x as the tap, max_pool2(x)); see _PoolWithTap."""
    return _PoolWithTap.apply(x)


# ----------------------------------------------------------------------------- LPIPS tap
class _LpipsTap(torch.autograd.Function):
    """val[n] = mean_hw sum_c w_c m_c (f0/(|f0|+eps) - f1/(|f1|+eps))^2 ; gradient to f0 only
    (utils.py:41 — the target branch carries no gradient in the trainer either)."""

    @staticmethod
    def forward(ctx, f0, f1, w, mask, seed):
        n, h, wd, c = f0.shape
        f0, f1 = f0.contiguous(), f1.contiguous()
        L = lib()
        hw = h * wd
        ws = workspace(f0.device, L.size("vq_lpips_workspace", n, hw))
        val = torch.zeros(n, dtype=torch.float32, device=f0.device)
        w32 = w.detach().float().reshape(-1).contiguous()
        _launch("hbm:lpips_tap", _nbytes(f0, f1), lambda: L.call("vq_lpips_tap_fwd", ptr(f0), ptr(f1), ptr(w32), ptr(mask), int(seed),
                                                                 n, hw, c, dtype_code(f0), ptr(val), ptr(ws), ws.numel(), stream_of(f0)))
        ctx.save_for_backward(f0, f1, w32, mask if mask is not None else torch.empty(0))
        ctx.seed = int(seed)
        ctx.prec = precision_of(f0)
        return val

    @staticmethod
    def backward(ctx, gval):
        f0, f1, w32, mask = ctx.saved_tensors
        mask = mask if mask.numel() else None
        n, h, wd, c = f0.shape
        # The taps are ReLU outputs; per the consumer contract of _Conv2d the tap masks its own
        # gradient with (f0 > 0).
        df0 = torch.empty_like(f0)
        g = gval.contiguous().float()
        with region(ctx.prec, backward=True):
            ev = _events()
        _launch("hbm:lpips_tap", _nbytes(f0, f1, df0),
                lambda: lib().call("vq_lpips_tap_bwd", ptr(f0), ptr(f1), ptr(w32), ptr(mask), ctx.seed, ptr(g), n, h * wd, c,
                                   dtype_code(f0), 1, ctx.prec.gs(), ptr(df0), ev, stream_of(f0)))
        return _watch(ctx.prec, df0), None, None, None, None


def lpips_tap(f0, f1, w, mask=None, seed=0):
    return _LpipsTap.apply(f0, f1, w, mask, seed)


# ----------------------------------------------------------------------------- GradNorm
class _GradNorm(torch.autograd.Function):
    """vae_trainer.py:27-48 without host syncs: the norm stays on the device; the cross-rank mean of
    the per-rank norms is a 4-byte all-reduce issued on the same stream."""

    @staticmethod
    def forward(ctx, x, weight, group, dp_chunks):
        ctx.weight = float(weight)
        ctx.group = group
        ctx.dp_chunks = int(dp_chunks)
        return x.clone()

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous().float()
        L = lib()
        st = stream_of(g)
        scratch = torch.empty(1024, dtype=torch.float32, device=g.device)
        norm = torch.empty(1, dtype=torch.float32, device=g.device)
        k = ctx.dp_chunks
        if k > 1:
            # One process standing in for k data-parallel ranks (the DP-equivalence check of tests/test_distributed.py): rank r
            # would hold batch chunk r, its loss a mean over B/k samples (gradients k times these), and the k ranks would
            # average their per-rank norms (vae_trainer.py:40-44) and then their gradients:  dx = g / sum_r ||g_r||  (SURVEY §4.4).
            assert g.shape[0] % k == 0
            parts = torch.empty(k, dtype=torch.float32, device=g.device)
            n = g.numel() // k
            for r in range(k):
                L.call("vq_l2norm", ptr(g[r * (g.shape[0] // k)]), n, C.c_void_p(parts.data_ptr() + 4 * r), ptr(scratch), st)
            norm = parts.sum().reshape(1)
        else:
            _launch("hbm:gradnorm", _nbytes(g), lambda: L.call("vq_l2norm", ptr(g), g.numel(), ptr(norm), ptr(scratch), st))
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(ctx.group) > 1:
            dist.all_reduce(norm, op=dist.ReduceOp.SUM, group=ctx.group)
            norm = norm / dist.get_world_size(ctx.group)
        dx = torch.empty_like(g)
        _launch("hbm:gradnorm", _nbytes(g, dx), lambda: L.call("vq_scale_by_norm", ptr(g), ptr(norm), ctx.weight, g.numel(), ptr(dx), st))
        return dx, None, None, None


def gradnorm(x, weight=1.0, group=None, dp_chunks=1):
    return _GradNorm.apply(x, weight, group, dp_chunks)
