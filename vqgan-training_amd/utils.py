"""LPIPS / VGG16 / PatchDiscriminator on the HIP path — drop-in for the reference's utils.py surface.

State-dict keys match the reference (so `vgg.pth` and reference checkpoints load):
  LPIPS:              scaling_layer.{shift,scale}, net.slice{1..5}.{idx}.{weight,bias}, lin{0..4}.model.1.weight
  PatchDiscriminator: scaling_layer.*, slice{1..5}.0.{idx}.*, binary_classifier{1..5}.{0,2}.*
where idx are torchvision's vgg16().features indices (0,2 | 5,7 | 10,12,14 | 17,19,21 | 24,26,28).
torchvision is not a dependency: the VGG16-D feature stack is declared here; ImageNet weights are
loaded from a checkpoint when one is given, otherwise the seeded default initialisation is used
(there is no network on the build/bench machines — bench.py says so in its `data` field).
"""
from __future__ import annotations

import os
import warnings

import torch
from torch import nn

from . import ops
from .ae import StandardizedC2d

# torchvision.models.vgg16 cfg "D" up to features[29] (relu5_3): (features index, Cin, Cout) per slice;
# every conv is followed by ReLU; slices 2..5 start with MaxPool2d(2,2) (features idx 4,9,16,23).
_VGG_SLICES = (
    ((0, 3, 64), (2, 64, 64)),
    ((5, 64, 128), (7, 128, 128)),
    ((10, 128, 256), (12, 256, 256), (14, 256, 256)),
    ((17, 256, 512), (19, 512, 512), (21, 512, 512)),
    ((24, 512, 512), (26, 512, 512), (28, 512, 512)),
)


class ScalingLayer(nn.Module):
    """utils.py:60-71.  Applied inside the NCHW->NHWC conversion kernel (zero padding of the first
    conv therefore happens after scaling, as in the reference)."""

    def __init__(self):
        super().__init__()
        self.register_buffer("shift", torch.Tensor([-0.030, -0.088, -0.188])[None, :, None, None])
        self.register_buffer("scale", torch.Tensor([0.458, 0.448, 0.450])[None, :, None, None])

    def forward(self, inp, precision=None):
        """[B,3,H,W] fp32 -> scaled NHWC tensor (channels padded to 8)."""
        return ops.to_nhwc(inp, precision, self.shift.reshape(-1).contiguous(), self.scale.reshape(-1).contiguous())


def _vgg_slice(spec) -> nn.Sequential:
    seq = nn.Sequential()
    for idx, cin, cout in spec:
        seq.add_module(str(idx), StandardizedC2d(cin, cout, kernel_size=3, stride=1, padding=1))
    return seq


def _run_vgg(slices, h):
    """13 x [3x3 conv + bias + ReLU] with 4 max-pools; returns the 5 taps (relu1_2 ... relu5_3)."""
    taps = []
    first = True
    for si, sl in enumerate(slices):
        if si > 0:
            h = ops.max_pool2(h)
        for conv in sl:
            h = conv(h, relu=True, mask_input_grad=not first)
            first = False
        taps.append(h)
    return taps


class vgg16(nn.Module):
    """utils.py:92-131 (the LPIPS backbone)."""

    def __init__(self, requires_grad=False, pretrained=True):
        super().__init__()
        for i, spec in enumerate(_VGG_SLICES):
            setattr(self, f"slice{i + 1}", _vgg_slice(spec))
        self.N_slices = 5
        if not requires_grad:
            for p in self.parameters():
                p.requires_grad = False

    def slices(self):
        return [getattr(self, f"slice{i}") for i in range(1, 6)]

    def forward(self, h):
        return _run_vgg(self.slices(), h)


class NetLinLayer(nn.Module):
    """utils.py:74-89: [Dropout,] 1x1 conv C->1 without bias; evaluated inside the tap kernel."""

    def __init__(self, chn_in, chn_out=1, use_dropout=False):
        super().__init__()
        layers = [nn.Dropout()] if use_dropout else []
        layers += [nn.Conv2d(chn_in, chn_out, 1, stride=1, padding=0, bias=False)]
        self.model = nn.Sequential(*layers)

    @property
    def weight(self):
        return self.model[-1].weight


class LPIPS(nn.Module):
    """utils.py:8-57.  forward(input, target) -> [B,1,1,1].

    Like the reference, the module is left in train mode by the trainer, so Dropout(0.5) is live on
    every tap: the keep-mask is generated inside the tap kernel from a per-call seed.
    `.eval()` gives the deterministic metric; `forward(..., masks=[...])` injects explicit masks.
    """

    def __init__(self, use_dropout=True, precision=None, pretrained_path="vgg.pth"):
        super().__init__()
        self.scaling_layer = ScalingLayer()
        self.chns = [64, 128, 256, 512, 512]
        self.net = vgg16(pretrained=True, requires_grad=False)
        for i, c in enumerate(self.chns):
            setattr(self, f"lin{i}", NetLinLayer(c, use_dropout=use_dropout))
        self.use_dropout = use_dropout
        self.precision = precision
        self.load_from_pretrained(pretrained_path)
        for p in self.parameters():
            p.requires_grad = False

    def load_from_pretrained(self, path="vgg.pth"):
        """utils.py:24-37 loads `vgg.pth` (the 5 lin weights) with strict=False; no download here."""
        if path and os.path.exists(path):
            self.load_state_dict(torch.load(path, map_location="cpu"), strict=False)
        else:
            warnings.warn(f"LPIPS: '{path}' not found — keeping seeded random VGG/lin weights (no network access)")
            with torch.no_grad():      # the real LPIPS lin weights are non-negative (a weighted squared distance):
                for i in range(len(self.chns)):      # keep the stand-ins so, or the "distance" can go negative
                    getattr(self, f"lin{i}").weight.abs_()

    def forward(self, input, target, masks=None):
        prec = ops.resolve_precision(self.precision)
        with ops.region(prec):
            f_in = self.net(self.scaling_layer(input, prec))
            with torch.no_grad():
                f_tg = self.net(self.scaling_layer(target, prec))
            val = None
            for k in range(len(self.chns)):
                lin = getattr(self, f"lin{k}")
                mask, seed = None, 0
                if masks is not None:
                    mask = masks[k]
                elif self.use_dropout and self.training:
                    seed = int(torch.randint(1, 2 ** 62, (1,)).item())
                v = ops.lpips_tap(f_in[k], f_tg[k], lin.weight, mask, seed)
                val = v if val is None else val + v
            return val.view(-1, 1, 1, 1)


class PatchDiscriminator(nn.Module):
    """utils.py:143-203: trainable VGG16 features + 5 non-overlapping patch-conv heads, summed.
    forward(x [B,3,H,W]) -> [B, (H/16)*(W/16)] logits."""

    def __init__(self, precision=None):
        super().__init__()
        self.scaling_layer = ScalingLayer()
        for i, spec in enumerate(_VGG_SLICES):
            setattr(self, f"slice{i + 1}", nn.Sequential(_vgg_slice(spec)))      # key: slice{i}.0.{idx}.*

        def head(cin, mid, k1, k2):
            if mid is None:
                seq = nn.Sequential(StandardizedC2d(cin, 1, kernel_size=k1, stride=k1, padding=0, bias=True))
            else:
                seq = nn.Sequential(StandardizedC2d(cin, mid, kernel_size=k1, stride=k1, padding=0, bias=True), nn.ReLU(),
                                    StandardizedC2d(mid, 1, kernel_size=k2, stride=k2, padding=0, bias=True))
            nn.init.zeros_(seq[-1].weight)                                         # utils.py:161,168,175,180,185
            return seq

        self.binary_classifier1 = head(64, 32, 4, 4)
        self.binary_classifier2 = head(128, 64, 4, 2)
        self.binary_classifier3 = head(256, 128, 2, 2)
        self.binary_classifier4 = head(512, None, 2, None)
        self.binary_classifier5 = head(512, None, 1, None)
        self.precision = precision

    def forward(self, x):
        prec = ops.resolve_precision(self.precision)
        with ops.region(prec):
            h = self.scaling_layer(x, prec)
            feats = _run_vgg([getattr(self, f"slice{i}")[0] for i in range(1, 6)], h)
            out = None
            for k, f in enumerate(feats):
                seq = getattr(self, f"binary_classifier{k + 1}")
                if len(seq) == 1:
                    o = seq[0](f, mask_input_grad=True)
                else:
                    o = seq[2](seq[0](f, relu=True, mask_input_grad=True), mask_input_grad=True)
                o = ops.to_nchw(o, 1).flatten(1)
                out = o if out is None else out + o
            return out


def prepare_filter(device):
    """utils.py:224-226 moves the wavelet filter bank to the device.  The HIP kernel carries the four fixed filters
    (utils.py:206-219) in constant memory, so there is nothing to move; kept for call-site compatibility."""
    return None


def wavelet_transform_multi_channel(x, levels=4):
    """utils.py:229-247: [B,C,H,W] -> [B,4C,H/2,W/2] (zero-pad 2, four separable 6x6 filters, stride 2; `levels` is
    unused in the reference too).  Forward-only: the trainer applies it to the image batch."""
    return ops.wavelet_nchw(x)
