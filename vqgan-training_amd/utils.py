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


_VGG_WEIGHT_FILES = ("vgg16.pth", "vgg16-397923af.pth", "vgg16_features.pth")     # looked up in the working directory


def find_vgg16_weights(path=None):
    """Where the ImageNet VGG16 weights come from: an explicit path, $VQ_VGG16_WEIGHTS, or one of the usual file names next to
    the run (torchvision's `vgg16-397923af.pth`).  The reference gets them from torchvision's download cache (utils.py:95,148:
    `models.vgg16(pretrained=True)`); there is no network on the build / bench machines, so a file it is."""
    if path:                   # an explicit path must exist: no silent fall-through to another file
        if not os.path.exists(path):
            raise FileNotFoundError(f"VGG16 backbone weights '{path}' not found")
        return path
    for cand in (os.environ.get("VQ_VGG16_WEIGHTS"),) + _VGG_WEIGHT_FILES:
        if cand and os.path.exists(cand):
            return cand
    return None


def load_vgg16_backbone(module: nn.Module, weights, prefix: str) -> int:
    """Copy a torchvision `vgg16` state dict (keys `features.{idx}.weight|bias`, or the bare `{idx}.*` of `.features`) onto
    the 13 convolutions of `module` whose parameters are named `{prefix}{slice}…{idx}.*`: prefix "net.slice" for LPIPS
    (utils.py:100-111), "slice" for the PatchDiscriminator (`slice{k}.0.{idx}`, utils.py:150-154).  `weights`: a path or a
    state dict.  All 26 tensors must be present with the right shapes; returns the number copied."""
    sd = torch.load(weights, map_location="cpu") if isinstance(weights, (str, os.PathLike)) else weights
    sd = {(k[len("features."):] if k.startswith("features.") else k): v for k, v in sd.items()}
    own = dict(module.named_parameters())
    n = 0
    with torch.no_grad():
        for si, spec in enumerate(_VGG_SLICES):
            for idx, cin, cout in spec:
                for leaf in ("weight", "bias"):
                    names = [k for k in own if k.startswith(f"{prefix}{si + 1}.") and k.endswith(f".{idx}.{leaf}")]
                    if len(names) != 1:
                        raise KeyError(f"{type(module).__name__}: no unique parameter for VGG16 features.{idx}.{leaf}")
                    src = sd.get(f"{idx}.{leaf}")
                    if src is None or tuple(src.shape) != tuple(own[names[0]].shape):
                        raise KeyError(f"VGG16 weights: features.{idx}.{leaf} missing or mis-shaped "
                                       f"({None if src is None else tuple(src.shape)} vs {tuple(own[names[0]].shape)})")
                    own[names[0]].copy_(src.to(own[names[0]].dtype))
                    n += 1
    ops.clear_pack_cache()
    return n


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
    last = len(slices) - 1
    for si, sl in enumerate(slices):
        for conv in sl:
            h = conv(h, relu=True, mask_input_grad=not first)
            first = False
        if si < last:
            # the slice output has two consumers (its tap and, through the pool, the next slice): one autograd node, so that the
            # two gradients are summed inside the pool's backward kernel (ops._PoolWithTap)
            tap, h = ops.pool_with_tap(h)
            taps.append(tap)
        else:
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

    def __init__(self, use_dropout=True, precision=None, pretrained_path="vgg.pth", backbone_path=None):
        super().__init__()
        self.scaling_layer = ScalingLayer()
        self.chns = [64, 128, 256, 512, 512]
        self.net = vgg16(pretrained=True, requires_grad=False)
        for i, c in enumerate(self.chns):
            setattr(self, f"lin{i}", NetLinLayer(c, use_dropout=use_dropout))
        self.use_dropout = use_dropout
        self.precision = precision
        self.backbone_loaded = False
        self.load_from_pretrained(pretrained_path, backbone_path)
        for p in self.parameters():
            p.requires_grad = False

    def load_from_pretrained(self, path="vgg.pth", backbone_path=None):
        """utils.py:24-37 loads `vgg.pth` (the 5 lin weights) with strict=False on top of torchvision's ImageNet VGG16
        (utils.py:95).  No download here: `vgg.pth` from `path`, the backbone from `backbone_path` / $VQ_VGG16_WEIGHTS / the
        usual file names (find_vgg16_weights); whatever is missing stays at its seeded random initialisation — loudly."""
        got_backbone = False
        if path and os.path.exists(path):
            data = torch.load(path, map_location="cpu")
            res = self.load_state_dict(data, strict=False)
            loaded = set(data) - set(res.unexpected_keys)
            got_backbone = any(k.startswith("net.") for k in loaded)          # a full LPIPS checkpoint carries it too
            if not any(k.startswith("lin") for k in loaded):
                warnings.warn(f"LPIPS: '{path}' holds none of the lin weights (keys: {sorted(data)[:4]} ...)")
        else:
            warnings.warn(f"LPIPS: '{path}' not found — keeping seeded random VGG/lin weights (no network access)")
            with torch.no_grad():      # the real LPIPS lin weights are non-negative (a weighted squared distance):
                for i in range(len(self.chns)):      # keep the stand-ins so, or the "distance" can go negative
                    getattr(self, f"lin{i}").weight.abs_()
        found = find_vgg16_weights(backbone_path)
        if found:
            load_vgg16_backbone(self, found, "net.slice")
            got_backbone = True
        self.backbone_loaded = got_backbone
        if not got_backbone and path and os.path.exists(path):
            warnings.warn("LPIPS: the lin weights were loaded but the VGG16 backbone (net.slice*) is at its random "
                          "initialisation: the perceptual loss is not LPIPS.  Pass backbone_path= / --vgg_backbone_path / "
                          "$VQ_VGG16_WEIGHTS (torchvision's vgg16-397923af.pth).")

    def target_features(self, target):
        """The VGG taps of the TARGET image (utils.py:116-131 under no_grad): they depend on the input batch alone, so the train step
        requests them early, on the side stream (ops.run_on_side_stream), and hands them to forward(target_feats=)."""
        prec = ops.resolve_precision(self.precision)
        with ops.region(prec), torch.no_grad():
            return self.net(self.scaling_layer(target, prec))

    def forward(self, input, target, masks=None, target_feats=None):
        prec = ops.resolve_precision(self.precision)
        with ops.region(prec):
            f_in = self.net(self.scaling_layer(input, prec))
            if target_feats is not None:
                f_tg = target_feats.wait() if hasattr(target_feats, "wait") else target_feats
            else:
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

    def __init__(self, precision=None, backbone_path=None):
        super().__init__()
        self.scaling_layer = ScalingLayer()
        for i, spec in enumerate(_VGG_SLICES):
            setattr(self, f"slice{i + 1}", nn.Sequential(_vgg_slice(spec)))      # key: slice{i}.0.{idx}.*
        self._backbone_path = backbone_path

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
        # utils.py:148: the discriminator starts from the ImageNet VGG16 features too (and then trains them)
        found = find_vgg16_weights(self._backbone_path)
        self.backbone_loaded = bool(found)
        if found:
            load_vgg16_backbone(self, found, "slice")


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
