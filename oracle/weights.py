"""ORACLE — TEST INFRASTRUCTURE ONLY.  Deterministic, RNG-library-independent tensors.

SURVEY F11: the reference's fresh initialisation hides kernel bugs (every ResnetBlock.conv2 has
std 1e-4/out_ch, discriminator heads are zero), so parity runs re-randomise every tensor.  Values come
from a splitmix64 hash of (seed, tensor name, element index) evaluated with numpy uint64 arithmetic —
bit-identical on any machine, nothing to store in the fixtures.
"""
from __future__ import annotations

import zlib

import numpy as np
import torch


def hash_uniform(n: int, seed: int) -> np.ndarray:
    """n doubles in [0,1)."""
    with np.errstate(over="ignore"):
        z = (np.arange(n, dtype=np.uint64) + np.uint64(seed & 0xFFFFFFFFFFFFFFFF)) * np.uint64(0x9E3779B97F4A7C15)
        z ^= z >> np.uint64(30)
        z *= np.uint64(0xBF58476D1CE4E5B9)
        z ^= z >> np.uint64(27)
        z *= np.uint64(0x94D049BB133111EB)
        z ^= z >> np.uint64(31)
    return (z >> np.uint64(11)).astype(np.float64) / float(1 << 53)


def uniform_tensor(shape, seed: int, lo=-1.0, hi=1.0) -> torch.Tensor:
    n = int(np.prod(shape)) if len(shape) else 1
    u = hash_uniform(n, seed) * (hi - lo) + lo
    return torch.from_numpy(u.astype(np.float32)).reshape(shape)


def _name_seed(name: str, seed: int) -> int:
    return (zlib.crc32(name.encode()) * 1000003 + seed * 7919) & 0x7FFFFFFFFFFFFFFF


PHOTO_BIAS_SCALE = 30.0        # VAE stacks of the photograph fixtures: a DC of up to +-30 behind every conv (activation std ~1)
PHOTO_VGG_BIAS_SCALE = 3.0     # ReLU stacks (LPIPS, discriminator): a larger negative DC would simply switch a layer off


def randomize_state_dict(sd: dict, seed: int = 0, relu_net: bool = False, bias_scale: float = 1.0) -> dict:
    """Same keys/shapes as `sd`, every float tensor replaced (buffers like ScalingLayer's kept):
    conv (2-D and 3-D) weights U(+-sqrt(3/fan_in)) (x sqrt(2) for ReLU stacks), biases U(+-0.1),
    GroupNorm gamma U(0.5,1.5), beta U(+-0.2).
    bias_scale > 1: "trained-like" conv biases — every conv bias tensor gets a common offset of bias_scale * U(+-1) (hash of its name)
    on top of per-channel U(+-0.1) * bias_scale^(1/2), so that the tensor it feeds has |mean| >> std per GroupNorm group (the
    zero-mean default never exercises that: round-5 verdict, weak 1b)."""
    out = {}
    for k, v in sd.items():
        s = _name_seed(k, seed)
        if "scaling_layer" in k or not v.dtype.is_floating_point:
            out[k] = v.clone()
        elif v.dim() in (4, 5):
            fan_in = v[0].numel()
            a = (3.0 / fan_in) ** 0.5 * (2.0 ** 0.5 if relu_net else 1.0)
            if k.startswith("lin"):
                out[k] = uniform_tensor(v.shape, s, 0.0, 1.0)          # LPIPS lin weights are non-negative
            else:
                out[k] = uniform_tensor(v.shape, s, -a, a)
        elif "norm" in k and k.endswith("weight"):
            out[k] = uniform_tensor(v.shape, s, 0.5, 1.5)
        elif "norm" in k and k.endswith("bias"):
            out[k] = uniform_tensor(v.shape, s, -0.2, 0.2)
        else:
            out[k] = uniform_tensor(v.shape, s, -0.1, 0.1)
            if bias_scale != 1.0:
                dc = float(hash_uniform(1, s ^ 0x5DEECE66D)[0] * 2.0 - 1.0) * bias_scale
                out[k] = out[k] * float(bias_scale) ** 0.5 + dc
    return out


_PHOTOS = None


def photo_batch(indices, res: int = 256) -> torch.Tensor:
    """Photographs of the reference's own sample set (tests/golden/photos_256.npz, made by tests/golden/make_golden.py photos) in [-1, 1]
    (ToTensor + Normalize(0.5, 0.5): vae_trainer.py:98-99); res < 256: area-resized like vae_trainer.py:531-533."""
    global _PHOTOS
    if _PHOTOS is None:
        import os
        path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "photos_256.npz")
        _PHOTOS = np.load(path)["images"]
    x = torch.from_numpy(_PHOTOS[list(indices)].astype(np.float32)) / 255.0 * 2.0 - 1.0
    if res != 256:
        x = torch.nn.functional.interpolate(x, size=(res, res), mode="area")
    return x


def image_batch(b: int, res: int, seed: int = 42) -> torch.Tensor:
    """Synthetic images in [-1,1] (vae_trainer.py:98,108 normalisation range)."""
    return uniform_tensor((b, 3, res, res), seed * 104729 + 17, -1.0, 1.0)
