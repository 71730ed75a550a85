"""MI355X-native VAE/VQGAN train-step hot path — drop-in for cloneofsimo/vqgan-training's
ae.py / utils.py / vae_trainer.py surface.  All compute lives in libvqhip.so (csrc/*.hip, C ABI in
include/vqhip.h); importing the package never falls back to PyTorch kernels."""
from . import _lib, ops  # noqa: F401
from . import ae, tae, utils, optim, distributed, quantizer, vae_trainer  # noqa: F401
from .ae import VAE  # noqa: F401
from .utils import LPIPS, PatchDiscriminator  # noqa: F401
