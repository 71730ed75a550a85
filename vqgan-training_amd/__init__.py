"""MI355X-native VAE/VQGAN train-step hot path (drop-in for cloneofsimo/vqgan-training's
ae.py / utils.py / vae_trainer.py surface).  Compute lives in libvqhip.so (csrc/*.hip)."""
from . import _lib, ops  # noqa: F401
