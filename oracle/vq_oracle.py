"""ORACLE — TEST INFRASTRUCTURE ONLY.  ctypes front-end of oracle/vq_oracle.c (+ the quantizer losses).

The reference has no quantizer => parity unpinned against the reference; the lookup and the losses below are pinned to
the published VQGAN arithmetic and to fp64 brute force by tests/test_vq_oracle_pin.py.  vq_oracle.c defines the lookup, this
file restates the standard VQGAN quantizer around it (row A12 of SURVEY §8(a)):
  z_q = e[idx];  loss = beta*mean((z_q.detach()-z)^2) + mean((z_q-z.detach())^2), beta = 0.25;
  output z + (z_q - z).detach()  (straight-through).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SO = os.path.join(ROOT, "oracle", "_ref", "libvq_oracle.so")
_lib = None


def _load():
    global _lib
    if _lib is None:
        src = os.path.join(ROOT, "oracle", "vq_oracle.c")
        if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
            subprocess.run(["make", "-C", ROOT, "oracle"], check=True, stdout=subprocess.DEVNULL)
        _lib = C.CDLL(_SO)
        _lib.vq_nearest_oracle.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        _lib.vq_nearest_oracle.restype = None
    return _lib


def nearest(z: torch.Tensor, codebook: torch.Tensor):
    """z [n, D] fp32, codebook [K, D] fp32 (CPU) -> (idx int64 [n], min_dist fp32 [n])."""
    z = z.detach().float().contiguous().cpu()
    cb = codebook.detach().float().contiguous().cpu()
    n, d = z.shape
    idx = torch.empty(n, dtype=torch.int64)
    md = torch.empty(n, dtype=torch.float32)
    _load().vq_nearest_oracle(z.data_ptr(), cb.data_ptr(), n, cb.shape[0], d, idx.data_ptr(), md.data_ptr())
    return idx, md


def quantize(z_nchw: torch.Tensor, codebook: torch.Tensor, beta: float = 0.25):
    """Standard VQGAN quantizer on [B,D,h,w]; differentiable wrt z and codebook (plain torch)."""
    b, d, h, w = z_nchw.shape
    tokens = z_nchw.permute(0, 2, 3, 1).reshape(-1, d)
    idx, _ = nearest(tokens, codebook)
    zq = codebook[idx]
    loss = beta * ((zq.detach() - tokens) ** 2).mean() + ((zq - tokens.detach()) ** 2).mean()
    out = tokens + (zq - tokens).detach()
    return out.reshape(b, h, w, d).permute(0, 3, 1, 2), loss, idx.reshape(b, h, w)
