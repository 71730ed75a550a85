"""ctypes binding of libvqhip.so (include/vqhip.h).

The product path has exactly one compute backend: the HIP library built for gfx950.  If the
shared object is missing or does not export the ABI we expect, importing this module's `lib()`
raises — there is no eager/PyTorch/CPU fallback anywhere in the package.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

import torch

VQ_BF16 = 0
VQ_F32 = 1
VQ_F16 = 2
VQ_F16X2 = 3      # two binary16 pieces per value (hi, lo), carried by torch.complex32 tensors: 4 bytes per element, same shapes
ABI_VERSION = 10

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libvqhip.so")


class VqConvDesc(C.Structure):
    """Mirror of `struct VqConvDesc` (include/vqhip.h)."""

    _fields_ = [(n, C.c_int32) for n in (
        "N", "H", "W", "Cin", "Ho", "Wo", "Cout", "Cin_w", "Cout_w", "R", "S",
        "stride", "dil_in", "up", "pad_t", "pad_l", "dtype", "split", "relu", "subpix")] + \
        [("alpha", C.c_float), ("kernel_hint", C.c_int32), ("alpha_dev", C.c_void_p), ("range_events", C.c_void_p),
         ("gn_bwd", C.c_void_p)]


class VqGnBwdFuse(C.Structure):
    """Mirror of `struct VqGnBwdFuse` (include/vqhip.h): the GroupNorm whose backward sums a data-gradient conv forms in its epilogue."""

    _fields_ = [("x", C.c_void_p), ("mean", C.c_void_p), ("rstd", C.c_void_p), ("gamma", C.c_void_p), ("beta", C.c_void_p),
                ("part", C.c_void_p), ("groups", C.c_int32), ("silu", C.c_int32)]


class VqAdamTensor(C.Structure):
    _fields_ = [("p", C.c_void_p), ("g", C.c_void_p), ("m", C.c_void_p), ("v", C.c_void_p), ("n", C.c_int64)]


class VqPackJob(C.Structure):
    """Mirror of `struct VqPackJob` (include/vqhip.h)."""

    _fields_ = [("w", C.c_void_p), ("out", C.c_void_p), ("total", C.c_int64), ("block_start", C.c_int64)] + \
               [(n, C.c_int32) for n in ("Cout_w", "Cin_w", "R", "S", "rows_pad", "kch_pad", "Kp", "split", "dgrad", "layout",
                                         "tiled")] + [("n_units", C.c_int64), ("op_dtype", C.c_int32), ("reserved0", C.c_int32),
                                                      ("scale", C.c_void_p)]


_P = C.c_void_p
_I = C.c_int
_L = C.c_int64
_F = C.c_float
_Z = C.c_size_t
_U64 = C.c_uint64
_DP = C.POINTER(VqConvDesc)

# name -> (restype, argtypes); every symbol include/vqhip.h declares
_SIGNATURES = {
    "vq_last_error": (C.c_char_p, []),
    "vq_abi_version": (_I, []),
    "vq_packed_weight_elems": (_Z, [_I, _I, _I, _I, _I, _I]),
    "vq_conv_weight_layout": (_I, [_P]),
    "vq_pack_job": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P]),
    "vq_pack_job_blocks": (_L, [_P]),
    "vq_pack_weights_multi": (_I, [_P, _I, _L, _I, _P]),
    "vq_pack_weight_fwd": (_I, [_P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P]),
    "vq_pack_weight_dgrad": (_I, [_P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P]),
    "vq_subpixel_weights": (_I, [_P, _P, _I, _I, _I, _P]),
    "vq_subpixel_wgrad_fold": (_I, [_P, _P, _I, _I, _I, _P]),
    "vq_attention_fwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "vq_attention_workspace": (_Z, [_I, _I, _I, _I]),
    "vq_attention_bwd": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _Z, _P]),
    "vq_wavelet_fwd": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "vq_flip_nchw": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "vq_area_downsample_nchw": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "vq_conv2d_fwd": (_I, [_DP, _P, _P, _P, _P, _P, _P, _P, _I, _P]),
    "vq_conv2d_gn_tile": (_I, [_DP, _I]),
    "vq_conv2d_gnb_rows": (_I, [_DP]),
    "vq_gn_stats_finalize": (_I, [_P, _I, _I, _L, _I, _I, _F, _P, _P, _P]),
    "vq_conv2d_wgrad_workspace": (_Z, [_DP]),
    "vq_conv2d_wgrad": (_I, [_DP, _P, _P, _P, _P, _I, _P, _Z, _P]),
    "vq_colsum_workspace": (_Z, [_L, _I]),
    "vq_colsum": (_I, [_P, _L, _I, _I, _P, _I, _I, _F, _P, _P, _Z, _P]),
    "vq_nchw_to_nhwc": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _P, _P, _F, _P, _P]),
    "vq_nhwc_to_nchw": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _P, _F, _P]),
    "vq_absmax": (_I, [_P, _L, _I, _P, _P]),
    "vq_gn_workspace": (_Z, [_I, _L, _I]),
    "vq_gn_stats": (_I, [_P, _I, _L, _I, _I, _F, _I, _P, _P, _P, _Z, _P]),
    "vq_gn_silu_fwd": (_I, [_P, _P, _P, _P, _P, _I, _L, _I, _I, _I, _I, _I, _P, _P]),
    "vq_gn_silu_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _L, _I, _I, _I, _I, _I, _P, _P, _P, _I, _F, _P, _F, _P, _P, _P, _I, _P, _Z, _P]),
    "vq_maxpool2_fwd": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "vq_maxpool2_bwd": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P]),
    "vq_sumpool2": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "vq_lpips_workspace": (_Z, [_I, _L]),
    "vq_lpips_tap_fwd": (_I, [_P, _P, _P, _P, _U64, _I, _L, _I, _I, _P, _P, _Z, _P]),
    "vq_lpips_tap_bwd": (_I, [_P, _P, _P, _P, _U64, _P, _I, _L, _I, _I, _I, _F, _P, _P, _P]),
    "vq_moments": (_I, [_P, _L, _P, _P, _P]),
    "vq_l2norm": (_I, [_P, _L, _P, _P, _P]),
    "vq_scale_by_norm": (_I, [_P, _P, _F, _L, _P, _P]),
    "vq_gan_disc_loss": (_I, [_P, _P, _L, _I, _P, _P, _P, _P]),
    "vq_adamw_multi": (_I, [_P, _P, _I, _L, _I, _F, _F, _F, _F, _F, _F, _F, _F, _P, _I, _I, _P]),
    "vq_scale": (_I, [_P, _F, _P, _L, _P, _P]),
    "vq_vq_workspace": (_Z, [_L, _I]),
    "vq_vq_nearest_fwd": (_I, [_P, _P, _L, _I, _I, _P, _P, _P, _P, _Z, _P]),
    "vq_vq_scatter_workspace": (_Z, [_I, _I]),
    "vq_vq_scatter_add": (_I, [_P, _P, _L, _I, _I, _P, _P, _Z, _P]),
    "vq_debug_probe": (_I, [_I, _P, _P, _P]),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)


class VqLibrary:
    """A loaded libvqhip with typed entry points; `call` raises RuntimeError on non-zero status."""

    def __init__(self, path: str):
        if not os.path.exists(path):
            raise RuntimeError(
                f"libvqhip.so not found at {path}: the HIP extension is required (run `make` or "
                f"`python -c 'import __graft_entry__ as g; g.build()'`); there is no fallback path.")
        self.path = path
        self.dll = C.CDLL(path)
        for name, (res, args) in _SIGNATURES.items():
            try:
                fn = getattr(self.dll, name)
            except AttributeError as e:  # pragma: no cover - build problem
                raise RuntimeError(f"{path} does not export {name}") from e
            fn.restype = res
            fn.argtypes = args
        got = self.dll.vq_abi_version()
        if got != ABI_VERSION:
            raise RuntimeError(f"{path}: ABI version {got}, host code expects {ABI_VERSION}")

    def last_error(self) -> str:
        return (self.dll.vq_last_error() or b"").decode()

    def call(self, name: str, *args):
        rc = getattr(self.dll, name)(*args)
        if rc != 0:
            raise RuntimeError(f"{name} failed ({rc}): {self.last_error()}")

    def size(self, name: str, *args) -> int:
        return int(getattr(self.dll, name)(*args))


_lock = threading.Lock()
_LIB: VqLibrary | None = None


def lib() -> VqLibrary:
    global _LIB
    if _LIB is None:
        with _lock:
            if _LIB is None:
                _LIB = VqLibrary(_LIB_PATH)
    return _LIB


def _set_library_for_tests(library: VqLibrary | None) -> None:
    """Test hook (tests/ only): inject the host-emulated build of the same kernel sources."""
    global _LIB
    _LIB = library


def ptr(t: torch.Tensor | None):
    return None if t is None else C.c_void_p(t.data_ptr())


def stream_of(t: torch.Tensor):
    if t.is_cuda:
        return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)
    return None


def dtype_code(t: torch.Tensor) -> int:
    if t.dtype == torch.bfloat16:
        return VQ_BF16
    if t.dtype == torch.float32:
        return VQ_F32
    if t.dtype == torch.float16:
        return VQ_F16
    if t.dtype == torch.complex32:       # a carrier only: torch never computes on these tensors (include/vqhip.h, VQ_F16X2)
        return VQ_F16X2
    raise TypeError(f"unsupported storage dtype {t.dtype}")


_ws_cache: dict = {}


_ws_slot = threading.local()       # ops' side-stream context sets .v = 1: launches on the second stream get their own scratch buffers


def workspace(device: torch.device, nbytes: int, slot: int | None = None) -> torch.Tensor:
    """Grow-only scratch buffer per (device, slot); all users of a slot enqueue on one stream in order (slot None = the slot of the
    stream context the caller runs in: 0 on the main stream, 1 inside ops' side-stream context)."""
    if slot is None:
        slot = getattr(_ws_slot, "v", 0)
    key = (str(device), slot)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        nbytes = max(int(nbytes * 1.25), 1 << 20)
        buf = torch.empty(nbytes, dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf
