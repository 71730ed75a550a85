"""Micro-benchmark of the patch-conv data gradients (PatchDiscriminator heads, utils.py:156-185) at the configs[2] shapes:
dx[n, R*y + r, S*x + s, :] = W[:, :, r, s]^T dy[n, y, x, :] — a store-bound 1x1 conv with a depth-to-space epilogue.
VQ_TILE = VqConvDesc.kernel_hint of the descriptors (768 = 48 << 4: the generic tile kernels instead of the persistent one)."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vqgan_training_amd as vq
from vqgan_training_amd import ops
from vqgan_training_amd._lib import lib, ptr, stream_of, dtype_code

dev = torch.device("cuda:0")
L = lib()
ops._hint_conv = int(os.environ.get("VQ_TILE", "0"))
SHAPES = [(32, 64, 32, 256, 4), (16, 64, 32, 256, 4), (32, 128, 64, 128, 4), (32, 256, 128, 64, 2)]    # B, Cin, Cout, H (input), k
for prec_name in ("fp16", "bf16"):
    prec = ops._PRECISIONS[prec_name]
    for (B, ci, co, h, k) in SHAPES:
        ho = h // k
        w = torch.randn(co, ci, k, k, device=dev) / (ci * k * k) ** 0.5
        dy = torch.randn(B, ho, ho, co, device=dev).to(prec.dtype)
        # data gradient as a conv over the k-fold zero-dilated dy with full padding (ops.conv_dgrad_raw's descriptor)
        dd = ops._desc(B, ho, ho, co, h, h, ci, co, ci, k, k, 1, k, 1, k - 1, k - 1, dtype_code(dy), prec.split, False)
        wpd, scd = ops._packed(w, "dgrad", co, ci, prec.split, dd, ops._op(dy))
        dd.alpha_dev = ops._adev(scd)
        dx = torch.empty(B, h, h, ci, device=dev, dtype=prec.dtype)
        st = stream_of(dy)
        fn = lambda: L.call("vq_conv2d_fwd", C.byref(dd), ptr(dy), ptr(wpd), None, None, None, ptr(dx), None, 0, st)
        fn(); torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(30): fn()
        e.record(); torch.cuda.synchronize()
        t = s.elapsed_time(e) / 30
        mb = (dx.numel() + dy.numel()) * 2 / 1e6
        print(f"{prec_name} B={B} {ci:3d}->{co:3d} k{k} in {h}x{h}: dgrad {t * 1e3:7.1f} us  {mb / t / 1e3:6.2f} TB/s on {mb:.0f} MB")
