"""Epilogue cost of the implicit-GEMM conv: plain / +bias / +bias+residual / +relu-mask, 128->128 @256^2 and 512->512 @64^2."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vqgan_training_amd as vq
from vqgan_training_amd import ops
from vqgan_training_amd._lib import lib, ptr, stream_of, dtype_code
dev = torch.device("cuda:0"); L = lib(); B = 16
ops._hint_conv = int(os.environ.get("VQ_TILE", "0"))
def timeit(fn, it=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / it
for (c, h) in [(128, 256), (256, 128), (512, 64)]:
    x = torch.randn(B, h, h, c, device=dev).to(torch.bfloat16); w = torch.randn(c, c, 3, 3, device=dev) / (c * 9) ** 0.5
    bias = torch.randn(c, device=dev); res = torch.randn_like(x); y = torch.empty_like(x)
    d = ops._desc(B, h, h, c, h, h, c, c, c, 3, 3, 1, 1, 1, 1, 1, dtype_code(x), 1, False)
    wp = ops._packed(w, "fwd", c, c, 1, d)[0]; st = stream_of(x); fl = 2.0 * B * h * h * c * c * 9
    row = L.dll.vq_conv2d_gn_tile(C.byref(d), 32)
    part = torch.empty(B, h * h // max(row, 1), 32, 2, device=dev)
    for name, b_, r_, m_, g_ in (("plain", None, None, None, None), ("bias", bias, None, None, None), ("bias+res", bias, res, None, None),
                                 ("mask", None, None, res, None), ("bias+res+gn", bias, res, None, part if row > 0 else None)):
        t = timeit(lambda: L.call("vq_conv2d_fwd", C.byref(d), ptr(x), ptr(wp), ptr(b_), ptr(r_), ptr(m_), ptr(y), ptr(g_), 32 if g_ is not None else 0, st))
        print(f"{c}->{c} @{h}: {name:11s} {t*1e3:7.1f} us {fl/t/1e9:7.1f} TF", flush=True)
