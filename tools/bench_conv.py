"""Micro-benchmark of the conv kernels at the config-2 layer shapes.
Usage: python tools/bench_conv.py [bf16|fp16|f16x3|fp32|fp32x3] [B] [number of shapes]   (VQ_TILE / VQ_WGTILE: forced tiles;
VQ_ZERO=1: all-zero operands — the chip clocks to its power budget, so the gap to random data is the DVFS share)"""
import ctypes as C
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vqgan_training_amd as vq
from vqgan_training_amd import ops
from vqgan_training_amd._lib import lib, ptr, stream_of, dtype_code, workspace

prec = ops._PRECISIONS[sys.argv[1] if len(sys.argv) > 1 else "bf16"]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
dev = torch.device("cuda:0")
SHAPES = [  # Cin, Cout, H(out), R, stride, up
    (128, 128, 256, 3, 1, 1), (256, 256, 128, 3, 1, 1), (512, 512, 64, 3, 1, 1), (512, 512, 32, 3, 1, 1),
    (512, 512, 128, 3, 1, 2), (256, 256, 256, 3, 1, 2), (512, 256, 128, 3, 1, 1), (256, 128, 256, 3, 1, 1),
    (128, 256, 128, 3, 1, 1), (256, 128, 256, 1, 1, 1), (128, 8, 256, 3, 1, 1), (8, 128, 256, 3, 1, 1),
    (64, 64, 256, 3, 1, 1), (512, 512, 16, 3, 1, 1), (512, 512, 8, 3, 1, 1), (8, 64, 256, 3, 1, 1),
    (64, 64, 512, 3, 1, 1), (128, 64, 512, 3, 1, 1),      # 16, 17: the HR decoder's last level of the reference's launch line (l1)
]

def timeit(fn, iters=int(os.environ.get("VQ_ITERS", "20"))):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters

if os.environ.get("VQ_ABLATE_LIB"):      # tools only: the `make ablate` library (experimental kernels, profiling ablations)
    vq._lib._set_library_for_tests(vq._lib.VqLibrary(os.environ["VQ_ABLATE_LIB"]))
L = lib()
# VqConvDesc.kernel_hint of the descriptors below (include/vqhip.h): VQ_TILE -> forward / data gradient, VQ_WGTILE (+ VQ_WGSPLIT << 16)
# -> weight gradient
ops._hint_conv = int(os.environ.get('VQ_TILE', '0'))
ops._hint_wgrad = int(os.environ.get('VQ_WGTILE', '0')) + (int(os.environ.get('VQ_WGSPLIT', '0')) << 16)
sel = sys.argv[3] if len(sys.argv) > 3 else str(len(SHAPES))
shapes = [SHAPES[int(i)] for i in sel.split(",")] if "," in sel else SHAPES[:int(sel)]
for (ci, co, ho, r, stride, up) in shapes:
    hi = ho // up * stride
    zero = 0.0 if os.environ.get("VQ_ZERO", "0") not in ("", "0") else 1.0
    # (through the layout kernel: the only way to fill a VQ_F16X2 tensor; the other storage types get the same values)
    x = ops.to_nhwc(torch.randn(B, ci, hi, hi, device=dev) * zero, prec).detach()
    w = (torch.randn(co, ci, r, r, device=dev) / (ci * r * r) ** 0.5) * zero
    dy = ops.to_nhwc(torch.randn(B, co, ho, ho, device=dev) * zero, prec).detach()
    pad = r // 2
    d = ops._desc(B, hi, hi, ci, ho, ho, co, ci, co, r, r, stride, 1, up, pad, pad, dtype_code(x), prec.split, False)
    wp, sc = ops._packed(w, "fwd", co, ci, prec.split, d, ops._op(x))
    d.alpha_dev = ops._adev(sc)
    y = torch.empty(B, ho, ho, co, device=dev, dtype=prec.dtype)
    st = stream_of(x)
    flops = 2.0 * B * ho * ho * co * ci * r * r
    t_f = timeit(lambda: L.call("vq_conv2d_fwd", C.byref(d), ptr(x), ptr(wp), None, None, None, ptr(y), None, 0, st))
    dd = ops._desc(B, ho, ho, co, ho, ho, ci, co, ci, r, r, 1, stride, 1, r - 1 - pad, r - 1 - pad, dtype_code(x), prec.split, False)
    wpd, scd = ops._packed(w, "dgrad", co, ci, prec.split, dd, ops._op(x))
    dd.alpha_dev = ops._adev(scd)
    du = torch.empty(B, ho, ho, ci, device=dev, dtype=prec.dtype)
    t_d = timeit(lambda: L.call("vq_conv2d_fwd", C.byref(dd), ptr(dy), ptr(wpd), None, None, None, ptr(du), None, 0, st))
    dwg = ops._desc(B, hi, hi, ci, ho, ho, co, ci, co, r, r, stride, 1, up, pad, pad, dtype_code(x), prec.split, False, wgrad=True)
    need = L.size("vq_conv2d_wgrad_workspace", C.byref(dwg))
    ws = workspace(dev, need)
    dw = torch.empty_like(w)
    t_w = timeit(lambda: L.call("vq_conv2d_wgrad", C.byref(dwg), ptr(x), ptr(dy), ptr(dw), None, 0, ptr(ws), ws.numel(), st))
    print(f"{prec.name} B={B} {ci:4d}->{co:4d} @{ho:3d} k{r} up{up}: fwd {t_f:7.3f} ms {flops/t_f/1e9:7.1f} TF | dgrad {t_d:7.3f} ms {flops/t_d/1e9:7.1f} TF | wgrad {t_w:7.3f} ms {flops/t_w/1e9:7.1f} TF (~{need / (4.0 * r * r * co * ci):.1f} splits)", flush=True)
