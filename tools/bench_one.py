"""Run ONE conv shape repeatedly (for rocprofv3 --pmc runs).  args: ci co ho r stride up B iters kind(fwd|wgrad)"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vqgan_training_amd as vq
from vqgan_training_amd import ops
from vqgan_training_amd._lib import lib, ptr, stream_of, dtype_code, workspace
ci, co, ho, r, stride, up, B, iters = map(int, sys.argv[1:9]); kind = sys.argv[9]
prec = ops.BF16; dev = torch.device("cuda:0"); L = lib()
hi = ho // up * stride
x = torch.randn(B, hi, hi, ci, device=dev).to(prec.dtype); w = torch.randn(co, ci, r, r, device=dev) / (ci*r*r)**0.5
dy = torch.randn(B, ho, ho, co, device=dev).to(prec.dtype); pad = r // 2
d = ops._desc(B, hi, hi, ci, ho, ho, co, ci, co, r, r, stride, 1, up, pad, pad, dtype_code(x), 1, False)
wp = ops._packed(w, "fwd", co, ci, 1, d)[0]; y = torch.empty(B, ho, ho, co, device=dev, dtype=prec.dtype); st = stream_of(x)
need = L.size("vq_conv2d_wgrad_workspace", C.byref(d)); ws = workspace(dev, need); dw = torch.empty_like(w)
for _ in range(iters):
    if kind == "fwd": L.call("vq_conv2d_fwd", C.byref(d), ptr(x), ptr(wp), None, None, None, ptr(y), None, 0, st)
    else: L.call("vq_conv2d_wgrad", C.byref(d), ptr(x), ptr(dy), ptr(dw), None, 0, ptr(ws), ws.numel(), st)
torch.cuda.synchronize()
