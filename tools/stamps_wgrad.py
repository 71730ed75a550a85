"""Cycle stamps of ONE three-tap weight-gradient launch (tools only; `make ablate` library): block 0 / thread 0.
usage: python tools/stamps_wgrad.py <fp16|bf16> <Cin> <Cout> <H> [B]     ids: 20 start, 21 first chunk staged, 22 chunk loop done, 23 slab stored"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vqgan_training_amd as vq
from vqgan_training_amd import ops
from vqgan_training_amd._lib import ptr, stream_of, dtype_code, workspace
path = os.environ.get("VQ_ABLATE_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "build", "ablate", "libvqhip_ablate.so"))
lib = vq._lib.VqLibrary(path)
vq._lib._set_library_for_tests(lib)
raw = C.CDLL(path)
raw.vq_debug_stamps_wgrad.restype = C.c_int
raw.vq_debug_stamps_wgrad.argtypes = [C.POINTER(C.c_longlong), C.c_int]
prec = ops._PRECISIONS[sys.argv[1]]
ci, co, ho = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
B = int(sys.argv[5]) if len(sys.argv) > 5 else 16
dev = torch.device("cuda:0")
x = torch.randn(B, ho, ho, ci, device=dev).to(prec.dtype)
dy = torch.randn(B, ho, ho, co, device=dev).to(prec.dtype)
dw = torch.empty(co, ci, 3, 3, device=dev)
d = ops._desc(B, ho, ho, ci, ho, ho, co, ci, co, 3, 3, 1, 1, 1, 1, 1, dtype_code(x), prec.split, False, wgrad=True)
need = lib.size("vq_conv2d_wgrad_workspace", C.byref(d))
ws = workspace(dev, need)
st = stream_of(x)
buf = (C.c_longlong * 64)()
for it in range(3):
    lib.call("vq_conv2d_wgrad", C.byref(d), ptr(x), ptr(dy), ptr(dw), None, 0, ptr(ws), ws.numel(), st)
    torch.cuda.synchronize()
    n = raw.vq_debug_stamps_wgrad(buf, 64)
rows = [(buf[i] >> 56, buf[i] & ((1 << 56) - 1)) for i in range(n)]
M = B * ho * ho
line = "  ".join(f"{sid}:+{0 if i == 0 else t - rows[i - 1][1]}" for i, (sid, t) in enumerate(rows))
print(f"{sys.argv[1]} wgrad {ci}->{co} @{ho} B={B} (M = {M}): {line}")
