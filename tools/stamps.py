"""Cycle stamps of ONE conv launch (tools only; needs the `make ablate` library): where block 0 / thread 0 spends a tile.
usage: VQ_TILE=<hint> python tools/stamps.py <fp16|bf16> <shape index of tools/bench_conv.py> [B]
ids: c64 kernel 0 tile start, 1 MFMA steps done, 2 DMA waited, 7 epilogue done, 8 barrier; nine-tap kernel 10 start, 11 first tile
landed, 12 main loop done, 13 end; inside the epilogue 3 operands requested, 4 transposition written, 5 barrier passed, 6 stores issued"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vqgan_training_amd as vq
from vqgan_training_amd import ops
from vqgan_training_amd._lib import ptr, stream_of, dtype_code
path = os.environ.get("VQ_ABLATE_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "build", "ablate", "libvqhip_ablate.so"))
lib = vq._lib.VqLibrary(path)
vq._lib._set_library_for_tests(lib)
raw = C.CDLL(path)
raw.vq_debug_stamps.restype = C.c_int
raw.vq_debug_stamps.argtypes = [C.POINTER(C.c_longlong), C.c_int]
prec = ops._PRECISIONS[sys.argv[1]]
SHAPES = {0: (128, 128, 256), 12: (64, 64, 256), 1: (256, 256, 128), 2: (512, 512, 64), 3: (512, 512, 32)}
ci, co, ho = SHAPES[int(sys.argv[2])]
B = int(sys.argv[3]) if len(sys.argv) > 3 else 16
ops._hint_conv = int(os.environ.get("VQ_TILE", "0"))
dev = torch.device("cuda:0")
x = torch.randn(B, ho, ho, ci, device=dev).to(prec.dtype)
w = torch.randn(co, ci, 3, 3, device=dev) / (ci * 9) ** 0.5
d = ops._desc(B, ho, ho, ci, ho, ho, co, ci, co, 3, 3, 1, 1, 1, 1, 1, dtype_code(x), prec.split, False)
wp, sc = ops._packed(w, "fwd", co, ci, prec.split, d, ops._op(x))
d.alpha_dev = ops._adev(sc)
y = torch.empty(B, ho, ho, co, device=dev, dtype=prec.dtype)
st = stream_of(x)
buf = (C.c_longlong * 512)()
for it in range(3):
    lib.call("vq_conv2d_fwd", C.byref(d), ptr(x), ptr(wp), None, None, None, ptr(y), None, 0, st)
    torch.cuda.synchronize()
    n = raw.vq_debug_stamps(buf, 512)
rows = [(buf[i] >> 56, buf[i] & ((1 << 56) - 1)) for i in range(n)]
print(f"{sys.argv[1]} {ci}->{co} @{ho} B={B} VQ_TILE={ops._hint_conv}: {n} stamps")
prev = None
line = []
for i, (sid, t) in enumerate(rows[:120]):
    line.append(f"{sid}:+{0 if prev is None else t - prev}")
    prev = t
    if len(line) == 12:
        print("  " + "  ".join(line)); line = []
if line:
    print("  " + "  ".join(line))
