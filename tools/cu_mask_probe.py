"""How a HIP CU mask maps onto the chip: one MFMA-bound launch (weight gradient, 128 -> 128 @256^2, B = 16) timed on streams created with
hipExtStreamCreateWithCUMask for several mask sizes (first n bits set), and on an ordinary stream.  usage: python tools/cu_mask_probe.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vqgan_training_amd as vq
from vqgan_training_amd import ops
from vqgan_training_amd._lib import lib, ptr, stream_of, dtype_code, workspace

dev = torch.device("cuda:0"); L = lib(); prec = ops._PRECISIONS["bf16"]
B, ci, co, h = 16, 128, 128, 256
x = ops.to_nhwc(torch.randn(B, ci, h, h, device=dev), prec).detach(); dy = ops.to_nhwc(torch.randn(B, co, h, h, device=dev), prec).detach()
d = ops._desc(B, h, h, ci, h, h, co, ci, co, 3, 3, 1, 1, 1, 1, 1, dtype_code(x), prec.split, False, wgrad=True)
ws = workspace(dev, L.size("vq_conv2d_wgrad_workspace", C.byref(d))); dw = torch.empty(co, ci, 3, 3, device=dev)
hip = C.CDLL("libamdhip64.so")

def masked(bits):
    words = (len(bits) + 31) // 32
    arr = (C.c_uint32 * words)(*[sum(1 << j for j in range(32) if w * 32 + j < len(bits) and bits[w * 32 + j]) for w in range(words)])
    hnd = C.c_void_p()
    assert hip.hipExtStreamCreateWithCUMask(C.byref(hnd), C.c_uint32(words), arr) == 0
    return torch.cuda.ExternalStream(hnd.value, device=0)

def run(stream, tag):
    with torch.cuda.stream(stream):
        st = stream_of(x)
        call = lambda: L.call("vq_conv2d_wgrad", C.byref(d), ptr(x), ptr(dy), ptr(dw), None, 0, ptr(ws), ws.numel(), st)
        for _ in range(3): call()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20): call()
        e.record()
    torch.cuda.synchronize()
    print(f"{tag}: {s.elapsed_time(e) / 20 * 1e3:.1f} us per launch", flush=True)

run(torch.cuda.Stream(), "ordinary stream")
for n in (256, 224, 128, 64, 32, 8):
    run(masked([1] * n), f"first {n} mask bits")
run(masked([1 if i % 2 == 0 else 0 for i in range(256)]), "every 2nd bit of 256")
run(masked([1 if i % 8 == 0 else 0 for i in range(256)]), "every 8th bit of 256 (one XCD if bits go round-robin over XCDs)")
run(masked([1 if i < 32 else 0 for i in range(256)]), "bits 0-31 of 256")
