"""What two HIP events around ONE launch add to the kernel's own duration (bench.py's roofline timing brackets every conv launch
that way): (a) two events with nothing between them, (b) events around a conv launch of ~100 / ~300 us, against the same launches timed
back to back in one bracket (per-launch average), on the same stream.  usage: python tools/event_overhead.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vqgan_training_amd as vq
from vqgan_training_amd import ops
from vqgan_training_amd._lib import lib, ptr, stream_of, dtype_code
dev = torch.device("cuda:0"); L = lib(); prec = ops._PRECISIONS["fp16"]
def ev(): return torch.cuda.Event(enable_timing=True)
pairs = [(ev(), ev()) for _ in range(200)]
torch.cuda.synchronize()
for s, e in pairs: s.record(); e.record()
torch.cuda.synchronize()
ts = sorted(s.elapsed_time(e) * 1e3 for s, e in pairs)
print(f"empty bracket: median {ts[100]:.2f} us, min {ts[0]:.2f}, p90 {ts[180]:.2f}")
for (B, c, h) in ((16, 128, 128), (16, 128, 256), (16, 512, 64)):
    x = ops.to_nhwc(torch.randn(B, c, h, h, device=dev), prec).detach(); w = torch.randn(c, c, 3, 3, device=dev) / (c * 9) ** 0.5
    d = ops._desc(B, h, h, c, h, h, c, c, c, 3, 3, 1, 1, 1, 1, 1, dtype_code(x), prec.split, False)
    wp, sc = ops._packed(w, "fwd", c, c, prec.split, d, ops._op(x)); d.alpha_dev = ops._adev(sc)
    y = torch.empty(B, h, h, c, device=dev, dtype=prec.dtype); st = stream_of(x)
    call = lambda: L.call("vq_conv2d_fwd", C.byref(d), ptr(x), ptr(wp), None, None, None, ptr(y), None, 0, st)
    for _ in range(5): call()
    torch.cuda.synchronize()
    s, e = ev(), ev(); s.record()
    for _ in range(100): call()
    e.record(); torch.cuda.synchronize()
    b2b = s.elapsed_time(e) * 10.0
    pairs = [(ev(), ev()) for _ in range(100)]
    for s, e in pairs: s.record(); call(); e.record()
    torch.cuda.synchronize()
    per = sorted(s.elapsed_time(e) * 1e3 for s, e in pairs)
    print(f"{c} ch @{h}^2: back to back {b2b:.1f} us per launch; bracketed one by one: median {per[50]:.1f} us, mean {sum(per) / 100:.1f} us  (+{per[50] - b2b:.1f} us)")
