"""Weight gradient at crop-invariance sizes (rows that are multiples of 16 but not powers of two): three-tap kernel vs the
register-staged fallback (VQ_WGTILE=4)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vqgan_training_amd as vq
from vqgan_training_amd import ops
from vqgan_training_amd._lib import lib, ptr, stream_of, dtype_code, workspace
dev = torch.device("cuda:0"); L = lib(); B = 16
ops._hint_wgrad = int(os.environ.get("VQ_WGTILE", "0"))
for (c, h, w) in [(128, 208, 272), (256, 104, 136), (512, 52, 68)]:
    x = torch.randn(B, h, w, c, device=dev).to(torch.bfloat16); dy = torch.randn_like(x)
    wt = torch.randn(c, c, 3, 3, device=dev)
    d = ops._desc(B, h, w, c, h, w, c, c, c, 3, 3, 1, 1, 1, 1, 1, dtype_code(x), 1, False)
    ws = workspace(dev, L.size("vq_conv2d_wgrad_workspace", C.byref(d))); dw = torch.empty_like(wt); st = stream_of(x)
    fn = lambda: L.call("vq_conv2d_wgrad", C.byref(d), ptr(x), ptr(dy), ptr(dw), None, 0, ptr(ws), ws.numel(), st)
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5): fn()
    e.record(); torch.cuda.synchronize(); t = s.elapsed_time(e) / 5
    print(f"wgrad {c}->{c} @{h}x{w}: {t:.3f} ms {2.0*B*h*w*c*c*9/t/1e9:.0f} TF", flush=True)
