"""Micro-benchmark of the phase-decomposed resampling convs (Upsample fwd / dgrad / wgrad, Downsample dgrad) at the
config-2 layer shapes, through ops.* (the host path the train step takes).  Knobs: VQ_TILE, VQ_WGTILE, VQ_SUBPIXEL(_WGRAD).
Usage: python tools/bench_subpix.py [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vqgan_training_amd as vq
from vqgan_training_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = torch.device("cuda:0")
L = vq._lib.lib()
ops._hint_conv = int(os.environ.get("VQ_TILE", "0"))
ops._hint_wgrad = int(os.environ.get("VQ_WGTILE", "0"))
def timeit(fn, it=8):
    fn(); fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / it
print(f"VQ_TILE={os.environ.get('VQ_TILE', '0')} VQ_WGTILE={os.environ.get('VQ_WGTILE', '0')} VQ_SUBPIXEL={os.environ.get('VQ_SUBPIXEL', '1')} "
      f"VQ_SUBPIXEL_WGRAD={os.environ.get('VQ_SUBPIXEL_WGRAD', '1')}")
for c, h in [(256, 128), (512, 64), (512, 32)]:
    x = torch.randn(B, h, h, c, device=dev).to(torch.bfloat16)
    w = torch.randn(c, c, 3, 3, device=dev) / (c * 9) ** 0.5
    b = torch.zeros(c, device=dev)
    dy = torch.randn(B, 2 * h, 2 * h, c, device=dev).to(torch.bfloat16)
    tf = timeit(lambda: ops.conv_fwd_raw(x, w, b, None, 1, 1, 1, 2, False, 1, None))
    td = timeit(lambda: ops.conv_dgrad_raw(dy, x, w, 1, 1, 1, 2, 1, False))
    tw = timeit(lambda: ops.conv_wgrad_raw(x, dy, w, b, 1, 1, 1, 2, 1))
    print(f"up   {c}->{c} in {B}x{h}x{h}: fwd {tf*1e3:7.1f} us | dgrad {td*1e3:7.1f} us | wgrad+bias {tw*1e3:7.1f} us", flush=True)
for c, h in [(128, 256), (256, 128), (512, 64)]:
    x = torch.randn(B, h, h, c, device=dev).to(torch.bfloat16)
    w = torch.randn(c, c, 3, 3, device=dev) / (c * 9) ** 0.5
    dy = torch.randn(B, h // 2, h // 2, c, device=dev).to(torch.bfloat16)
    td = timeit(lambda: ops.conv_dgrad_raw(dy, x, w, 2, 0, 0, 1, 1, False))
    print(f"down {c}->{c} in {B}x{h}x{h}: dgrad {td*1e3:7.1f} us", flush=True)
