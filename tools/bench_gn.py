"""Micro-benchmark of the GroupNorm+SiLU kernels at the config-2 layer shapes: effective HBM GB/s per pass.
Usage: python tools/bench_gn.py [B] [shape index: only that shape (for per-kernel rocprofv3 statistics of one shape)]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vqgan_training_amd as vq
from vqgan_training_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = torch.device("cuda:0")
def timeit(fn, it=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / it
SHAPES = [(256, 128), (128, 256), (64, 512), (32, 512), (128, 128), (256, 256)]
if len(sys.argv) > 2:
    SHAPES = [SHAPES[int(sys.argv[2])]]
for (h, c) in SHAPES:
    x = torch.randn(B, h, h, c, device=dev).to(torch.bfloat16); dy = torch.randn_like(x); add = torch.randn_like(x)
    g = torch.ones(c, device=dev); b = torch.zeros(c, device=dev)
    y, stats = ops.gn_fwd_raw(x, g, b, 32, 1e-6, True)
    tf = timeit(lambda: ops.gn_fwd_raw(x, g, b, 32, 1e-6, True))
    tb = timeit(lambda: ops.gn_bwd_raw(x, dy, stats, g, b, 32, True, add=None))
    tba = timeit(lambda: ops.gn_bwd_raw(x, dy, stats, g, b, 32, True, add=add))
    nbytes = x.numel() * 2
    print(f"B={B} {h}x{h}x{c}: tensor {nbytes/1e6:7.1f} MB | fwd {tf*1e3:7.1f} us = {3*nbytes/tf/1e6:6.0f} GB/s (3 passes) | "
          f"bwd {tb*1e3:7.1f} us = {5*nbytes/tb/1e6:6.0f} GB/s (5 passes) | bwd+add {tba*1e3:7.1f} us = {6*nbytes/tba/1e6:6.0f} GB/s", flush=True)
