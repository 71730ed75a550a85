"""GroupNorm(+SiLU) passes over a [16,H,W,C] tensor: the whole batch per launch vs chunks of samples small enough that the second
read of a two-pass op (statistics -> apply, backward reduce -> backward apply) can hit the 256 MiB Infinity Cache instead of HBM.
usage: python tools/bench_gn_chunk.py [fp16|bf16]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vqgan_training_amd as vq
from vqgan_training_amd import ops
from vqgan_training_amd._lib import lib, ptr, stream_of, dtype_code
dev = torch.device("cuda:0"); L = lib(); B, G = 16, 32
dt = torch.float16 if (sys.argv[1:] or ["fp16"])[0] == "fp16" else torch.bfloat16
def timeit(fn, it=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / it
for (c, h) in [(128, 256), (256, 128), (512, 64), (256, 256)]:
    x = torch.randn(B, h, h, c, device=dev).to(dt); dy = torch.randn_like(x); y = torch.empty_like(x); dx = torch.empty_like(x)
    gam, bet = torch.randn(c, device=dev), torch.randn(c, device=dev)
    dg, db = torch.zeros(c, device=dev), torch.zeros(c, device=dev)
    stats = torch.empty(2, B * G, device=dev); st = stream_of(x); hw = h * h; code = dtype_code(x)
    ws = ops.workspace(dev, L.size("vq_gn_workspace", B, hw, c))
    def fwd(ch):
        for i in range(0, B, ch):
            L.call("vq_gn_stats", ptr(x[i:]), ch, hw, c, G, 1e-6, code, ptr(stats[0][i * G:]), ptr(stats[1][i * G:]), ptr(ws), ws.numel(), st)
            L.call("vq_gn_silu_fwd", ptr(x[i:]), ptr(stats[0][i * G:]), ptr(stats[1][i * G:]), ptr(gam), ptr(bet), ch, hw, c, G, c, code, 1, ptr(y[i:]), st)
    def bwd(ch):
        for i in range(0, B, ch):
            L.call("vq_gn_silu_bwd", ptr(x[i:]), ptr(dy[i:]), ptr(stats[0][i * G:]), ptr(stats[1][i * G:]), ptr(gam), ptr(bet), None, ch, hw, c,
                   G, c, code, 1, ptr(dx[i:]), ptr(dg), ptr(db), 1 if i else 0, 1.0, None, 1.0, None, ptr(ws), ws.numel(), st)
    mb = x.numel() * x.element_size() / 1e6
    for ch in (16, 8, 4, 2, 1):
        tf, tb = timeit(lambda: fwd(ch)), timeit(lambda: bwd(ch))
        print(f"{c}ch @{h}x{h} ({mb:.0f} MB/tensor) chunk {ch:2d}: fwd {tf*1e3:7.1f} us ({2*mb/tf/1e3:5.2f} TB/s algorithmic)   "
              f"bwd {tb*1e3:7.1f} us ({3*mb/tb/1e3:5.2f} TB/s algorithmic)", flush=True)
