"""Nearest-code search alone (vq_vq_nearest_fwd) at configs[4]'s per-GPU size: 8192 tokens x 16384 codes x 32 dims.
usage: python tools/bench_vq.py [tokens] [codes] [dim]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vqgan_training_amd as vq
from vqgan_training_amd._lib import lib, ptr, stream_of, workspace
n, k, d = (int(a) for a in (sys.argv[1:4] + ["8192", "16384", "32"][len(sys.argv) - 1:]))
dev = torch.device("cuda:0"); L = lib()
g = torch.Generator(device=dev).manual_seed(1)
z = torch.randn(n, d, device=dev, generator=g); cb = torch.randn(k, d, device=dev, generator=g) * 0.5
if os.environ.get("VQ_ZERO", "0") not in ("", "0"):      # all-zero operands: the clock the part holds when nothing toggles (DVFS share)
    z.zero_(); cb.zero_()
ws = workspace(dev, L.size("vq_vq_workspace", n, k))
idx = torch.empty(n, dtype=torch.int64, device=dev); zq = torch.empty_like(z); md = torch.empty(n, device=dev)
call = lambda: L.call("vq_vq_nearest_fwd", ptr(z), ptr(cb), n, k, d, ptr(idx), ptr(zq), ptr(md), ptr(ws), ws.numel(), stream_of(z))
for _ in range(3): call()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(50): call()
e.record(); torch.cuda.synchronize()
us = s.elapsed_time(e) * 1e3 / 50
tf = 2.0 * n * k * d / (us * 1e-6) / 1e12
ref = ((z * z).sum(1, keepdim=True) - 2 * z @ cb.t() + (cb * cb).sum(1)).argmin(1)
print(f"vq_nearest {n} x {k} x {d}: {us:.1f} us per lookup (search + finalize) = {tf:.1f} TFLOP/s = {tf / 157.3:.3f} of the fp32 peak; "
      f"indices equal to a torch fp32 argmin on {(idx == ref).float().mean().item() * 100:.2f} % of the tokens")
