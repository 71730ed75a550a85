"""3-D TVAE (reference tae.py __main__ config: ch=64, ch_mult 1,2,4,4, 2 res blocks, z=16) forward + backward on one GPU.
Usage: python tools/bench_tvae.py [frames=16] [res=256] [iters=3]      (the reference's smoke input is 48 x 256 x 256)
Prints ms per fwd+bwd, the conv launches' aggregate TFLOP/s (HIP events on the launch stream) and peak memory."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vqgan_training_amd as vq
from vqgan_training_amd import ops

T = int(sys.argv[1]) if len(sys.argv) > 1 else 16
R = int(sys.argv[2]) if len(sys.argv) > 2 else 256
IT = int(sys.argv[3]) if len(sys.argv) > 3 else 3
dev = torch.device("cuda:0")
ops.set_default_precision("bf16")
torch.manual_seed(0)
vae = vq.tae.TVAE(resolution=R, in_channels=3, ch=64, out_ch=3, ch_mult=[1, 2, 4, 4], num_res_blocks=2, z_channels=16).to(dev)
x = torch.rand(1, 3, T, R, R, device=dev) * 2 - 1
rec = []


def hook(kind, flops, fn, tag=""):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); fn(); e.record()
    rec.append((kind, flops, s, e))


def step():
    for p in vae.parameters():
        p.grad = None
    recon, z = vae(x)
    ((recon - x).square().mean() + 1e-3 * z.square().mean()).backward()


step(); torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(IT):
    step()
e.record(); torch.cuda.synchronize()
ms = s.elapsed_time(e) / IT
ops.set_launch_hook(hook)
step(); torch.cuda.synchronize()
ops.set_launch_hook(None)
agg = {}
for kind, fl, a, b in rec:
    d = agg.setdefault(kind, [0, 0.0, 0.0]); d[0] += 1; d[1] += fl; d[2] += a.elapsed_time(b)
tot_fl = sum(v[1] for v in agg.values())
print(f"TVAE ch=64 1,2,4,4 input 1x3x{T}x{R}x{R} bf16: {ms:.1f} ms per fwd+bwd, {tot_fl / 1e12:.2f} TFLOP of conv work "
      f"-> {tot_fl / ms / 1e9:.0f} TFLOP/s end to end; peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB", flush=True)
for kind, (n, fl, t) in agg.items():
    print(f"  {kind}: {n} launches, {t:.1f} ms, {fl / t / 1e9:.0f} TFLOP/s", flush=True)
