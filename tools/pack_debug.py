"""Which weights are still packed one launch at a time inside a steady-state config-3 step, and why (ops.pack_stats).
usage (GPU): python tools/pack_debug.py"""
import collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import vqgan_training_amd as vq
from vqgan_training_amd import ops
dev = torch.device("cuda:0")
cfg = {"ch": 128, "ch_mult": (1, 2, 4, 4), "z": 16, "res": 256, "gan": True, "vq": None}
step = bench.build_step(vq, cfg, dev, "ref", 16)
x = vq.vae_trainer.synthetic_batch(16, 256, dev, torch.Generator(device=dev).manual_seed(42))
bench.calibrate(step, x)
for _ in range(3): step(x)
ops.pack_stats = collections.Counter()
step(x); torch.cuda.synchronize()
for k, v in sorted(ops.pack_stats.items(), key=lambda kv: str(kv[0])): print(v, k)
print("single-weight pack launches in one steady-state step:", sum(ops.pack_stats.values()))
