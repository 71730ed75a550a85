import cProfile, pstats, sys, os, io
sys.path.insert(0, "/root/repo")
import torch, bench
import vqgan_training_amd as vq
dev = torch.device("cuda:0")
cfg = {"ch": 128, "ch_mult": (1, 2, 4, 4), "z": 16, "res": 256, "gan": True, "vq": None}
B = 2
step = bench.build_step(vq, cfg, dev, "ref", B)
gen = torch.Generator(device=dev).manual_seed(1)
x = vq.vae_trainer.synthetic_batch(B, 256, dev, gen)
step.calibrate_grad_scales(x)
for _ in range(3): step(x)
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(5): step(x)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"host enqueue {(t1 - t0) / 5 * 1e3:.2f} ms/step, with sync {(t2 - t0) / 5 * 1e3:.2f} ms/step")
pr = cProfile.Profile(); pr.enable()
for _ in range(5): step(x)
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28); print(s.getvalue()[:6000])
