"""Does a second read hit the 256 MiB Infinity Cache?  Back-to-back `out.copy_(a)` and `a.sum()` over working sets of 8 MB .. 2 GB:
achieved GB/s per size (the premise of every "schedule the GroupNorm backward so that its second pass over x and dy is an L3 hit" plan).
GPU only:  python tools/micro/l3_probe.py > gpurun_out/r4_l3_probe.txt"""
import torch

dev = torch.device("cuda:0")
for mb in (8, 16, 32, 64, 96, 128, 192, 256, 384, 512, 1024, 2048):
    n = mb * (1 << 20) // 2
    a = torch.randn(n, device=dev, dtype=torch.float16)
    out = torch.empty_like(a)
    for name, fn, bytes_ in (("read (sum)", lambda: a.sum(), 2 * n), ("copy", lambda: out.copy_(a), 4 * n)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        s.record()
        for _ in range(reps):
            fn()
        e.record()
        torch.cuda.synchronize()
        ms = s.elapsed_time(e) / reps
        print(f"{mb:5d} MB tensor  {name:10s} {ms * 1e3:9.1f} us  {bytes_ / ms / 1e6:8.0f} GB/s", flush=True)
    del a, out
