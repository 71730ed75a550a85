"""AttnBlock attention kernels at the bottleneck shape (B=16, 32x32 tokens, C=512): time per call."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vqgan_training_amd as vq
dev = torch.device("cuda:0")
for dt in (torch.bfloat16,):
    qkv = torch.randn(16, 32, 32, 1536, device=dev).to(dt).requires_grad_()
    g = torch.randn(16, 32, 32, 512, device=dev).to(dt)
    def run():
        o = vq.ops.attention(qkv); o.backward(g); qkv.grad = None
    run(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5): run()
    e.record(); torch.cuda.synchronize()
    print(f"attention fwd+bwd B=16 T=1024 C=512 {dt}: {s.elapsed_time(e)/5:.3f} ms", flush=True)
