"""Vendor-library yardstick (hipBLASLt through torch.matmul, bf16) at the implicit-GEMM shapes of the conv layers:
what a tuned dense GEMM reaches on this box, to judge the conv kernels against.  Not part of the product path."""
import torch
dev = torch.device("cuda:0")
def t(fn, it=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / it
for (M, N, K) in [(65536, 512, 4608), (16384, 512, 4608), (262144, 256, 2304), (1048576, 128, 1152), (262144, 512, 4608), (8192, 8192, 8192)]:
    a = torch.randn(M, K, device=dev, dtype=torch.bfloat16); b = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
    ms = t(lambda: torch.matmul(a, b.t()))
    print(f"M={M} N={N} K={K}: {ms:.3f} ms  {2.0*M*N*K/ms/1e9:.0f} TF", flush=True)
