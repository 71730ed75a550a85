"""What would folding the GroupNorm BACKWARD sums into the data-gradient conv cost?  (`make ablate` library, pricing hint 8197 << 4)
The conv that produces dy for a GroupNorm runs (a) plain, (b) with the fused-sum arithmetic in its epilogue: the GroupNorm input read as
a residual-like operand, per element the normalised value, the affine output, silu'(.), four running sums, and the sums written through
the GroupNorm-partial rows.  The extra time per launch is what the fusion costs; what it saves is the reduction pass of
vq_gn_silu_bwd on the same tensor (tools/bench_gn.py under rocprofv3: gn_reduce_kernel<DT, 1, 1>)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vqgan_training_amd as vq
from vqgan_training_amd import ops
from vqgan_training_amd._lib import ptr, stream_of, dtype_code
path = os.environ.get("VQ_ABLATE_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "build", "ablate", "libvqhip_ablate.so"))
L = vq._lib.VqLibrary(path)
vq._lib._set_library_for_tests(L)
dev = torch.device("cuda:0"); B = 16
def timeit(fn, it=20):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / it
for prec_name in ("bf16", "fp16"):
    prec = ops._PRECISIONS[prec_name]
    for (c, h) in [(128, 256), (256, 128), (512, 64)]:
        x = torch.randn(B, h, h, c, device=dev).to(prec.dtype); w = torch.randn(c, c, 3, 3, device=dev) / (c * 9) ** 0.5
        bias = torch.randn(c, device=dev) * 0.1; xg = torch.randn_like(x); y = torch.empty_like(x)
        res = {}
        for name, hint in (("plain", 0), ("fused sums", 8197 << 4)):
            ops._hint_conv = hint
            d = ops._desc(B, h, h, c, h, h, c, c, c, 3, 3, 1, 1, 1, 1, 1, dtype_code(x), prec.split, False)
            wp, sc = ops._packed(w, "fwd", c, c, prec.split, d, ops._op(x))
            d.alpha_dev = ops._adev(sc)
            st = stream_of(x)
            row = L.dll.vq_conv2d_gn_tile(C.byref(d), 32)
            part = torch.empty(B, h * h // max(row, 1), 32, 2, device=dev)
            if hint:
                fn = lambda: L.call("vq_conv2d_fwd", C.byref(d), ptr(x), ptr(wp), ptr(bias), ptr(xg), None, ptr(y), ptr(part), 32, st)
            else:
                fn = lambda: L.call("vq_conv2d_fwd", C.byref(d), ptr(x), ptr(wp), None, None, None, ptr(y), None, 0, st)
            res[name] = timeit(fn)
        print(f"{prec_name} {c}->{c} @{h} B={B}: plain {res['plain'] * 1e3:7.1f} us | with the fused GroupNorm-backward sums {res['fused sums'] * 1e3:7.1f} us "
              f"| extra {1e3 * (res['fused sums'] - res['plain']):6.1f} us per launch", flush=True)
