"""Hardware check of the experimental nine-tap kernel (vq_debug_set_conv_tile(5)) against the default kernels on the same
inputs (bias, residual, ReLU, the nearest-2x gather, ragged cout tile).  Prints max |diff| / max |ref| per case."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vqgan_training_amd as vq
from vqgan_training_amd import ops
dev = torch.device("cuda:0")
L = vq._lib.lib()
ops.set_subpixel(False)
g = torch.Generator(device=dev).manual_seed(3)
bad = 0
for (n, h, w, ci, co, up, relu, res) in [(2, 16, 32, 128, 128, 1, False, True), (16, 256, 256, 128, 128, 1, False, False),
                                         (4, 64, 64, 512, 512, 1, True, False), (3, 24, 48, 64, 192, 1, False, True),
                                         (2, 16, 16, 256, 256, 2, False, False)]:
    x = torch.randn(n, h, w, ci, device=dev, generator=g).to(torch.bfloat16)
    wt = torch.randn(co, ci, 3, 3, device=dev, generator=g) / (ci * 9) ** 0.5
    b = torch.randn(co, device=dev, generator=g)
    r = torch.randn(n, h * up, w * up, co, device=dev, generator=g).to(torch.bfloat16) if res else None
    out = []
    for mode in (0, int(os.environ.get("VQ_TAP9_MODE", "5"))):
        L.dll.vq_debug_set_conv_tile(mode)
        ops.clear_caches()
        out.append(ops.conv_fwd_raw(x, wt, b, r, 1, 1, 1, up, relu, 1, None).float())
    torch.cuda.synchronize()
    err = ((out[0] - out[1]).abs().max() / out[0].abs().max()).item()
    bad += err > 2e-2
    print(f"{n}x{h}x{w} {ci}->{co} up{up} relu={relu} res={res}: rel diff {err:.2e}", flush=True)
L.dll.vq_debug_set_conv_tile(0)
print("TAP9 CHECK", "FAILED" if bad else "ok")
