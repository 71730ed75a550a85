"""Round-6 A/B: what does a collective's CU footprint cost the step on ONE GPU?

No 8-GPU node has been available in any round, so the DP gradient exchange (387 MB per step: 326.6 MB VAE + 60 MB discriminator,
SURVEY section 8(e)) has never shared the chip with the backward chain on hardware.  This tool runs the bench workload (configs[2],
B = 16, policy `ref`) with a THIRD stream on which, for every step, tools/micro/comm_shadow.hip keeps `blocks` workgroups resident
streaming `bytes` through the CUs while the step's two own streams (chain, weight gradients) run — unpaced (as fast as HBM lets it:
the worst case for the chain) and paced to the ~9 ms a link-bound xGMI ring needs.  The shadow is launched where
the bucketed reducer would put its first bucket on the wire: right before the generator backward (and once before the discriminator
backward for its 60 MB).  Alternating off / on legs on the same box; prints ms/step per leg.

    python tools/comm_shadow.py [--steps 10] [--blocks 16,32,64]
"""
import argparse
import ctypes
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import vqgan_training_amd as vq  # noqa: E402
from vqgan_training_amd import ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--blocks", type=str, default="16,32,64")
    ap.add_argument("--bytes_g", type=int, default=326_600_000)
    ap.add_argument("--bytes_d", type=int, default=60_050_000)
    ap.add_argument("--pace_ms", type=float, default=9.0)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = ctypes.CDLL(os.path.join(ROOT, "build", "tools", "libcomm_shadow.so"))
    lib.comm_shadow_launch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int64, ctypes.c_void_p]
    lib.comm_shadow_launch.restype = ctypes.c_int
    cfg = {"ch": 128, "ch_mult": (1, 2, 4, 4), "z": 16, "res": 256, "gan": True, "vq": None}
    B = 16
    step = bench.build_step(vq, cfg, dev, "ref", B)
    gen = torch.Generator(device=dev).manual_seed(42)
    batches = [vq.vae_trainer.synthetic_batch(B, 256, dev, gen) for _ in range(4)]
    bench.calibrate(step, batches[0])
    src = torch.zeros(a.bytes_g // 4, dtype=torch.float32, device=dev)
    dst = torch.empty_like(src)
    shadow = torch.cuda.Stream(device=dev)
    mode = {"blocks": 0, "pace_ns": 0}
    n_launch = [0]

    def launch(nbytes):
        if mode["blocks"] <= 0:
            return
        shadow.wait_stream(torch.cuda.current_stream(dev))       # like a bucket: after the gradients written so far
        rc = lib.comm_shadow_launch(src.data_ptr(), dst.data_ptr(), int(nbytes) // 16 * 16, mode["blocks"],
                                    int(mode["pace_ns"] * nbytes / a.bytes_g), shadow.cuda_stream)
        assert rc == 0, rc
        n_launch[0] += 1

    # the reducer's own place: right before each backward (the collectives then fly under it), joined before the optimizer step
    orig_backward = torch.Tensor.backward
    calls = [0]

    def backward(self, *args, **kw):
        calls[0] += 1
        launch(a.bytes_d if calls[0] % 2 == 1 else a.bytes_g)      # (GAN step: D backward first, then the generator's)
        return orig_backward(self, *args, **kw)

    torch.Tensor.backward = backward
    orig_finish = step._finish

    def finish(reducer):
        torch.cuda.current_stream(dev).wait_stream(shadow)          # reducer.finish(): the optimizer waits for the exchange
        return orig_finish(reducer)

    step._finish = finish

    def leg(blocks, pace_ms):
        mode["blocks"], mode["pace_ns"] = blocks, int(pace_ms * 1e6)
        calls[0] = 0
        for i in range(a.warmup):
            step(batches[i % 4])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(a.steps):
            step(batches[i % 4])
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / a.steps * 1e3

    rows = []
    for rep in range(2):
        rows.append(("off", 0, 0.0, leg(0, 0.0)))
        for blocks in [int(b) for b in a.blocks.split(",")]:
            rows.append(("unpaced", blocks, 0.0, leg(blocks, 0.0)))
            rows.append(("paced", blocks, a.pace_ms, leg(blocks, a.pace_ms)))
    print(f"comm shadow A/B: configs[2] B=16 policy ref, {a.steps} steps per leg, shadow = {a.bytes_d / 1e6:.0f} MB before the D backward + "
          f"{a.bytes_g / 1e6:.0f} MB before the G backward on a third stream ({n_launch[0]} shadow launches in all)")
    base = min(r[3] for r in rows if r[0] == "off")
    for kind, blocks, pace, ms in rows:
        print(f"  {kind:8s} blocks={blocks:3d} pace_ms={pace:4.1f}   {ms:7.2f} ms/step   {ms / base - 1:+.1%} vs the best shadow-less leg")


if __name__ == "__main__":
    main()
