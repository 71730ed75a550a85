#!/usr/bin/env python
"""Which stack owns the deviation of bench.py's constructor-initialised `parity` object (`ref` gave
overall_vae_loss_rel 1.4e-4, g_gan_loss_rel 8.9e-4 and 5304 flushed waves in the discriminator stack)?  The benchmark's model at
batch 2 against the CPU fp32 oracle under policy variants that move ONE stack at a time to binary16 / bf16, and under per-stack
calibration targets.  GPU only:  python tools/parity_attrib.py > gpurun_out/r4_parity_attrib.txt"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                   # noqa: E402
import vqgan_training_amd as vq                # noqa: E402

P = vq.vae_trainer.PRECISION_POLICIES
P["only_enc16"] = dict(encoder="fp16", decoder="fp32x3", lpips="fp32x3", disc="fp32x3")
P["only_dec_bf16"] = dict(encoder="fp32x3", decoder="bf16", lpips="fp32x3", disc="fp32x3")
P["only_lpips16"] = dict(encoder="fp32x3", decoder="fp32x3", lpips="fp16", disc="fp32x3")
P["only_disc16"] = dict(encoder="fp32x3", decoder="fp32x3", lpips="fp32x3", disc="fp16")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--res", type=int, default=256)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    args = bench.parse(["--cpu-baseline-res", str(a.res)])
    cfg = {"ch": 128, "ch_mult": (1, 2, 4, 4), "z": 16, "res": 256, "gan": True, "vq": None}
    _line, ref = bench.cpu_baseline(args, cfg, configs0=False)
    sds, x, first, kw = ref
    print("oracle first step:", {k: float(first[k]) for k in ("perceptual_loss", "overall_vae_loss", "vae_loss", "d_loss", "g_gan_loss")}, flush=True)
    for pol, target in (("fp32x6", 10), ("fp32x3", 10), ("ref", 10), ("ref", 12), ("ref", {"disc": 12, "encoder": 10, "lpips": 10}),
                        ("ref", {"disc": 14, "encoder": 10, "lpips": 10}), ("only_enc16", 10), ("only_dec_bf16", 10), ("only_lpips16", 10),
                        ("only_disc16", 10), ("only_disc16", 14), ("bf16", 10)):
        step, _vae = bench._hip_step_from(sds, x.shape[-1], kw, pol, dev)
        rep = step.calibrate_grad_scales(x.to(dev), target_log2=target)
        got = step(x.to(dev))
        torch.cuda.synchronize()
        dev_ = bench._deviation(got, first)
        ev = step.poll_range_events()
        print(json.dumps({"policy": pol, "target_log2": target, **dev_,
                          "scales_log2": {r["region"]: round(__import__("math").log2(r["grad_scale"]), 1) for r in rep if r.get("grad_scale", 0) > 0},
                          "events": [{k: e[k] for k in ("region", "saturated", "flushed", "fwd_saturated", "fwd_flushed")} for e in ev["stacks"]]}), flush=True)
        del step, _vae, got
        vq.ops.clear_caches()
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
