#!/usr/bin/env python
"""Round 5: which stacks must run in the f16x3 arithmetic for north_star's 1e-4?  The headline model (configs[2], batch 2, 256x256)
on RE-RANDOMISED weights (bench.parity_randomized's set-up) and on constructor-initialised weights, first step, against the CPU fp32
oracle, under policies that mix binary16 (fp16), f16x3 and fp32x6 per stack — and the throughput of the candidates at batch 16.
GPU only:  python tools/parity_attrib_r5.py [--no-timing] > gpurun_out/r5_parity_attrib.txt"""
import json
import os
import sys
import time
import warnings

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
warnings.filterwarnings("ignore")
import bench                                   # noqa: E402
import vqgan_training_amd as vq                # noqa: E402
from oracle import model_ref as M              # noqa: E402
from oracle import weights as W                # noqa: E402

P = vq.vae_trainer.PRECISION_POLICIES
P["x_enc16"] = dict(encoder="fp16", decoder="f16x3", lpips="f16x3", disc="f16x3")            # = ref2
P["x_enc16_lp16"] = dict(encoder="fp16", decoder="f16x3", lpips="fp16", disc="f16x3")
P["x_lp16"] = dict(encoder="f16x3", decoder="f16x3", lpips="fp16", disc="f16x3")
P["x_disc_fp32x6"] = dict(encoder="f16x3", decoder="f16x3", lpips="f16x3", disc="fp32x6")
P["x_dec_fp32x6"] = dict(encoder="f16x3", decoder="fp32x6", lpips="f16x3", disc="f16x3")
P["x_disc16"] = dict(encoder="f16x3", decoder="f16x3", lpips="f16x3", disc="fp16")
P["x_dec_bf16"] = dict(encoder="f16x3", decoder="bf16", lpips="f16x3", disc="f16x3")
POLS = ["f16x3", "x_enc16", "x_enc16_lp16", "x_lp16", "x_disc_fp32x6", "x_dec_fp32x6", "x_disc16", "x_dec_bf16"]


def main():
    dev = torch.device("cuda:0")
    cfg = {"ch": 128, "ch_mult": (1, 2, 4, 4), "z": 16, "res": 256, "gan": True, "vq": None}
    res = 256
    torch.manual_seed(7)
    vae0 = vq.ae.VAE(res, 3, cfg["ch"], 3, list(cfg["ch_mult"]), 2, cfg["z"], False, False, False)
    lp0 = vq.utils.LPIPS(pretrained_path=None)
    disc0 = vq.utils.PatchDiscriminator()
    ctor = (vae0.state_dict(), lp0.state_dict(), disc0.state_dict())
    rnd = (W.randomize_state_dict(vae0.state_dict(), 1), W.randomize_state_dict(lp0.state_dict(), 2, relu_net=True),
           W.randomize_state_dict(disc0.state_dict(), 4, relu_net=True))
    kw = dict(do_ganloss=True, disc_type="hinge", learning_rate_vae=1e-5, vae_ch=cfg["ch"], max_steps=1000, warmup_steps=0)
    x = W.image_batch(2, res, seed=11)
    torch.set_num_threads(min(32, os.cpu_count() or 8))
    for tag, sds in (("randomised", rnd), ("constructor", ctor)):
        want = M.train_step_ref(M.RefState(*sds), x, **kw)
        for pol in POLS:
            g2 = {}
            step, vae = bench._hip_step_from(sds, res, kw, pol, dev, on_backward=lambda s_: g2.update(
                {n: p.grad.detach().clone() for n, p in vae.named_parameters()}) if not g2 else None)
            step.calibrate_grad_scales(x.to(dev))
            d = bench._deviation(step(x.to(dev)), want, g2)
            worst = max(v for k, v in d.items() if k.endswith("_loss_rel"))
            print(json.dumps({"weights": tag, "policy": pol, "stacks": P[pol], "worst_loss_rel": worst, **d}), flush=True)
            del step, vae, g2
            vq.ops.clear_caches()
            torch.cuda.empty_cache()
    if "--no-timing" in sys.argv:
        return
    args = bench.parse(["--no-cpu-baseline", "--no-secondary"])
    gen = torch.Generator(device=dev).manual_seed(42)
    batches = [vq.vae_trainer.synthetic_batch(16, 256, dev, gen) for _ in range(2)]
    for pol in ("f16x3", "x_enc16", "x_enc16_lp16", "x_lp16"):
        st = bench.build_step(vq, cfg, dev, pol, 16)
        bench.calibrate(st, batches[0])
        e, _ = bench.timed_run(st, batches, 6, 2, 1)
        print(json.dumps({"policy": pol, "images_per_sec": round(6 * 16 / e, 2), "ms_per_step": round(e / 6 * 1e3, 2)}), flush=True)
        del st
        vq.ops.clear_caches()
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
