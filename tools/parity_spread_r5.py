#!/usr/bin/env python
"""Round 5: how much of the f16x3 policy's deviation on the GAN term is ARITHMETIC and how much is the luck of the summation order?
The headline-model first step (re-randomised weights, batch 2) against the CPU fp32 oracle under kernel hints that change WHICH
kernels run — i.e. the order in which the same products are added — but not the arithmetic: 0 = the library's choice, conv 6 = no
nine-tap / three-tap kernels, conv 1 = 128x128 tiles, conv 3 = 256x256 tiles wherever possible, wgrad 4 = never the three-tap
weight gradient, wgrad 64 = 64-wide one-tap tiles.  GPU only: python tools/parity_spread_r5.py > gpurun_out/r5_parity_spread.txt"""
import json
import os
import sys
import warnings

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
warnings.filterwarnings("ignore")
import bench                                   # noqa: E402
import vqgan_training_amd as vq                # noqa: E402
from vqgan_training_amd import ops             # noqa: E402
from oracle import model_ref as M              # noqa: E402
from oracle import weights as W                # noqa: E402


def main():
    dev = torch.device("cuda:0")
    cfg = {"ch": 128, "ch_mult": (1, 2, 4, 4), "z": 16, "res": 256, "gan": True, "vq": None}
    res = 256
    torch.manual_seed(7)
    vae0 = vq.ae.VAE(res, 3, cfg["ch"], 3, list(cfg["ch_mult"]), 2, cfg["z"], False, False, False)
    lp0 = vq.utils.LPIPS(pretrained_path=None)
    disc0 = vq.utils.PatchDiscriminator()
    sds = (W.randomize_state_dict(vae0.state_dict(), 1), W.randomize_state_dict(lp0.state_dict(), 2, relu_net=True),
           W.randomize_state_dict(disc0.state_dict(), 4, relu_net=True))
    kw = dict(do_ganloss=True, disc_type="hinge", learning_rate_vae=1e-5, vae_ch=cfg["ch"], max_steps=1000, warmup_steps=0)
    x = W.image_batch(2, res, seed=11)
    torch.set_num_threads(min(32, os.cpu_count() or 8))
    want = M.train_step_ref(M.RefState(*sds), x, **kw)
    for pol in ("f16x3", "fp32x6"):
        for conv, wgrad in ((0, 0), (6, 0), (1, 0), (3, 0), (0, 4), (0, 64), (6, 4)):
            if pol == "fp32x6" and (conv, wgrad) not in ((0, 0), (1, 0)):
                continue
            with ops.kernel_hints(conv=conv, wgrad=wgrad):
                ops.clear_caches()
                step, vae = bench._hip_step_from(sds, res, kw, pol, dev)
                step.calibrate_grad_scales(x.to(dev))
                d = bench._deviation(step(x.to(dev)), want)
            print(json.dumps({"policy": pol, "hint_conv": conv, "hint_wgrad": wgrad, **d}), flush=True)
            del step, vae
            ops.clear_caches()
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
