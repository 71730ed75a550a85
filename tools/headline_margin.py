"""The headline-model parity gate (tests/test_model.py::_headline_case: configs[2]'s model, one full GAN iteration on re-randomised
weights, policy f16x3 against the fp32 oracle) on MORE seeded batches than the suite runs, each beside the oracle's own conditioning:
the same fp32 oracle step on the batch times (1 + one ulp of noise).  GPU + the box's host cores; test tooling.

    [HEADLINE_PHOTOS=1] [HEADLINE_POLICY=ref] python tools/headline_margin.py [first_seed] [n_batches] [noise_draws]
"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))     # tests/test_model.py imports its fixtures package as `golden`
import tests.test_model as tm                                        # noqa: E402
import vqgan_training_amd as vq                                      # noqa: E402
from oracle import model_ref as M                                    # noqa: E402
from oracle import weights as W                                      # noqa: E402


def main():
    s0 = int(sys.argv[1]) if len(sys.argv) > 1 else 14
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    draws = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    keys = ("perceptual_loss", "overall_vae_loss", "vae_loss", "d_loss", "g_gan_loss")
    res, ch, mult = 256, 128, [1, 2, 4, 4]
    worst = []
    policy = os.environ.get("HEADLINE_POLICY", "f16x3")      # "ref": the timed policy, beside the reference's own GPU arithmetic emulated by the oracle
    photos = os.environ.get("HEADLINE_PHOTOS", "0") == "1"     # the reference's photographs (all six pairs) through biased weights instead
    pairs = [(0, 2), (0, 3), (1, 2), (1, 3), (2, 3), (1, 0)]
    for seed in (range(len(pairs)) if photos else range(s0, s0 + n)):
        if photos:
            case, bs = f"photo{pairs[seed][0]}{pairs[seed][1]}", (W.PHOTO_BIAS_SCALE, W.PHOTO_VGG_BIAS_SCALE)
            make_x = (lambda pr=pairs[seed]: W.photo_batch(list(pr), 256))
        else:
            case, bs = f"noise{seed}", (1.0, 1.0)
            make_x = (lambda s=seed: W.image_batch(2, 256, seed=s))
        tm.HEADLINE_CASES[case] = (bs, make_x)
        t0 = time.time()
        meas = tm._headline_case(policy, case)
        # conditioning of the oracle itself
        torch.manual_seed(7)
        vae = vq.ae.VAE(res, 3, ch, 3, list(mult), 2, 16, False, False, False)
        vae.load_state_dict(W.randomize_state_dict(vae.state_dict(), 1, bias_scale=bs[0]))
        lp = vq.utils.LPIPS(pretrained_path=None)
        lp.load_state_dict(W.randomize_state_dict(lp.state_dict(), 2, relu_net=True, bias_scale=bs[1]))
        disc = vq.utils.PatchDiscriminator()
        disc.load_state_dict(W.randomize_state_dict(disc.state_dict(), 4, relu_net=True, bias_scale=bs[1]))
        sds = (vae.state_dict(), lp.state_dict(), disc.state_dict())
        kw = dict(do_ganloss=True, disc_type="hinge", learning_rate_vae=1e-5, vae_ch=ch, max_steps=1000, warmup_steps=0)
        want = tm._ORACLE_CACHE[case][0]
        x = make_x()
        spread = {k: 0.0 for k in keys}
        for t in range(draws):
            xp = x * (1 + 2e-7 * torch.randn(x.shape, generator=torch.Generator().manual_seed(100 + t)))
            rp = M.train_step_ref(M.RefState(*sds), xp, **kw)
            for k in keys:
                spread[k] = max(spread[k], tm.rel(rp[k], want[k]))
        emu = ""
        if policy == "ref":        # what the reference's CUDA path computes in (TF32 encoder / LPIPS / discriminator, bf16 decoder), emulated
            re_ = M.train_step_ref(M.RefState(*sds), x, arith=M.REFERENCE_GPU_ARITH, **kw)
            emu = "  |  the reference's GPU arithmetic (emulated) vs the fp32 oracle  " + " ".join(f"{k}={tm.rel(re_[k], want[k]):.2e}" for k in keys)
        wk = max(keys, key=lambda k: meas[k])
        worst.append(meas[wk])
        print(f"batch {case}: {policy} vs fp32 oracle  " + " ".join(f"{k}={meas[k]:.2e}" for k in keys) + f" recon={meas['recon']:.2e}"
              + "  |  fp32 oracle under one ulp of input noise  " + " ".join(f"{k}={spread[k]:.2e}" for k in keys) + emu + f"   [{time.time() - t0:.0f} s]", flush=True)
    worst.sort()
    n = len(worst)
    print(f"worst logged-loss deviation over {n} batches: max {worst[-1]:.2e}, median {worst[len(worst) // 2]:.2e}; "
          f"{sum(w > 1e-4 for w in worst)} of {n} beyond 1e-4")


if __name__ == "__main__":
    main()
