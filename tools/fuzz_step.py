"""Random-option sweep of ONE full train step (vae_trainer.py:525-708) against oracle.model_ref.train_step_ref on the host emulator:
random augmentation streams (image flip, flip / crop invariance on the latent, the pre-LPIPS flips), HR decoder on / off, the GAN
branch with both discriminator losses, LeCam, the latent clamp — the logged losses of the first step to north_star's 1e-4, target
and reconstruction tensors.  The GAN terms are judged beside the oracle's OWN spread (fp32 vs fp64, and fp32 under one ulp of input
noise): the generator's GAN loss is taken after the discriminator's first AdamW update and is discontinuous in round-off.  Test tooling.

    [FUZZ_DEVICE=cuda] python tools/fuzz_step.py [n_cases] [seed] [precision]
"""
import os
import random
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import vqgan_training_amd as vq                                      # noqa: E402
from oracle import model_ref as M                                    # noqa: E402
from oracle import weights as W                                      # noqa: E402


def rel(a, b):
    a, b = torch.as_tensor(a).detach().double().cpu(), torch.as_tensor(b).detach().double().cpu()
    # (scalars: against at least 1e-3 — a saturated BCE generator term is exp(-logit) ~ 1e-16, where a relative error of the VALUE is
    # the absolute error of a logit of 36)
    return ((a - b).norm() / b.norm().clamp_min(1e-3 if b.numel() == 1 else 1e-30)).item()


def check_step(i, opts, prec, device="cpu"):
    res, ch = 32, 32
    hr = opts["decoder_also_perform_hr"]
    vq.ops.clear_caches()
    vq.ops.set_default_precision(prec)
    vae = vq.ae.VAE(res, 3, ch, 3, [1, 2], 1, 4, False, hr, False)
    vae.load_state_dict(W.randomize_state_dict(vae.state_dict(), 1 + i))
    lp = vq.utils.LPIPS(pretrained_path=None)
    lp.load_state_dict(W.randomize_state_dict(lp.state_dict(), 2, relu_net=True))
    disc = None
    if opts["do_ganloss"]:
        disc = vq.utils.PatchDiscriminator()
        disc.load_state_dict(W.randomize_state_dict(disc.state_dict(), 4 + i, relu_net=True))
    st = M.RefState(vae.state_dict(), lp.state_dict(), None if disc is None else disc.state_dict())
    st64 = M.RefState(vae.state_dict(), lp.state_dict(), None if disc is None else disc.state_dict(), dtype=torch.float64)
    start = tuple({k: v.detach().clone() for k, v in m.state_dict().items()} if m is not None else None for m in (vae, lp, disc))
    vae, lp = vae.to(device), lp.to(device).eval()
    disc = None if disc is None else disc.to(device)
    kw = dict(opts, downscale_factor=2, enc_size=(res, res), learning_rate_vae=1e-3, vae_ch=ch, max_steps=10, warmup_steps=1)
    seed = kw.pop("rng_seed")
    step = vq.vae_trainer.VAETrainStep(vae, lp, disc, rng=random.Random(seed), **kw)
    x = W.image_batch(2, 2 * res if hr else res, seed=8 + i)
    o = step(x.to(device))
    r = M.train_step_ref(st, x, rng=random.Random(seed), **kw)
    # the yardstick for the GAN terms: the same restated step in fp64.  The generator's GAN loss is taken AFTER the discriminator's first
    # AdamW update — sign-like steps of +-lr on every parameter, also on those whose gradient is round-off — so two correct fp32
    # evaluations differ there by more than their arithmetic (tests/test_model.py).
    r64 = M.train_step_ref(st64, x.double(), rng=random.Random(seed), **kw)
    errs = {"target": rel(o["target"], r["target"]), "reconstructed": rel(o["reconstructed"], r["reconstructed"])}
    for k in ("overall_vae_loss", "perceptual_loss", "vae_loss") + (("d_loss", "g_gan_loss") if opts["do_ganloss"] else ()):
        errs[k] = rel(o[k], r[k])
    tol = {"target": 1e-6, "reconstructed": 5e-4}
    yard = {k: rel(r[k], r64[k]) for k in errs if k in ("g_gan_loss", "overall_vae_loss", "d_loss")}
    if opts["do_ganloss"]:
        # ... and its conditioning, measured on the oracle itself: the same fp32 step on the batch times (1 + one ulp of noise), three
        # draws.  d_loss moves by ~1e-6; the generator's GAN term jumps by up to 7e-4 in one case out of ten (a handful of discriminator
        # weights whose gradient is ~1e-8 — Adam's eps — step by +lr instead of -lr), so that spread is part of the yardstick.
        for t in range(3):
            stp = M.RefState(start[0], start[1], start[2])
            xp = x * (1 + 2e-7 * torch.randn(x.shape, generator=torch.Generator().manual_seed(t + 1)))
            rp = M.train_step_ref(stp, xp, rng=random.Random(seed), **kw)
            for k in yard:
                yard[k] = max(yard[k], rel(rp[k], r[k]))
    bound = lambda k: max(tol.get(k, 1e-4), 3 * yard.get(k, 0.0))                     # noqa: E731
    worst = max(errs, key=lambda k: errs[k] / bound(k))
    return errs[worst] < bound(worst), " ".join(f"{k} {v:.1e}" + (f" (the oracle's own spread: {yard[k]:.1e})" if k in yard and v > 1e-4 else "") for k, v in errs.items())


def main():
    import tests.conftest as cf
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    prec = sys.argv[3] if len(sys.argv) > 3 else "fp32x6"
    device = os.environ.get("FUZZ_DEVICE", "cpu")       # "cuda": the product library on the GPU instead of the host emulator
    if device == "cpu":
        cf._build("emu", cf.EMU_LIB)
        vq._lib._set_library_for_tests(vq._lib.VqLibrary(cf.EMU_LIB))
    bad = 0
    for i in range(n):
        gan = rnd.random() < 0.5
        opts = dict(flip_invariance=rnd.random() < 0.6, crop_invariance=rnd.random() < 0.6,
                    augment_before_perceptual_loss=rnd.random() < 0.6, decoder_also_perform_hr=rnd.random() < 0.5,
                    do_ganloss=gan, disc_type=rnd.choice(["hinge", "bce"]), use_lecam=gan and rnd.random() < 0.5,
                    do_clamp=rnd.random() < 0.3, clamp_th=rnd.choice([8.0, 0.5]), rng_seed=rnd.randint(0, 10 ** 6))
        if os.environ.get("FUZZ_ONLY") and str(i) not in os.environ["FUZZ_ONLY"].split(","):
            continue
        try:
            ok, msg = check_step(i, opts, prec, device)
        except Exception as e:           # noqa: BLE001
            import traceback
            ok, msg = False, repr(e)[:300] + "\n" + "".join(traceback.format_tb(e.__traceback__)[-3:])
        bad += 0 if ok else 1
        print("ok  " if ok else "FAIL", i, {k: v for k, v in opts.items() if v not in (False,)}, msg, flush=True)
    print(f"{n - bad} / {n} ok")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
