"""Random-configuration sweep of the VAE (ae.py: widths, level multipliers, blocks per level, latent channels, attention, the HR
decoder level, the wavelet front-end, image sizes that are not squares or powers of two) against the oracle's restatement:
reconstruction, latent and EVERY parameter gradient.  Test tooling (tests/test_model.py runs a fixed handful of these
configurations through `check_config`).

    [FUZZ_DEVICE=cuda] python tools/fuzz_model.py [n_cases] [seed] [precision]        # on the host emulator build

The yardstick is the SAME restatement in fp64 (GroupNorm included): tiny images leave GroupNorm groups of 2-4 elements whose
1/sqrt(var + eps) amplifies any fp32 rounding, so a tensor fails only beyond max(tol, 10 x the fp32 oracle's own distance to fp64) — that distance is ONE draw of fp32 rounding noise,
dominated in a 32-element tensor by its one or two worst-conditioned groups: two correct fp32 evaluations differ by such factors (the
fp32x6 sweep's outliers sat at 4x and 8x, in GroupNorm groups of three elements).
A per-channel bias in front of a GroupNorm whose groups are single channels (width 32) has an analytically ZERO gradient — both sides
hold rounding noise there — so every tensor's error is measured against at least 1e-3 of the model's typical per-element gradient.
Binary16-range storage (fp16 / f16x3) gets range-event counters as in the train step: a configuration whose gradient stores clipped
is reported as such (the train step drops that update and lowers the loss scale), not compared.
"""
import os
import random
import sys
import traceback

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import vqgan_training_amd as vq                                      # noqa: E402
from oracle import model_ref as M                                    # noqa: E402
from oracle import ops_ref as R                                      # noqa: E402
from oracle import weights as W                                      # noqa: E402

TOL = {"fp32x3": 5e-4, "f16x3": 5e-4, "fp32x6": 5e-4}


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def random_config(rnd):
    ch = rnd.choice([32, 32, 64, 96])
    mult = rnd.choice([[1], [1, 2], [1, 1], [2, 1], [1, 2, 2], [1, 2, 1]])
    nrb = rnd.choice([1, 1, 2, 3])
    zc = rnd.choice([2, 4, 8, 16])
    attn = rnd.random() < 0.3
    hr = rnd.random() < 0.3
    wav = rnd.random() < 0.3
    f = 2 ** (len(mult) - 1) * (2 if wav else 1)
    H, Wd = f * rnd.randint(1, 4), f * rnd.randint(1, 5)
    return (max(H, Wd), 3, ch, 3, mult, nrb, zc, attn, hr, wav), (rnd.choice([1, 2]), 3, H, Wd)


def check_config(cfg, xshape, prec, seed, device="cpu"):
    """-> (ok, message).  cfg = the VAE constructor's arguments (ae.py:356-386), xshape = the image batch."""
    tol = TOL.get(prec, 8e-2)
    vq.ops.clear_caches()
    vq.ops.set_default_precision(prec)
    vae = vq.ae.VAE(*cfg)
    vae.load_state_dict(W.randomize_state_dict(vae.state_dict(), seed=10 + seed), strict=True)
    p = {k: v.clone().requires_grad_() for k, v in vae.state_dict().items()}
    P = vq.ops.resolve_precision(prec)
    ev = None
    if P.half_range():     # its own loss-scale domain + range-event counters (a train step polls them)
        P = (vq.ops.f16x3_region if prec == "f16x3" else vq.ops.fp16_region)("fuzz", 2.0 ** 8)
        P.events = ev = torch.zeros(12, dtype=torch.int32, device=device)
    vae = vae.to(device).set_precision(P)
    x = W.uniform_tensor(tuple(xshape), 50 + seed)
    recon, z = vae(x.to(device))
    rr, zr = M.vae_forward(p, x)
    p64 = {k: v.detach().double().requires_grad_() for k, v in p.items()}
    keep, keep_w = R.group_norm_fp32, R.wavelet_transform
    R.group_norm_fp32 = lambda t, ga, be, groups=32, eps=1e-6: torch.nn.functional.group_norm(t, groups, ga, be, eps)
    R.wavelet_transform = lambda t: keep_w(t.float()).double()          # (Haar sums of four pixels: fp32 filters)
    try:
        r64, z64 = M.vae_forward(p64, x.double())
    finally:
        R.group_norm_fp32, R.wavelet_transform = keep, keep_w
    if tuple(recon.shape) != tuple(rr.shape) or tuple(z.shape) != tuple(zr.shape):
        return False, f"shapes {tuple(recon.shape)} {tuple(z.shape)} vs {tuple(rr.shape)} {tuple(zr.shape)}"
    gy = W.uniform_tensor(tuple(rr.shape), 99)
    (recon * gy.to(device)).sum().backward(); (rr * gy).sum().backward(); (r64 * gy.double()).sum().backward()
    errs = {"recon": rel(recon, r64), "z": rel(z, z64)}
    yard = {"recon": rel(rr, r64), "z": rel(zr, z64)}
    per_el = sorted(float(p64[k].grad.norm()) / p64[k].grad.numel() ** 0.5 for k, _ in vae.named_parameters())
    typical = per_el[len(per_el) // 2]
    for k, q in vae.named_parameters():
        a, b, c = q.grad.detach().double().cpu(), p64[k].grad.detach(), p[k].grad.detach().double()
        den = max(float(c.norm()), 1e-3 * typical * c.numel() ** 0.5)
        errs[k] = float((a - b).norm()) / den
        yard[k] = float((c - b).norm()) / den
    excess = {k: errs[k] / max(tol, 10 * yard[k]) for k in errs}
    worst = max(excess, key=excess.get)
    ok = excess[worst] < 1
    note = ""
    if ev is not None and (int(ev[0]) or int(ev[6])):
        ok, note = True, f"  [range events: {int(ev[0])} gradient / {int(ev[6])} forward stores clipped — reported, as the train step requires]"
    return ok, f"worst {worst} {errs[worst]:.2e} (oracle fp32 vs fp64 there: {yard[worst]:.2e}; {len(errs)} tensors)" + note


def main():
    import tests.conftest as cf
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    prec = sys.argv[3] if len(sys.argv) > 3 else "fp32x6"
    device = os.environ.get("FUZZ_DEVICE", "cpu")       # "cuda": the product library on the GPU instead of the host emulator
    if device == "cpu":
        cf._build("emu", cf.EMU_LIB)
        vq._lib._set_library_for_tests(vq._lib.VqLibrary(cf.EMU_LIB))
    bad = 0
    only = os.environ.get("FUZZ_ONLY")
    for i in range(n):
        cfg, xshape = random_config(rnd)
        if only and str(i) not in only.split(","):
            continue
        try:
            ok, msg = check_config(cfg, xshape, prec, i, device)
        except RuntimeError as e:        # a configuration the kernels refuse must say so (and the reference refuses it too: see the message)
            ok, msg = "head dim" in str(e), "refused: " + str(e)[:200]
        except ValueError as e:          # one value per GroupNorm group: F.group_norm refuses it, and so does ops.gn_fwd_raw (raised by OUR side first)
            ok, msg = "vqgan" in "".join(traceback.format_tb(e.__traceback__)) and "more than 1 value" in str(e), "refused: " + str(e)[:200]
        bad += 0 if ok else 1
        print("ok  " if ok else "FAIL", i, cfg, xshape, msg, flush=True)
    print(f"{n - bad} / {n} ok")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
