"""Host-side guarantees around the fp16 ("ref") precision policy, on the emulator (`not gpu`) and on the MI355X (`gpu`):
  * the MFMA operand split of fp32-storage convs comes from the REGION's precision object (policy "fp32" = one bf16 product,
    "fp32x3" = the 3-term split), not from a process-wide default;
  * a calibration pass moves no state at all: parameters, optimizer counters, LeCam EMA, RNG streams, test hooks;
  * `--do_attn True` works under the default "ref" policy: the attention kernels take binary16 storage;
  * the live overflow signal: a loss scale that is far too large clips binary16 gradients -> the kernels' range-event counters
    fire, the optimizer step is DROPPED ON THE DEVICE (no parameter, no moment changes), run_training's poll re-calibrates and the
    following steps train again (the reference's fp32 / TF32 path cannot overflow: vae_trainer.py:18-19,538)."""
import random

import pytest
import torch

import vqgan_training_amd as vq
from vqgan_training_amd import ops
from oracle import model_ref as M
from oracle import weights as W


def _toy(dev, gan, policy, attn=False, lecam=False, **kw):
    res, ch = 16, 32 if not attn else 64
    vae = vq.ae.VAE(res, 3, ch, 3, [1, 2] if not attn else [1], 1, 4, attn, False, False)
    vae.load_state_dict(W.randomize_state_dict(vae.state_dict(), 1))
    lp = vq.utils.LPIPS(pretrained_path=None)
    lp.load_state_dict(W.randomize_state_dict(lp.state_dict(), 2, relu_net=True))
    disc = None
    if gan:
        disc = vq.utils.PatchDiscriminator()
        disc.load_state_dict(W.randomize_state_dict(disc.state_dict(), 4, relu_net=True))
    sds = (vae.state_dict(), lp.state_dict(), None if disc is None else disc.state_dict())
    vae, lp = vae.to(dev), lp.to(dev).eval()
    disc = disc.to(dev) if gan else None
    vq.vae_trainer.apply_precision_policy(policy, vae, lp, disc)
    step = vq.vae_trainer.VAETrainStep(vae, lp, disc, do_ganloss=gan, disc_type="hinge", use_lecam=lecam, learning_rate_vae=1e-2,
                                       vae_ch=ch, max_steps=20, warmup_steps=0, **kw)
    return step, vae, lp, disc, sds, W.image_batch(2, res, seed=8)


@pytest.mark.parametrize("policy,want", [("fp32", 1), ("fp32x3", 3)])
def test_fp32_policies_launch_the_split_their_name_says(backend, policy, want, monkeypatch):
    """Every conv descriptor of a step under policy fp32 carries split 1, under fp32x3 split 3 — whatever the process-wide default
    (set to the OPPOSITE here) says."""
    ops.set_default_precision("fp32" if want == 3 else "fp32x3")
    seen = []
    real = ops._desc

    def spy(*a, **k):
        d = real(*a, **k)
        seen.append((int(d.dtype), int(d.split)))
        return d

    monkeypatch.setattr(ops, "_desc", spy)
    try:
        step, *_rest, x = _toy(backend.device, False, policy)
        step(x.to(backend.device))
    finally:
        ops.set_default_precision("bf16")
    fp32_launches = [s for dt, s in seen if dt == vq._lib.VQ_F32]
    assert len(fp32_launches) > 20 and set(fp32_launches) == {want}, (policy, set(fp32_launches))


def test_calibration_pass_moves_no_state(backend):
    dev = backend.device
    fired = []
    rng = random.Random(5)
    step, vae, lp, disc, _sds, x = _toy(dev, True, "ref", lecam=True, on_backward=lambda s: fired.append("g"),
                                        on_d_backward=lambda s: fired.append("d"), rng=rng)
    step(x.to(dev))                                    # one real step: the EMA, the moments and the counters are non-trivial now
    fired.clear()
    before = {k: v.clone() for k, v in vae.state_dict().items()}
    before_d = {k: v.clone() for k, v in disc.state_dict().items()}
    anchor = step.lecam_anchor.clone()
    moments = [f.flat_m.clone() for f in step.optimizer_G._flat] + [f.flat_v.clone() for f in step.optimizer_G._flat]
    counters = (step.global_step, step.optimizer_G._step, step.optimizer_D._step)
    py_state, t_state = rng.getstate(), torch.get_rng_state()
    ev = step.range_events.clone()
    step.calibrate_grad_scales(x.to(dev), rounds=2 if backend.name == "gpu" else 1)
    assert fired == [], "test hooks must not see a calibration pass"
    assert torch.equal(step.lecam_anchor, anchor), "the LeCam EMA moved during calibration"
    assert all(torch.equal(before[k], v) for k, v in vae.state_dict().items())
    assert all(torch.equal(before_d[k], v) for k, v in disc.state_dict().items())
    now = [f.flat_m for f in step.optimizer_G._flat] + [f.flat_v for f in step.optimizer_G._flat]
    assert all(torch.equal(a, b) for a, b in zip(moments, now))
    assert counters == (step.global_step, step.optimizer_G._step, step.optimizer_D._step)
    assert rng.getstate() == py_state and torch.equal(torch.get_rng_state(), t_state)
    for win, tot in ((slice(0, 3), slice(3, 6)), (slice(6, 9), slice(9, 12))):      # gradient stores, forward stores
        assert torch.equal(step.range_events[:, win], torch.zeros_like(ev[:, win])) and torch.equal(step.range_events[:, tot], ev[:, tot])
    assert all(float(f.flat_g.abs().max()) == 0.0 for f in step.optimizer_G._flat)      # gradients zeroed afterwards


def test_attention_runs_under_the_default_ref_policy(backend):
    """`--do_attn True` with default flags (policy "ref": binary16 encoder): forward + backward + optimizer step, close to the fp32
    oracle of the same weights (ae.py:56-93; unreachable in the reference at HEAD)."""
    dev = backend.device
    step, vae, lp, disc, sds, x = _toy(dev, False, "ref", attn=True)
    exact = M.train_step_ref(M.RefState(*sds), x, do_ganloss=False, learning_rate_vae=1e-2, vae_ch=64, max_steps=20, warmup_steps=0)
    step.calibrate_grad_scales(x.to(dev), rounds=1)
    got = step(x.to(dev))
    rel = lambda a, b: abs(float(a) - float(b)) / max(abs(float(b)), 1e-30)   # noqa: E731
    assert rel(got["perceptual_loss"], exact["perceptual_loss"]) < 5e-3 and rel(got["overall_vae_loss"], exact["overall_vae_loss"]) < 5e-3
    err = ((got["z"].cpu() - exact["z"]).abs().max() / exact["z"].abs().max()).item()
    assert err < 1e-2, err


def test_clipped_gradients_drop_the_step_on_the_device_and_the_poll_recalibrates(backend):
    dev = backend.device
    gan = backend.name == "gpu"                      # (emulator minutes: the discriminator's twin of every assertion runs on the GPU)
    step, vae, lp, disc, _sds, x = _toy(dev, gan, "ref")
    xd = x.to(dev)
    step.calibrate_grad_scales(xd, rounds=2 if gan else 1)
    step(xd)
    healthy = step.poll_range_events()
    assert healthy["skipped_G"] == 0 and healthy["skipped_D"] == 0 and all(e["saturated"] == 0 for e in healthy["stacks"])
    good_scales = {p.region: p.grad_scale for p in step.fp16_stacks()}
    # ---- "the loss times 2^20 mid-run": every stack's loss scale is suddenly far too large
    for p in step.fp16_stacks():
        p.grad_scale *= 2.0 ** 20
    before = torch.cat([f.flat_p.clone() for f in step.optimizer_G._flat])
    before_m = torch.cat([f.flat_m.clone() for f in step.optimizer_G._flat])
    before_d = torch.cat([f.flat_p.clone() for f in step.optimizer_D._flat]) if gan else None
    g_step, a_step = step.global_step, step.optimizer_G._step
    out = step(xd)
    assert torch.isfinite(out["overall_vae_loss"]).item()          # the forward is untouched: only gradients carry the loss scale
    after = torch.cat([f.flat_p for f in step.optimizer_G._flat])
    assert torch.equal(before, after), "a step whose gradients were clipped must not reach the parameters"
    assert torch.equal(before_m, torch.cat([f.flat_m for f in step.optimizer_G._flat])), "... nor the Adam moments"
    if gan:
        assert torch.equal(before_d, torch.cat([f.flat_p for f in step.optimizer_D._flat])), "the discriminator's step is dropped too"
    ev = step.poll_range_events()                                  # the host learns about it at its logging cadence: ONE sync
    assert ev["skipped_G"] == 1 and ev["skipped_D"] == (1 if gan else 0) and sum(e["saturated"] for e in ev["stacks"]) > 0, ev
    assert (step.global_step, step.optimizer_G._step) == (g_step, a_step), "dropped steps do not advance the schedule / bias correction"
    # ---- what run_training does next: re-calibrate from measured maxima; the scales come back, the next step trains
    step.calibrate_grad_scales(xd, rounds=3)        # (a saturated pass backs off by 2^8: 2^20 takes two of those, then the measured one)
    for p in step.fp16_stacks():
        assert p.grad_scale <= good_scales[p.region] * 8 and p.grad_scale >= good_scales[p.region] / 8, (p.region, p.grad_scale)
    step(xd)
    ev = step.poll_range_events()
    assert ev["skipped_G"] == 0 and ev["skipped_D"] == 0 and all(e["saturated"] == 0 for e in ev["stacks"]), ev
    assert not torch.equal(before, torch.cat([f.flat_p for f in step.optimizer_G._flat])), "training resumed"


@pytest.mark.parametrize("policy,wide,wide_dtype", [("ref", "bf16", torch.bfloat16), ("f16x3", "fp32x6", torch.float32)])
def test_forward_saturation_is_reported_but_never_gates_the_optimizer(backend, policy, wide, wide_dtype):
    """forward conv stores of an fp16 stack used to feed the same counter the optimizers' device-side skip reads — an
    activation at binary16's limit (stored unscaled: no loss scale can help) would have dropped every G step for ever.  Now forward
    and gradient stores have their own counters: a saturating forward store is reported (`fwd_saturated`), the step is APPLIED, and
    after three polls in a row the stack is moved to a wider type (escalate_forward_saturation, what run_training does): a binary16
    stack to bf16, an f16x3 stack (two binary16 pieces: the same range; round-5 advice — it used to be polled but never moved) to fp32x6."""
    dev = backend.device
    step, vae, lp, disc, _sds, x = _toy(dev, False, policy)
    xd = x.to(dev)
    step.calibrate_grad_scales(xd, rounds=1)
    with torch.no_grad():                         # an encoder whose first activation leaves binary16's range: |conv_in(x)| ~ 1e6
        vae.encoder.conv_in.weight.mul_(3e6)
    ops.clear_caches()
    before = torch.cat([f.flat_p.clone() for f in step.optimizer_G._flat])
    for _ in range(3):
        step(xd)
        ev = step.poll_range_events()
        enc = next(e for e in ev["stacks"] if e["region"] == "encoder")
        assert enc["fwd_saturated"] > 0, ev
        assert ev["skipped_G"] == 0, "a forward clip must not drop the optimizer step"
    assert enc["fwd_saturated_polls"] == 3
    assert not torch.equal(before, torch.cat([f.flat_p for f in step.optimizer_G._flat])), "the steps were applied"
    moved = step.escalate_forward_saturation([e["region"] for e in ev["stacks"] if e["fwd_saturated_polls"] >= 3])
    assert moved == [("encoder", wide)] and vae.encoder.precision.dtype == wide_dtype
    rest = [p.region for p in step.fp16_stacks()]
    assert "encoder" not in rest and "lpips" in rest and step.range_events.shape == (len(rest), step.EV_COLS)
    assert "encoder" not in step._fwd_sat_polls
    if backend.name == "emu" and policy == "f16x3":
        return                                    # (the confirming step in fp32x6 — six products on the generic kernel — runs on the GPU only)
    out = step(xd)                                # the wider type holds 1e6: the step runs, nothing saturates
    ev = step.poll_range_events()
    assert torch.isfinite(out["overall_vae_loss"]).item() and all(e["fwd_saturated"] == 0 for e in ev["stacks"]), ev


def test_headroom_events_lower_a_loss_scale_before_anything_clips(backend):
    """Round 6: kernels count the waves that stored a binary16 gradient of 2^13 or more (range-event counter 2: three bits under the
    limit, nothing lost).  With a loss scale 2^4 too high for the encoder's gradients the step is APPLIED (no clip, no drop), the poll
    reports `headroom` events for that stack alone, `relax_hot_scales` lowers its scale by 2^3 — no pass over the model, parameters and
    optimizer state untouched — and the next step is silent again.  (The reference's fp32 / TF32 path never drops a step:
    vae_trainer.py:525-708; here a dropped step is the last resort, this is the first.)"""
    dev = backend.device
    step, vae, lp, disc, _sds, x = _toy(dev, False, "ref")
    xd = x.to(dev)
    step.calibrate_grad_scales(xd, rounds=2)               # every stack's largest gradient at ~2^10
    step(xd)
    assert all(e["headroom"] == 0 and e["saturated"] == 0 for e in step.poll_range_events()["stacks"])
    enc = vae.encoder.precision
    s0 = enc.grad_scale
    enc.grad_scale = s0 * 16.0                             # largest gradient now ~2^14: past 2^13, below 2^16
    before = torch.cat([f.flat_p.clone() for f in step.optimizer_G._flat])
    step(xd)
    ev = step.poll_range_events()
    row = {e["region"]: e for e in ev["stacks"]}
    assert row["encoder"]["headroom"] > 0 and row["encoder"]["saturated"] == 0 and ev["skipped_G"] == 0, ev
    assert row["lpips"]["headroom"] == 0, "the other stacks' scales were fine"
    assert not torch.equal(before, torch.cat([f.flat_p for f in step.optimizer_G._flat])), "the step was applied"
    moved = step.relax_hot_scales(ev)
    assert [r for r, _ in moved] == ["encoder"] and enc.grad_scale == s0 * 16.0 / 8.0
    step(xd)
    ev = step.poll_range_events()
    assert all(e["headroom"] == 0 and e["saturated"] == 0 for e in ev["stacks"]) and ev["skipped_G"] == 0, ev


def test_discriminator_step_leaves_the_other_stacks_windows_alone_when_it_is_not_an_fp16_stack(backend):
    """with hand-assigned precisions (encoder binary16, discriminator bf16) the D step used to close EVERY stack's
    saturation window — the encoder's clipped gradients... of the G step that follows were safe, but clips recorded before the D
    step (the encoder forward / the previous window) were wiped, and a D step that WAS applied got counted as dropped.  Now the D
    side is a no-op when the discriminator has no counters of its own."""
    dev = backend.device
    step, vae, lp, disc, _sds, x = _toy(dev, True, "bf16")
    vae.encoder.precision = ops.fp16_region("encoder", 2.0 ** 10)
    step.bind_range_events()
    assert step._disc_row() is None and step.optimizer_D.skip_flags is None and step.optimizer_G.skip_flags is not None
    vae.encoder.precision.grad_scale = 2.0 ** 40                      # the encoder's gradients will clip in the G backward
    d_before = torch.cat([f.flat_p.clone() for f in step.optimizer_D._flat])
    g_before = torch.cat([f.flat_p.clone() for f in step.optimizer_G._flat])
    step.range_events[0, 0] = 7                                       # a pending gradient clip when the D step comes by
    step._close_window(1, row=step._disc_row())
    assert int(step.range_events[0, 0]) == 7 and int(step._skipped[1]) == 0, "the D step must not touch the encoder's window"
    step.range_events[0, 0] = 0
    step(x.to(dev))
    ev = step.poll_range_events()
    assert ev["skipped_D"] == 0 and ev["skipped_G"] == 1, ev
    assert not torch.equal(d_before, torch.cat([f.flat_p for f in step.optimizer_D._flat])), "D's step was applied"
    assert torch.equal(g_before, torch.cat([f.flat_p for f in step.optimizer_G._flat])), "G's step was dropped on the device"


def test_run_training_logs_the_reference_scalar_names(backend):
    """vae_trainer.py:713-748: the names the reference sends to wandb every 5 steps, from the device-side statistics of the step."""
    hist = vq.vae_trainer.run_training(batch_size=2, do_ganloss=True, disc_type="hinge", vae_resolution=16, vae_ch=32, vae_ch_mult="1,2",
                                       vae_num_res_blocks=1, vae_z_channels=4, max_steps=1 if backend.name == "emu" else 3,
                                       evaluate_every_n_steps=0, precision="bf16", backend="gloo" if backend.name == "emu" else "nccl",
                                       log_every=1)
    assert len(hist) == (1 if backend.name == "emu" else 3)
    want = {"overall_vae_loss", "mse_loss", "kl_loss", "perceptual_loss", "gan/generator_gan_loss", "z_quantiles/abs_z",
            "z_quantiles/std_z", "z_quantiles/logvar", "gan/avg_real_logits", "gan/avg_fake_logits", "gan/discriminator_loss",
            "gan/discriminator_accuracy", "gan/lecam_loss", "gan/lecam_anchor_real_logits", "gan/lecam_anchor_fake_logits",
            "z_quantiles/qs", "time_taken_till_step"}
    assert want <= set(hist[0]), want - set(hist[0])
    assert set(hist[0]["z_quantiles/qs"]) == {"0.0", "0.2", "0.4", "0.6", "0.8", "1.0", "kurtosis", "skewness"}
    assert 0.0 <= hist[0]["gan/discriminator_accuracy"] <= 1.0 and hist[0]["mse_loss"] == 0.0
