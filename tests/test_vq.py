"""VQ nearest-code lookup: BIT-EXACT indices against oracle/vq_oracle.c (the reference has no
quantizer — SURVEY F1 — so the oracle defines the algorithm; parity unpinned), plus the quantizer
module's straight-through / commitment / codebook gradients against the plain-torch restatement."""
import os

import pytest
import torch

import vqgan_training_amd as vq
from vqgan_training_amd.quantizer import VectorQuantizer
from oracle import vq_oracle
from oracle import weights as W


def _lookup(dev, z, cb):
    from vqgan_training_amd.quantizer import _VQLookup
    zq, loss, idx = _VQLookup.apply(z.to(dev), cb.to(dev), 0.25)
    return idx.cpu(), zq.cpu()


@pytest.mark.parametrize("n,k,d", [(300, 1000, 32), (257, 129, 4), (64, 513, 8), (1000, 77, 16), (5, 3, 64), (1, 1, 32),
                                   (8200, 3000, 32), (8200, 1000, 64)])   # fp32-MFMA search: several code tiles per split, ragged last tile, split and token block
def test_indices_bit_exact_random(backend, n, k, d):
    z = W.uniform_tensor((n, d), 100 + n, -1, 1)
    cb = W.uniform_tensor((k, d), 200 + k, -1, 1)
    idx, zq = _lookup(backend.device, z, cb)
    want, _ = vq_oracle.nearest(z, cb)
    assert torch.equal(idx, want)
    assert torch.equal(zq, cb[want])


def test_indices_ties_and_near_ties(backend):
    """Exact duplicates must resolve to the lowest index; codes one ulp apart and tokens placed on the
    bisector between two codes must resolve exactly as the oracle's fixed fmaf order does."""
    d, k = 32, 512
    cb = W.uniform_tensor((k, d), 7, -1, 1)
    cb[300] = cb[17]                               # exact duplicate -> 17 wins
    cb[301] = cb[17]
    nxt = torch.nextafter(cb[40], torch.full((d,), 2.0))
    cb[41] = nxt                                   # one ulp away from code 40
    z = torch.cat([cb[17:18] + 1e-3, cb[40:41], (cb[40:41] + cb[41:42]) / 2, (cb[5:6] + cb[6:7]) / 2,
                   W.uniform_tensor((60, d), 9, -1, 1)])
    # adversarial: many tokens exactly half-way between random code pairs
    a, b = cb[torch.arange(0, 200, 2)], cb[torch.arange(1, 200, 2)]
    z = torch.cat([z, (a + b) / 2])
    idx, _ = _lookup(backend.device, z, cb)
    want, _ = vq_oracle.nearest(z, cb)
    assert torch.equal(idx, want)
    assert idx[0].item() == 17


@pytest.mark.gpu
def test_indices_bit_exact_config5_size(hip_library):
    """BASELINE config 5: 8192 tokens/GPU x 16384 codes x dim 32."""
    vq._lib._set_library_for_tests(hip_library)
    try:
        z = W.uniform_tensor((8192, 32), 1, -1, 1)
        cb = W.uniform_tensor((16384, 32), 2, -1, 1)
        idx, _ = _lookup(torch.device("cuda:0"), z, cb)
        want, _ = vq_oracle.nearest(z, cb)
        assert torch.equal(idx, want)
    finally:
        vq._lib._set_library_for_tests(None)


def test_codebook_scatter_add_is_order_independent(backend):
    """dcodebook[idx_i] += gq_i through 64-bit fixed-point accumulation (vq_vq_scatter_add): bit-identical under any permutation
    of the tokens (fp32 atomics are not), within a few ulp of the fp64 sum, magnitudes spanning 2^-20..1 in one call, heavy
    collisions on a few codes, accumulation onto existing contents."""
    from vqgan_training_amd._lib import lib, ptr, stream_of, workspace
    dev = backend.device
    n, K, D = 1024, 32, 8
    g = torch.Generator().manual_seed(3)
    gq = torch.randn(n, D, generator=g) * torch.exp2(-20 * torch.rand(n, 1, generator=g))
    idx = (torch.rand(n, generator=g) ** 3 * K).long().clamp(0, K - 1)              # skewed: code 0 collects a third of the tokens
    base = torch.randn(K, D, generator=g) * 0.01
    want = base.double().index_add(0, idx, gq.double())
    outs = []
    for perm in (torch.arange(n), torch.randperm(n, generator=g), torch.arange(n).flip(0)):
        dcb = base.clone().to(dev)
        L = lib()
        ws = workspace(dev, L.size("vq_vq_scatter_workspace", K, D), slot=1)
        gp, ip = gq[perm].contiguous().to(dev), idx[perm].contiguous().to(dev)      # (kept alive across the call)
        L.call("vq_vq_scatter_add", ptr(gp), ptr(ip), n, K, D, ptr(dcb), ptr(ws), ws.numel(), stream_of(dcb))
        outs.append(dcb.cpu())
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    assert (outs[0].double() - want).abs().max().item() <= 4e-7 * want.abs().max().item()


def test_quantizer_module_matches_oracle(backend):
    dev = backend.device
    q = VectorQuantizer(n_codes=64, dim=8, beta=0.25)
    q.embedding.weight.data.copy_(W.uniform_tensor((64, 8), 3, -1, 1))
    z = W.uniform_tensor((2, 8, 4, 4), 4, -1, 1)
    cbr = q.embedding.weight.detach().clone().requires_grad_()
    zr = z.clone().requires_grad_()
    out_r, loss_r, idx_r = vq_oracle.quantize(zr, cbr, 0.25)
    gy = W.uniform_tensor(tuple(out_r.shape), 5)
    (out_r * gy).sum().backward(retain_graph=True)
    (3.0 * loss_r).backward()
    q = q.to(dev)
    zd = z.to(dev).requires_grad_()
    out, loss, idx = q(zd)
    ((out * gy.to(dev)).sum() + 3.0 * loss).backward()
    assert torch.equal(idx.cpu(), idx_r)
    assert torch.allclose(out.detach().cpu(), out_r.detach(), atol=1e-6)
    assert abs(loss.item() - loss_r.item()) < 1e-6 * max(1.0, abs(loss_r.item()))
    assert torch.allclose(zd.grad.cpu(), zr.grad, atol=1e-6)
    assert torch.allclose(q.embedding.weight.grad.cpu(), cbr.grad, atol=1e-6)


def test_lookup_from_an_exact_evaluation_while_gradients_flow_through_another(backend):
    """VectorQuantizer(z, lookup_from=z_exact) — policy ref_vq: the nearest-code search reads an fp32-class evaluation of the encoder,
    the losses / straight-through output / gradients use the binary16 one.  Indices == the oracle's on z_exact (bit-exact, although
    z itself would pick other codes for some tokens), output rows = codebook[idx], loss and gradients = the published formulas
    evaluated with z and those indices (plain torch restatement below)."""
    dev = backend.device
    K, D, beta = 64, 8, 0.25
    q = VectorQuantizer(n_codes=K, dim=D, beta=beta)
    q.embedding.weight.data.copy_(W.uniform_tensor((K, D), 3, -1, 1))
    z_exact = W.uniform_tensor((2, D, 4, 4), 4, -1, 1)
    z = z_exact + 0.15 * W.uniform_tensor((2, D, 4, 4), 6, -1, 1)          # a perturbed evaluation: some tokens would flip
    cb = q.embedding.weight.detach().clone()
    _, _, idx_exact = vq_oracle.quantize(z_exact, cb, beta)
    _, _, idx_pert = vq_oracle.quantize(z, cb, beta)
    assert (idx_exact != idx_pert).any(), "the perturbation must matter for the test to mean anything"
    # restatement with the indices FIXED to the exact ones
    zr, cbr = z.clone().requires_grad_(), cb.clone().requires_grad_()
    tok = zr.permute(0, 2, 3, 1).reshape(-1, D)
    zq = cbr[idx_exact.reshape(-1)]
    loss_r = beta * (zq.detach() - tok).pow(2).mean() + (zq - tok.detach()).pow(2).mean()
    out_r = (tok + (zq - tok).detach()).reshape(2, 4, 4, D).permute(0, 3, 1, 2)
    gy = W.uniform_tensor(tuple(out_r.shape), 5)
    ((out_r * gy).sum() + 3.0 * loss_r).backward()
    q = q.to(dev)
    zd = z.to(dev).requires_grad_()
    out, loss, idx = q(zd, lookup_from=z_exact.to(dev))
    ((out * gy.to(dev)).sum() + 3.0 * loss).backward()
    assert torch.equal(idx.cpu(), idx_exact)
    assert torch.allclose(out.detach().cpu(), out_r.detach(), atol=1e-6)
    assert abs(loss.item() - loss_r.item()) < 1e-6 * max(1.0, abs(loss_r.item()))
    assert torch.allclose(zd.grad.cpu(), zr.grad, atol=1e-6)
    assert torch.allclose(q.embedding.weight.grad.cpu(), cbr.grad, atol=1e-6)


def test_train_step_with_quantizer_matches_oracle(backend):
    """Config-5 wiring: encoder -> VQ (in place of `reg`) -> decoder -> LPIPS, codebook in optimizer_G's main group.
    Indices bit-exact, losses to 1e-4, codebook updated like the oracle's AdamW."""
    from oracle import model_ref as M
    from oracle import weights as W
    from vqgan_training_amd import ops
    dev = backend.device
    ops.set_default_precision("fp32x3")
    res, ch, mult, zc, K = (32 if backend.name == "gpu" else 16), 32, [1, 2], 4, 64
    vae = vq.ae.VAE(res, 3, ch, 3, list(mult), 1, zc, False, False, False)
    vae.load_state_dict(W.randomize_state_dict(vae.state_dict(), 1))
    lp = vq.utils.LPIPS(pretrained_path=None)
    lp.load_state_dict(W.randomize_state_dict(lp.state_dict(), 2, relu_net=True))
    quant = vq.quantizer.VectorQuantizer(K, zc, beta=0.25)
    with torch.no_grad():
        quant.embedding.weight.copy_(W.uniform_tensor((K, zc), 77, -1.5, 1.5))
    sd = dict(vae.state_dict()); sd[M.VQ_KEY] = quant.embedding.weight.detach().clone()
    st = M.RefState(sd, lp.state_dict(), None)
    vae, lp, quant = vae.to(dev), lp.to(dev).eval(), quant.to(dev)
    step = vq.vae_trainer.VAETrainStep(vae, lp, None, learning_rate_vae=1e-2, vae_ch=ch, max_steps=10, warmup_steps=1,
                                       quantizer=quant)
    x = W.image_batch(2, res, seed=8)
    o = step(x.to(dev))
    r = M.train_step_ref(st, x, learning_rate_vae=1e-2, vae_ch=ch, max_steps=10, warmup_steps=1)
    assert torch.equal(o["indices"].cpu(), r["indices"])
    for k in ("overall_vae_loss", "perceptual_loss", "vae_loss", "vq_loss"):
        a, b = float(o[k]), float(r[k])
        assert abs(a - b) <= 1e-4 * abs(b) + 1e-7, (k, a, b)
    assert (quant.embedding.weight.detach().cpu() - st.vae[M.VQ_KEY].detach()).abs().max().item() < 1e-3
    assert len(set(o["indices"].flatten().tolist())) > 4              # the test really quantizes to several codes


def test_train_step_with_attention_and_quantizer_under_ref_vq(backend):
    """`--do_attn True` with a quantizer under the workload's own policy (`ref_vq`: binary16 encoder for the gradients + a gradient-
    free f16x3 evaluation for the code lookup): the AttnBlocks run in BOTH storage types (round-5 advice: the f16x3 lookup pass raised
    NotImplementedError in _Attention).  Indices bit-exact against the oracle, losses to the policy's own accuracy."""
    from oracle import model_ref as M
    from oracle import weights as W
    from vqgan_training_amd import ops
    dev = backend.device
    ops.set_default_precision("bf16")
    res, ch, mult, zc, K = 16, 32, [1, 2], 4, 64
    vae = vq.ae.VAE(res, 3, ch, 3, list(mult), 1, zc, True, False, False)
    vae.load_state_dict(W.randomize_state_dict(vae.state_dict(), 1))
    lp = vq.utils.LPIPS(pretrained_path=None)
    lp.load_state_dict(W.randomize_state_dict(lp.state_dict(), 2, relu_net=True))
    quant = vq.quantizer.VectorQuantizer(K, zc, beta=0.25)
    with torch.no_grad():
        quant.embedding.weight.copy_(W.uniform_tensor((K, zc), 77, -1.5, 1.5))
    sd = dict(vae.state_dict()); sd[M.VQ_KEY] = quant.embedding.weight.detach().clone()
    st = M.RefState(sd, lp.state_dict(), None)
    vae, lp, quant = vae.to(dev), lp.to(dev).eval(), quant.to(dev)
    vq.vae_trainer.apply_precision_policy("ref_vq", vae, lp, None)
    step = vq.vae_trainer.VAETrainStep(vae, lp, None, learning_rate_vae=1e-3, vae_ch=ch, max_steps=10, warmup_steps=1, quantizer=quant)
    x = W.image_batch(2, res, seed=8)
    step.calibrate_grad_scales(x.to(dev))
    o = step(x.to(dev))
    r = M.train_step_ref(st, x, learning_rate_vae=1e-3, vae_ch=ch, max_steps=10, warmup_steps=1)
    assert torch.equal(o["indices"].cpu(), r["indices"])
    for k in ("overall_vae_loss", "perceptual_loss", "vq_loss"):
        a, b = float(o[k]), float(r[k])
        assert abs(a - b) <= 3e-2 * abs(b) + 1e-6, (k, a, b)
    ops.clear_caches()


def test_run_training_from_an_iterable_with_a_quantizer_evaluates_and_checkpoints_the_codebook(backend, tmp_path, monkeypatch):
    """run_training on caller-supplied batches (any iterable of [-1,1] NCHW tensors or (tensor, label) pairs: the reference's
    loader yields pairs, vae_trainer.py:530) instead of synthetic noise, with the VQ quantizer in `reg`'s place: the loop stops
    at num_epochs passes, the evaluation grid goes through the quantizer, the checkpoint carries the codebook under
    `quantizer.*` next to the reference-format VAE keys, and --load_path restores both (strict)."""
    from oracle import weights as W
    from vqgan_training_amd import ops, vae_trainer as T
    monkeypatch.chdir(tmp_path)
    for var in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        monkeypatch.delenv(var, raising=False)
    monkeypatch.setattr(ops, "_default_precision", ops.default_precision())
    monkeypatch.setattr(ops, "_fp32_split", ops._fp32_split)
    res, zc, K = (32 if backend.name == "gpu" else 16), 4, 32
    train = [W.image_batch(2, res, seed=20), (W.image_batch(2, res, seed=21), torch.zeros(2))]        # host tensors, one as a pair
    test = (W.image_batch(2, res, seed=30 + i) for i in range(3))                                      # a generator: two are used
    quant = vq.quantizer.VectorQuantizer(K, zc, beta=0.25)
    with torch.no_grad():
        quant.embedding.weight.copy_(W.uniform_tensor((K, zc), 77, -1.5, 1.5))
    kw = dict(batch_size=2, vae_resolution=res, vae_ch=32, vae_ch_mult="1,2", vae_num_res_blocks=1, vae_z_channels=zc,
              run_name="it", precision="bf16", backend="nccl" if backend.name == "gpu" else "gloo", synthetic=False, log_every=1)
    seen = []
    orig_eval = T.evaluate
    monkeypatch.setattr(T, "evaluate", lambda *a, **k: (seen.append((len(a[1]), k.get("quantizer") is not None)), orig_eval(*a, **k))[1])
    hist = T.run_training(train_batches=train, test_batches=test, quantizer=quant, num_epochs=2, max_steps=100,
                          evaluate_every_n_steps=3, **kw)
    assert len(hist) == 4 and all(v == v for h in hist for v in h.values())       # 2 epochs x 2 batches, then the iterable is dry
    assert seen == [(2, True), (2, True)]                                        # steps 0 and 3, two test batches, through the quantizer
    ck = tmp_path / "ckpt" / "it" / "vae_epoch_0_step_4.pt"
    sd = torch.load(ck, map_location="cpu")
    assert "quantizer.embedding.weight" in sd and "module.encoder.conv_in.weight" in sd
    trained = quant.embedding.weight.detach().cpu().clone()
    assert not torch.equal(trained, W.uniform_tensor((K, zc), 77, -1.5, 1.5))     # the codebook trained ...
    # ... and a fresh run resumes from it (and from the VAE weights) instead of a fresh codebook
    quant2 = vq.quantizer.VectorQuantizer(K, zc, beta=0.25)
    vae2 = vq.ae.VAE(res, 3, 32, 3, [1, 2], 1, zc, False, False, False)
    T.load_checkpoint(vae2, str(ck), quantizer=quant2)
    assert torch.equal(quant2.embedding.weight.detach().cpu(), sd["quantizer.embedding.weight"])
    assert torch.equal(vae2.encoder.conv_in.weight.detach().cpu(), sd["module.encoder.conv_in.weight"])
    T.load_checkpoint(vq.ae.VAE(res, 3, 32, 3, [1, 2], 1, zc, False, False, False), str(ck))          # a VAE-only consumer ignores the codebook
    with pytest.raises(NotImplementedError):
        T.run_training(max_steps=1, **kw)                                         # synthetic=False without batches: loud


def test_bench_parity_leg_of_the_quantized_workload(emu_library, monkeypatch):
    """`bench.py --workload c5` carries a `parity` object (bench.parity_quantized): one image through the full quantized step on
    re-randomised weights against the oracle — identical code indices and 1e-4 losses in the parity mode, the number of tokens
    that pick another code at the timed policy.  Here on the emulator with a small model; the GPU run uses configs[4] itself."""
    import bench
    from vqgan_training_amd import ops
    monkeypatch.setattr(bench, "TEST_DEVICE", "emu")
    monkeypatch.setitem(bench._TEST_SHRINK, "calibrate_rounds", 1)          # emulator seconds, not behaviour
    vq._lib._set_library_for_tests(emu_library)
    ops.clear_caches()
    try:
        cfg = {"ch": 32, "ch_mult": (1, 2), "z": 4, "res": 16, "gan": False, "vq": (128, 4)}
        out = bench.parity_quantized("ref_vq", cfg, 16, torch.device("cpu"))
    finally:
        vq._lib._set_library_for_tests(None)
        ops.clear_caches()
    assert out["tokens"] == 64 and out["codes_used_by_the_oracle"] > 16
    pm = out["parity_mode"]
    assert pm["precision"] == "fp32x6" and pm["indices_identical"] and pm["tokens_with_another_code"] == 0
    for k in ("perceptual_loss_rel", "overall_vae_loss_rel", "vq_loss_rel"):
        assert pm[k] < 1e-4, (k, pm)
    assert out["fp32x3"]["indices_identical"]
    tp = out["timed_policy"]                    # the workload's default policy: fp32-class encoder in front of the lookup
    assert tp["precision"] == "ref_vq" and tp["indices_identical"] and tp["tokens_with_another_code"] == 0, tp
    rp = out["ref_policy"]                      # binary16 encoder: near-ties may flip
    assert rp["precision"] == "ref" and rp["tokens_with_another_code"] <= 0.05 * out["tokens"] and rp["vq_loss_rel"] < 5e-3


@pytest.mark.gpu
def test_config5_policy_keeps_the_code_indices_bit_exact_at_512(hip_library):
    """BASELINE configs[4] as stated — VQ 16384 x 32, vae_ch=128 ch_mult=1,2,4,4,4 (f=16), 512x512, full loss — one image through the
    whole quantized step (bench.parity_quantized) against oracle/model_ref.py + oracle/vq_oracle.c on the box's host cores:
    north_star's "bit-exact for the VQ argmin indices" holds END TO END under the workload's timed policy (`ref_vq`: the encoder,
    whose output the integer lookup reads, in the fp32-class split; LPIPS / discriminator binary16, decoder bf16) and in both parity
    modes; the plain `ref` policy (binary16 encoder) is reported beside it — its rounding moves a few near-tie tokens."""
    import bench
    from vqgan_training_amd import ops
    vq._lib._set_library_for_tests(hip_library)
    ops.clear_caches()
    try:
        cfg = {"ch": 128, "ch_mult": (1, 2, 4, 4, 4), "z": 32, "res": 512, "gan": True, "vq": (16384, 32)}
        torch.set_num_threads(min(32, os.cpu_count() or 8))
        out = bench.parity_quantized("ref_vq", cfg, 512, torch.device("cuda:0"))
    finally:
        vq._lib._set_library_for_tests(None)
        ops.clear_caches()
    print("configs[4] parity:", {k: v for k, v in out.items() if k != "vs"})
    assert out["tokens"] == 1024 and out["codes_used_by_the_oracle"] > 256
    for name in ("parity_mode", "fp32x3", "timed_policy"):
        assert out[name]["indices_identical"] and out[name]["tokens_with_another_code"] == 0, (name, out[name])
        assert out[name]["vq_loss_rel"] < 1e-4 and out[name]["d_loss_rel"] < (1e-4 if name != "timed_policy" else 2e-3), (name, out[name])
    pm = out["parity_mode"]
    for k in ("perceptual_loss_rel", "overall_vae_loss_rel", "vq_loss_rel", "d_loss_rel", "g_gan_loss_rel"):
        assert pm[k] < 1e-4, (k, pm)
    assert out["ref_policy"]["tokens_with_another_code"] <= 0.02 * out["tokens"]
