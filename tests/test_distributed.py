"""The N>1 path on CPU: 2 ranks, gloo, kernels through the host emulator (tests/dist_worker.py).

Checks the data-parallel exchange of DESIGN.md §multi-GPU:
  * bucketed in-place all-reduce + 1/world folded into AdamW  => all ranks hold identical parameters
    after optimizer steps although they saw different batches (the reference does NOT have this
    property: SURVEY F2), and `sync_vae_grads=False` reproduces the reference (ranks diverge);
  * GradNorm's backward divides by the MEAN over ranks of the per-rank ||g||_2 (vae_trainer.py:40-44).
"""
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="session")   # (session: the row-first test order interleaves modules)
def two_rank_results(tmp_path_factory, emu_library):
    """ONE 2-rank launch runs both scenarios (with and without the VAE gradient exchange) back to back."""
    out = tmp_path_factory.mktemp("dist")
    env = dict(os.environ, VQ_DIST_OUT=str(out), VQ_DIST_MODE="reducer,sync,gan,ddp", OMP_NUM_THREADS="4", VQ_EMU_THREADS="4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29611", os.path.join(ROOT, "tests", "dist_worker.py")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    return {mode: [torch.load(os.path.join(out, f"rank{k}_{mode}.pt")) for k in range(2)] for mode in ("reducer", "sync", "gan", "ddp")}


def test_bucketed_allreduce_keeps_ranks_in_lockstep(two_rank_results):
    r0, r1 = two_rank_results["sync"]
    assert r0["world"] == 2 and r0["n_buckets"] >= 2 and r0["grad_scale"] == 0.5
    # identical parameters on both ranks after an optimizer step on different data
    for k in r0["params"]:
        assert torch.equal(r0["params"][k], r1["params"][k]), k
    # after finish() both ranks hold the same summed gradients in their flat buffers
    diff = max((r0["local_grads"][k] - r1["local_grads"][k]).abs().max().item() for k in r0["local_grads"])
    assert diff == 0
    # avg_scalar_over_nodes (vae_trainer.py:56-60): rank 0 passed 1.0, rank 1 passed 4.0
    assert r0["avg_scalar"] == r1["avg_scalar"] == 2.5 and isinstance(r0["avg_scalar"], float)
    # GradNorm: both ranks scale by the mean of the two norms
    n0, n1 = r0["gradnorm_g"].norm(), r1["gradnorm_g"].norm()
    mean = (n0 + n1) / 2
    assert torch.allclose(r0["gradnorm_probe"], r0["gradnorm_g"] / (mean + 1e-8), rtol=1e-5, atol=1e-8)
    assert torch.allclose(r1["gradnorm_probe"], r1["gradnorm_g"] / (mean + 1e-8), rtol=1e-5, atol=1e-8)


def test_reducer_modes_without_kernels(two_rank_results):
    """The reducer by itself (plain CPU tensors): the discriminator's single exchange in finish() (overlap=False), the
    autograd-hook countdown, a parameter without gradient (its bucket still takes part in the exchange), re-arming."""
    r0, r1 = two_rank_results["reducer"]
    n = r0["a_sum"].numel()
    assert r0["a_buckets"] >= 2 and torch.equal(r0["a_sum"], r1["a_sum"])
    assert torch.equal(r0["a_sum"], torch.arange(n, dtype=torch.float32) * 3)          # rank 0: x1, rank 1: x2
    assert r0["b_buckets"] >= 3 and r0["b_handles_left"] == 0
    sizes = [6, 9, 8, 5]
    for it in range(2):
        want = torch.cat([torch.zeros(sz) if k == 2 else torch.full((sz,), float((1 + it + k) + (2 + it + k)))
                          for k, sz in enumerate(sizes)])
        assert torch.equal(r0[f"b_sum{it}"], want) and torch.equal(r1[f"b_sum{it}"], want), it
    # both report channels (kernel gradient sinks + autograd's post-accumulate hook) feed the same countdown; a re-registered sink
    # keeps the reducer's callback; every bucket was on the wire before finish(); a late second contribution raises
    assert r0["d_callback_kept"] and r1["d_callback_kept"]
    for it in range(2):
        want = torch.cat([torch.full((sz,), float((1 + it + k) + (2 + it + k))) for k, sz in enumerate(sizes)])
        assert torch.equal(r0[f"d_sum{it}"], want) and torch.equal(r1[f"d_sum{it}"], want), it
        assert r0[f"d_launched_before_finish{it}"] == r0["b_buckets"], "overlap: all buckets launched from the two hook channels"
    assert r0["e_raised"] is True and r1["e_raised"] is True
    # coalesced parameter broadcast (vae_trainer.py:438,450: DDP's constructor): rank 1 ends with rank 0's parameters and buffers, in
    # fewer collectives than tensors (fp32 tensors in buckets of 600 bytes here + one int64 buffer)
    assert set(r0["g_state"]) == set(r1["g_state"]) and all(torch.equal(r0["g_state"][k], r1["g_state"][k]) for k in r0["g_state"])
    assert int(r1["g_state"]["1.num_batches_tracked"]) == 5
    assert 2 <= r0["g_collectives"] < r0["g_n_tensors"] and r0["g_collectives"] == r1["g_collectives"]


def test_bucket_launch_order_is_identical_when_ranks_report_gradients_in_different_orders(two_rank_results):
    """Rank 0 reports its gradients in backward order, rank 1 in the opposite host order (dist_worker case f): the buckets still go on
    the wire in index order on both ranks — a bucket that completes early waits for its predecessors — so every all-reduce pairs
    the same slice on both sides (collectives are matched by issue order; sizes 9 / 7 / 5 / 3 would not even match otherwise)."""
    r0, r1 = two_rank_results["reducer"]
    assert r0["f_buckets"] == 4
    for it in range(2):
        assert r0["f_logs"][it] == r1["f_logs"][it] == [0, 1, 2, 3], (r0["f_logs"], r1["f_logs"])
        want = torch.cat([torch.full((sz,), float(1 * (k + 1) + it + 2 * (k + 1) + it)) for k, sz in enumerate((3, 5, 7, 9))])
        assert torch.equal(r0[f"f_sum{it}"], want) and torch.equal(r1[f"f_sum{it}"], want), it


def test_reference_behaviour_without_vae_grad_sync(two_rank_results):
    """--sync_vae_grads False == the reference (the DDP wrapper around the VAE is never armed): the reducer is inert —
    no buckets, no hooks, gradients stay rank-local, AdamW applies them unscaled — so replicas that see different batches drift."""
    r0, r1 = two_rank_results["reducer"]
    assert r0["off_enabled"] is False and r0["off_buckets"] == 0 and r0["off_grad_scale"] == 1.0
    assert not torch.equal(r0["off_grads"], r1["off_grads"])            # nothing was exchanged: rank-local gradients
    assert torch.equal(r0["off_grads"], torch.arange(r0["off_grads"].numel(), dtype=torch.float32))        # rank 0's own values, untouched


@pytest.fixture(scope="session")   # (session: the row-first test order interleaves modules)
def eight_rank_results(tmp_path_factory, emu_library):
    """The same worker at the world size of the reference's launch line (launcher.sh:3-9: `torchrun --nproc_per_node=8`): the reducer
    cases and one full synchronised step with 8 gloo ranks (one emulator thread each), so that the first run on an 8-GPU node is
    not also the first 8-rank run of the exchange logic."""
    out = tmp_path_factory.mktemp("dist8")
    env = dict(os.environ, VQ_DIST_OUT=str(out), VQ_DIST_MODE="reducer,sync", OMP_NUM_THREADS="1", VQ_EMU_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr",
           "127.0.0.1", "--master-port", "29617", os.path.join(ROOT, "tests", "dist_worker.py")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    return {mode: [torch.load(os.path.join(out, f"rank{k}_{mode}.pt")) for k in range(8)] for mode in ("reducer", "sync")}


def test_eight_ranks_stay_in_lockstep(eight_rank_results):
    rs = eight_rank_results["sync"]
    assert all(r["world"] == 8 and r["grad_scale"] == 0.125 and r["n_buckets"] >= 2 for r in rs)
    for r in rs[1:]:                                   # identical parameters everywhere after a step on 8 different batches
        for k in rs[0]["params"]:
            assert torch.equal(rs[0]["params"][k], r["params"][k]), (r["rank"], k)
        assert max((rs[0]["local_grads"][k] - r["local_grads"][k]).abs().max().item() for k in rs[0]["local_grads"]) == 0
    assert all(r["avg_scalar"] == 11.5 for r in rs)    # mean of 3 * rank + 1 over 8 ranks (vae_trainer.py:56-60)
    mean = sum(r["gradnorm_g"].norm() for r in rs) / 8   # GradNorm: the mean over ranks of the per-rank norms (vae_trainer.py:40-44)
    for r in rs:
        assert torch.allclose(r["gradnorm_probe"], r["gradnorm_g"] / (mean + 1e-8), rtol=1e-5, atol=1e-8)


def test_eight_rank_bucket_order_and_reducer_modes(eight_rank_results):
    rs = eight_rank_results["reducer"]
    tri = 36.0                                         # sum of (rank + 1) over 8 ranks
    n = rs[0]["a_sum"].numel()
    for r in rs:
        assert torch.equal(r["a_sum"], torch.arange(n, dtype=torch.float32) * tri)
        for it in range(2):
            want_b = torch.cat([torch.zeros(sz) if k == 2 else torch.full((sz,), tri + 8.0 * (it + k)) for k, sz in enumerate((6, 9, 8, 5))])
            want_d = torch.cat([torch.full((sz,), tri + 8.0 * (it + k)) for k, sz in enumerate((6, 9, 8, 5))])
            assert torch.equal(r[f"b_sum{it}"], want_b) and torch.equal(r[f"d_sum{it}"], want_d), (r["rank"], it)
            # rank 0 reports 3, 2, 1, 0 and the seven others 0, 1, 2, 3: every rank still launches buckets 0, 1, 2, 3
            assert r["f_logs"][it] == [0, 1, 2, 3], (r["rank"], r["f_logs"])
            want_f = torch.cat([torch.full((sz,), tri * (k + 1) + 8.0 * it) for k, sz in enumerate((3, 5, 7, 9))])
            assert torch.equal(r[f"f_sum{it}"], want_f), (r["rank"], it)
        assert r["e_raised"] is True and r["b_handles_left"] == 0 and r["off_enabled"] is False


class _StopAfterDiscriminatorBackward(Exception):
    pass


def _single_process_reference(gan, emu_library):
    """One process on the CONCATENATED batch of the two ranks, GradNorm computed the way two data-parallel ranks compute it
    (ops._GradNorm dp_chunks: mean over ranks of the per-rank norms, vae_trainer.py:40-44).  With the GAN branch only the
    discriminator's half of the step is run (its gradients are what the comparison needs; emulator time)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import dist_worker as DW
    import vqgan_training_amd as vq
    from vqgan_training_amd import ops
    from oracle import weights as W
    vq._lib._set_library_for_tests(emu_library)
    ops.clear_caches()
    prev = ops.default_precision()
    ops.set_default_precision("fp32x3")
    try:
        res, ch, vae, lp, disc = DW.build_models(gan)
        grads, d_grads = {}, {}
        step = vq.vae_trainer.VAETrainStep(
            vae, lp, disc, do_ganloss=gan, disc_type="hinge", learning_rate_vae=1e-2, vae_ch=ch, max_steps=10, warmup_steps=0,
            gradnorm_dp_chunks=2, on_backward=lambda s: grads.update({n: p.grad.detach().clone() for n, p in vae.named_parameters()}),
            on_d_backward=None)
        xcat = torch.cat([W.image_batch(1, res, seed=50), W.image_batch(1, res, seed=51)], 0)
        if not gan:
            o = step(xcat)
            return grads, d_grads, {k: v.clone() for k, v in vae.state_dict().items()}, o

        def grab_and_stop(s):
            d_grads.update({n: p.grad.detach().clone() for n, p in disc.named_parameters()})
            raise _StopAfterDiscriminatorBackward()
        step.on_d_backward = grab_and_stop
        orig = vq.vae_trainer.gan_disc_loss_device

        def spy(*a, **k):                      # the step's d_loss (it never returns: the hook above ends it)
            out = orig(*a, **k)
            d_grads["__d_loss__"] = float(out[0])
            return out
        vq.vae_trainer.gan_disc_loss_device = spy
        try:
            step(xcat)
        except _StopAfterDiscriminatorBackward:
            pass
        finally:
            vq.vae_trainer.gan_disc_loss_device = orig
        d_loss = d_grads.pop("__d_loss__")
        return grads, d_grads, None, {"d_loss": d_loss}
    finally:
        ops.set_default_precision(prev)
        vq._lib._set_library_for_tests(None)
        ops.clear_caches()


def _max_rel(a, b):
    """Worst per-tensor relative error; tensors whose exact gradient is zero (a conv bias in front of a GroupNorm only sees
    round-off) are compared on the scale of the largest gradient instead."""
    gmax = max(v.abs().max().item() for v in b.values())
    return max(((a[k].double() - b[k].double()).abs().max() / max(b[k].double().abs().max().item(), 1e-3 * gmax)).item() for k in b)


def test_data_parallel_equivalence_with_one_process_on_the_concatenated_batch(two_rank_results, emu_library):
    """every loss is a batch mean and GroupNorm is per sample, so the averaged gradients of two ranks equal the
    gradients of one process on the concatenated batch — up to GradNorm, whose cross-rank semantics (mean of per-rank norms) the
    single process reproduces through dp_chunks.  Same kernels on both sides: agreement to fp32 summation order."""
    r0, r1 = two_rank_results["sync"]
    grads, _, params, _ = _single_process_reference(False, emu_library)
    dp = {k: 0.5 * v for k, v in r0["local_grads"].items()}          # the flat buffers hold the SUM; AdamW applies 1/world
    assert _max_rel(dp, grads) < 2e-4
    # the first AdamW step moves every element by lr * g / (|g| + eps) ~ +-lr: elements whose gradient is round-off may take the
    # other sign, so parameters agree to 2 lr (lr = 1e-2 / 32 for the main group) and almost everywhere far better
    dpar = torch.cat([(r0["params"][k] - params[k]).abs().flatten() for k in params])
    assert dpar.max().item() <= 2.1 * 1e-2 / 32 and (dpar > 1e-6).float().mean().item() < 0.02


def test_two_ranks_with_the_gan_branch(two_rank_results, emu_library):
    """The full step incl. the discriminator on two ranks (D reducer: one exchange in finish(), under the LPIPS forward):
    both replicas of D and of the VAE stay identical, the lecam anchors use rank-averaged logits, and D's averaged gradients
    equal those of one process on the concatenated batch."""
    r0, r1 = two_rank_results["gan"]
    assert r0["d_buckets"] >= 1 and r0["n_buckets"] >= 2
    for k in r0["d_params"]:
        assert torch.equal(r0["d_params"][k], r1["d_params"][k]), k
    for k in r0["params"]:
        assert torch.equal(r0["params"][k], r1["params"][k]), k
    assert torch.equal(r0["lecam_anchor"], r1["lecam_anchor"]) and float(r0["lecam_anchor"].abs().sum()) > 0
    assert all(torch.equal(r0["d_grads"][k], r1["d_grads"][k]) for k in r0["d_grads"])
    _, d_grads, _, o = _single_process_reference(True, emu_library)
    assert _max_rel({k: 0.5 * v for k, v in r0["d_grads"].items()}, d_grads) < 2e-4
    assert abs(0.5 * (r0["d_loss"] + r1["d_loss"]) - float(o["d_loss"])) < 1e-5 * abs(float(o["d_loss"]))


def test_modules_survive_the_references_own_ddp_wrappers(two_rank_results):
    """SURVEY §8(b1) / INTEGRATION.md §1: `DDP(vae)`, `DDP(discriminator)` (vae_trainer.py:438,450) around this package's modules,
    driven by the reference's loop body — vae.module.encoder / .decoder called directly, the discriminator through DDP.forward three
    times per step with backward(retain_graph=True) in between, plain torch.optim.AdamW (tests/ddp_reference_loop.py) — 2 ranks,
    2 steps, different images per rank:
      * no "expected to have finished reduction" error (the launch returned 0);
      * the discriminator's gradients are the MEAN over the ranks of what the bare modules compute locally, its parameters stay
        identical on both ranks;
      * the VAE's gradients stay local — its wrapper never sees a forward, so nothing is exchanged — and the ranks' VAE
        parameters drift apart, as the reference's do.  (The 1-rank hardware twin below compares wrapped and bare runs bit for bit.)"""
    r0, r1 = two_rank_results["ddp"]
    for it in range(2):
        if it == 0:     # (what is evaluated BEFORE the discriminator's first update: that update uses the rank-averaged gradients under DDP)
            for r in (r0, r1):
                for k in ("percep", "d_loss"):
                    assert abs(r["ddp"][0][k] - r["bare"][0][k]) <= 1e-6 * max(1.0, abs(r["bare"][0][k])), k
        # D: averaged across ranks by DDP's hooks (first step: identical parameters everywhere, so the comparison is exact to round-off)
        if it == 0:
            for k in r0["ddp"][0]["d_grads"]:
                mean = 0.5 * (r0["bare"][0]["d_grads"][k] + r1["bare"][0]["d_grads"][k])
                scale = mean.abs().max().item() + 1e-12
                for r in (r0, r1):
                    assert (r["ddp"][0]["d_grads"][k] - mean).abs().max().item() <= 1e-5 * scale, k
            # VAE: local gradients only (its wrapper never sees a forward): the two ranks' gradients differ (different images, nothing exchanged)
            assert any(not torch.equal(r0["ddp"][0]["g_grads"][k], r1["ddp"][0]["g_grads"][k]) for k in r0["ddp"][0]["g_grads"])
        assert any((r0["ddp"][it]["d_grads"][k] != 0).any() for k in r0["ddp"][it]["d_grads"])
    for k in r0["ddp_d_params"]:
        assert torch.equal(r0["ddp_d_params"][k], r1["ddp_d_params"][k]), k          # DDP kept the discriminators in lock-step
    assert any(not torch.equal(r0["bare_d_params"][k], r1["bare_d_params"][k]) for k in r0["bare_d_params"])
    assert any(not torch.equal(r0["ddp_vae_params"][k], r1["ddp_vae_params"][k]) for k in r0["ddp_vae_params"])   # F2
    assert r0["ddp"][0]["avg_real"] == r1["ddp"][0]["avg_real"]                      # avg_scalar_over_nodes


@pytest.mark.gpu
def test_reference_ddp_wrappers_on_hardware():
    """The same loop body on the MI355X with `DDP(module, device_ids=[0])` over a 1-rank RCCL group (what a test box offers), in the
    timed `ref` arithmetic: two steps wrapped == two steps bare, bit for bit (an all-reduce over one rank is the identity)."""
    import torch.distributed as dist
    from torch.nn.parallel import DistributedDataParallel as DDP
    import vqgan_training_amd as vq
    from vqgan_training_amd import ops
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import ddp_reference_loop as R
    from oracle import weights as W
    dev = torch.device("cuda:0")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29563")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        runs = {}
        for tag in ("ddp", "bare"):
            ops.clear_caches()
            vae, lp, disc = R.build(dev, res=32)
            vq.vae_trainer.apply_precision_policy("ref", vae, lp, disc)
            vae_w, disc_w = (DDP(vae, device_ids=[0]), DDP(disc, device_ids=[0])) if tag == "ddp" else (vae, disc)
            opt_g, opt_d = R.optimizers(vae_w, disc_w)
            runs[tag] = [R.reference_step(vae_w, disc_w, lp, opt_g, opt_d, W.image_batch(2, 32, seed=90 + it).to(dev)) for it in range(2)]
            runs[tag + "_p"] = torch.cat([p.detach().flatten() for p in list(vae.parameters()) + list(disc.parameters())])
    finally:
        dist.destroy_process_group()
        ops.clear_caches()
    for it in range(2):
        for k in ("overall", "percep", "g_gan", "d_loss"):
            assert runs["ddp"][it][k] == runs["bare"][it][k], (it, k)
        for kind in ("d_grads", "g_grads"):
            for k, v in runs["bare"][it][kind].items():
                assert torch.equal(runs["ddp"][it][kind][k], v), (it, kind, k)
    assert torch.equal(runs["ddp_p"], runs["bare_p"])


@pytest.mark.gpu
def test_rccl_single_rank_path_on_hardware():
    """The multi-GPU machinery on real silicon with the one GPU a test box has: a 1-rank RCCL ("nccl") process group with the
    bucketed reducers forced on — gradient-ready callbacks from the kernels' gradient sinks on the autograd thread, in-place async
    all-reduces of flat-buffer slices on the communicator's stream, D's all-reduce under the LPIPS forward, the waits before the
    fused AdamW.  An all-reduce over one rank is the identity, so two steps must reproduce the non-distributed run bit for bit."""
    import os
    import torch.distributed as dist
    import vqgan_training_amd as vq
    from vqgan_training_amd import ops
    dev = torch.device("cuda:0")
    ops.set_default_precision("bf16")

    def run(distributed):
        torch.manual_seed(42)
        ops.clear_caches()
        vae = vq.ae.VAE(64, 3, 64, 3, [1, 2, 4], 2, 8, False, False, False).to(dev)
        disc = vq.utils.PatchDiscriminator().to(dev)
        lp = vq.utils.LPIPS(pretrained_path=None).to(dev)
        step = vq.vae_trainer.VAETrainStep(vae, lp, disc, do_ganloss=True, disc_type="hinge", vae_ch=64, bucket_bytes=1 << 20,
                                           single_rank_collectives=distributed)
        if distributed:
            assert step.reducer_G.enabled and step.reducer_D.enabled and len(step.reducer_G.buckets) > 3
        gen = torch.Generator(device=dev).manual_seed(1)
        losses = [float(step(vq.vae_trainer.synthetic_batch(4, 64, dev, gen))["overall_vae_loss"]) for _ in range(2)]
        return losses, torch.cat([p.detach().flatten() for p in vae.parameters()])

    ref_losses, ref_params = run(False)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29561")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        losses, params = run(True)
    finally:
        dist.destroy_process_group()
    assert losses == ref_losses
    assert torch.equal(params, ref_params)
