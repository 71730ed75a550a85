"""Worker for tests/test_distributed.py: one process per rank (gloo, CPU tensors, kernels through the
host emulator).  Launched with torch.distributed.run; writes per-rank results into $VQ_DIST_OUT."""
import os
import sys
import warnings

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
warnings.filterwarnings("ignore")
import vqgan_training_amd as vq  # noqa: E402
from vqgan_training_amd import ops  # noqa: E402
from oracle import weights as W  # noqa: E402


def build_models(gan):
    """Same seeded weights in the workers and in the single-process reference of tests/test_distributed.py."""
    res, ch = 16, 32
    vae = vq.ae.VAE(res, 3, ch, 3, [1, 2], 1, 4, False, False, False)
    vae.load_state_dict(W.randomize_state_dict(vae.state_dict(), 1))
    lp = vq.utils.LPIPS(pretrained_path=None)
    lp.load_state_dict(W.randomize_state_dict(lp.state_dict(), 2, relu_net=True))
    lp.eval()
    disc = None
    if gan:
        disc = vq.utils.PatchDiscriminator()
        disc.load_state_dict(W.randomize_state_dict(disc.state_dict(), 4, relu_net=True))
    return res, ch, vae, lp, disc


def run_mode(mode, out_dir, rank, world):
    ops.clear_caches()
    gan = mode == "gan"                    # the full step with the discriminator: both reducers live
    res, ch, vae, lp, disc = build_models(gan)
    vq.distributed.broadcast_parameters(vae)
    if gan:
        vq.distributed.broadcast_parameters(disc)
    grads, d_grads = {}, {}

    def grab(st_):
        if not grads:
            grads.update({n: p.grad.detach().clone() for n, p in vae.named_parameters()})

    def grab_d(st_):
        d_grads.update({n: p.grad.detach().clone() for n, p in disc.named_parameters()})

    step = vq.vae_trainer.VAETrainStep(vae, lp, disc, do_ganloss=gan, disc_type="hinge", learning_rate_vae=1e-2, vae_ch=ch,
                                       max_steps=10, warmup_steps=0, sync_vae_grads=(mode != "nosync"), bucket_bytes=64 << 10,
                                       on_backward=grab, on_d_backward=grab_d if gan else None)
    x = W.image_batch(1, res, seed=50 + rank)
    # GradNorm probe: mean over ranks of the per-rank norms (vae_trainer.py:40-44)
    g = W.uniform_tensor((1, 3, 4, 4), 70 + rank)
    probe = torch.zeros(1, 3, 4, 4, requires_grad=True)
    ops.gradnorm(probe, 1.0).backward(g)
    o = step(x)
    # after finish() the flat gradient buffers were zeroed by zero_grad; capture the post-step parameters (no warm-up: with
    # one the schedule gives step 0 a learning rate of zero, vae_trainer.py:486-490, and a second step would be needed)
    o2 = o
    torch.save({"rank": rank, "world": world, "local_grads": grads, "params": {k: v.clone() for k, v in vae.state_dict().items()},
                "loss0": float(o["overall_vae_loss"]), "loss1": float(o2["overall_vae_loss"]),
                "gradnorm_probe": probe.grad.clone(), "gradnorm_g": g, "n_buckets": len(step.reducer_G.buckets),
                "grad_scale": step.optimizer_G.grad_scale, "d_grads": d_grads,
                "d_params": {} if disc is None else {k: v.clone() for k, v in disc.state_dict().items()},
                "d_loss": float(o["d_loss"]) if gan else None, "lecam_anchor": step.lecam_anchor.clone(),
                "d_buckets": len(step.reducer_D.buckets) if gan else 0,
                # vae_trainer.py:56-60: all_reduce(AVG) of a python float, returned as a float on every rank
                "avg_scalar": vq.vae_trainer.avg_scalar_over_nodes(float(3 * rank + 1), torch.device("cpu"))},
               os.path.join(out_dir, f"rank{rank}_{mode}.pt"))
    dist.barrier()


def run_reducer_only(out_dir, rank, world):
    """BucketedGradReducer alone on plain CPU tensors (no kernels): the discriminator's mode (overlap=False: every bucket
    launched by start()), the autograd-hook mode, a parameter that receives no gradient (its bucket is launched by start() so
    the ranks stay in lock-step), and a second backward after finish() re-arming the countdowns."""
    from types import SimpleNamespace
    from vqgan_training_amd.distributed import BucketedGradReducer

    def flat_group(shapes):
        params = [torch.nn.Parameter(torch.zeros(s)) for s in shapes]
        numel = sum(p.numel() for p in params)
        flat_g = torch.zeros(numel)
        offsets, off = [], 0
        for p in params:
            p.grad = flat_g[off:off + p.numel()].view(p.shape)
            offsets.append(off)
            off += p.numel()
        return SimpleNamespace(params=params, numel=numel, flat_g=flat_g, offsets=offsets)

    res = {}
    # (a) overlap=False: gradients written behind autograd's back (as the discriminator's two passes do), one exchange in finish()
    fg = flat_group([(5, 3), (7,), (4, 4), (9,)])
    red = BucketedGradReducer([fg], bucket_bytes=64, overlap=False)
    fg.flat_g.copy_(torch.arange(fg.numel, dtype=torch.float32) * (rank + 1))
    red.start()
    red.finish()
    res["a_sum"] = fg.flat_g.clone()
    res["a_buckets"] = len(red.buckets)
    # (b) overlap=True through autograd's post-accumulate hooks; parameter 2 takes no part in the loss
    fg = flat_group([(6,), (3, 3), (8,), (5,)])
    red = BucketedGradReducer([fg], bucket_bytes=32, overlap=True)
    for it in range(2):
        fg.flat_g.zero_()
        w = [torch.full_like(p, float(rank + 1 + it + k)) for k, p in enumerate(fg.params)]
        loss = sum((p * wk).sum() for k, (p, wk) in enumerate(zip(fg.params, w)) if k != 2)
        loss.backward()
        red.finish()
        res[f"b_sum{it}"] = fg.flat_g.clone()
    res["b_buckets"] = len(red.buckets)
    res["b_handles_left"] = len(red._handles)
    red.remove()
    # (c) enabled=False (VAETrainStep(sync_vae_grads=False) == the reference's unarmed DDP wrapper): inert
    fg = flat_group([(5, 3), (7,)])
    red = BucketedGradReducer([fg], bucket_bytes=64, enabled=False)
    fg.flat_g.copy_(torch.arange(fg.numel, dtype=torch.float32) * (rank + 1))
    red.start(); red.finish()
    res.update(off_enabled=red.enabled, off_buckets=len(red.buckets), off_grad_scale=red.grad_scale(), off_grads=fg.flat_g.clone())
    # (d) the two report channels together: parameters 0 and 1 are "sink" parameters (a kernel writes their gradient behind
    # autograd's back and fires the ops callback), 2 and 3 arrive through autograd; re-registering a sink (FlatGroup.rebind_grads)
    # must keep the reducer's callback
    fg = flat_group([(6,), (3, 3), (8,), (5,)])
    for p in fg.params[:2]:
        ops.register_grad_sink(p, p.grad)
    red = BucketedGradReducer([fg], bucket_bytes=32, overlap=True)
    ops.register_grad_sink(fg.params[0], fg.params[0].grad)                 # rebind: callback must survive
    res["d_callback_kept"] = ops._sink_of(fg.params[0])[1] is not None
    for it in range(2):
        fg.flat_g.zero_()
        loss = sum((p * float(rank + 1 + it + k)).sum() for k, p in enumerate(fg.params) if k >= 2)
        loss.backward()                                                       # 2, 3 through the post-accumulate hook
        for k in (1, 0):                                                      # 1, 0 as the kernels do it
            view, cb = ops._sink_of(fg.params[k])
            view.add_(float(rank + 1 + it + k))
            cb()
        res[f"d_launched_before_finish{it}"] = len(red._handles)
        red.finish()
        res[f"d_sum{it}"] = fg.flat_g.clone()
    red.remove()
    ops.unregister_grad_sinks([p.data_ptr() for p in fg.params])
    # (e) a SECOND contribution to a parameter whose bucket is already on the wire is an error, not a silent race
    fg = flat_group([(4,), (4,)])
    red = BucketedGradReducer([fg], bucket_bytes=16, overlap=True)            # one parameter per bucket
    (fg.params[1] * 2.0).sum().backward()                                     # bucket of parameter 1 launched
    try:
        (fg.params[1] * 3.0).sum().backward()
        res["e_raised"] = False
    except RuntimeError as e:
        res["e_raised"] = "second gradient contribution" in str(e)
    red.finish(); red.remove()
    # (f) the ranks report their gradients in DIFFERENT host orders (rank 0: the backward order 3, 2, 1, 0; rank 1: 0, 1, 2, 3 — as if
    # its autograd ready queue, or its sink callbacks vs hooks, were served the other way round): collectives of one communicator are
    # matched by issue order, so the buckets must still go on the wire in the same order on both — index order — and sum correctly
    fg = flat_group([(3,), (5,), (7,), (9,)])
    for p in fg.params:
        ops.register_grad_sink(p, p.grad)
    red = BucketedGradReducer([fg], bucket_bytes=8, overlap=True)             # one parameter per bucket: bucket k <-> parameter 3 - k
    order = (3, 2, 1, 0) if rank == 0 else (0, 1, 2, 3)
    logs = []
    for it in range(2):
        fg.flat_g.zero_()
        for k in order:
            view, cb = ops._sink_of(fg.params[k])
            view.add_(float((rank + 1) * (k + 1) + it))
            cb()
        logs.append(list(red.launch_log))
        red.finish()
        res[f"f_sum{it}"] = fg.flat_g.clone()
    res["f_logs"] = logs
    res["f_buckets"] = len(red.buckets)
    red.remove()
    ops.unregister_grad_sinks([p.data_ptr() for p in fg.params])
    # (g) broadcast_parameters: coalesced (one flat buffer per dtype and <= bucket_bytes), every rank ends with rank 0's tensors —
    # parameters AND buffers, mixed dtypes, a bucket boundary inside the list
    torch.manual_seed(100 + rank)                                             # (different weights per rank before the broadcast)
    m = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3), torch.nn.BatchNorm2d(8), torch.nn.Conv2d(8, 4, 1))
    m[1].num_batches_tracked += 5 + rank                                      # an int64 buffer
    res["g_collectives"] = vq.distributed.broadcast_parameters(m, bucket_bytes=600)
    res["g_state"] = {k: v.clone() for k, v in m.state_dict().items()}
    res["g_n_tensors"] = len(list(m.parameters())) + len(list(m.buffers()))
    torch.save({"rank": rank, **res}, os.path.join(out_dir, f"rank{rank}_reducer.pt"))
    dist.barrier()


def run_reference_ddp(out_dir, rank, world):
    """the package's modules under the reference's OWN wrappers and loop body (tests/ddp_reference_loop.py) — two steps
    wrapped in torch DDP, then the same two steps on bare modules with the same weights and batches."""
    from torch.nn.parallel import DistributedDataParallel as DDP
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import ddp_reference_loop as R
    dev = torch.device("cpu")
    xs = [W.image_batch(1, 16, seed=90 + 10 * it + rank) for it in range(2)]     # every rank its own images
    res = {"rank": rank, "world": world}
    for tag in ("ddp", "bare"):
        ops.clear_caches()
        vae, lp, disc = R.build(dev)
        vae_w, disc_w = (DDP(vae), DDP(disc)) if tag == "ddp" else (vae, disc)   # vae_trainer.py:438,450 (device_ids: CPU tensors here)
        opt_g, opt_d = R.optimizers(vae_w, disc_w)
        res[tag] = [R.reference_step(vae_w, disc_w, lp, opt_g, opt_d, x) for x in xs]
        res[tag + "_d_params"] = {k: v.detach().clone() for k, v in disc.state_dict().items()}
        res[tag + "_vae_params"] = {k: v.detach().clone() for k, v in vae.state_dict().items()}
    torch.save(res, os.path.join(out_dir, f"rank{rank}_ddp.pt"))
    dist.barrier()


def main():
    out_dir = os.environ["VQ_DIST_OUT"]
    modes = os.environ.get("VQ_DIST_MODE", "sync").split(",")
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    vq._lib._set_library_for_tests(vq._lib.VqLibrary(os.path.join(ROOT, "tests", "emu", "libvqhip_emu.so")))
    ops.set_default_precision("fp32x3")
    for mode in modes:
        if mode == "reducer":
            run_reducer_only(out_dir, rank, world)
        elif mode == "ddp":
            run_reference_ddp(out_dir, rank, world)
        else:
            run_mode(mode, out_dir, rank, world)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
