"""bench.py — whole-job throughput of the VAE train step on MI355X (driver contract in the task).

A "step" is one full iteration of the reference's loop body (vae_trainer.py:525-708) on one synthetic
batch already resident in HBM: encoder -> reg -> decoder -> [D(real), D(fake), D loss bwd, D AdamW]
-> GradNorm -> LPIPS -> z-regulariser -> G GAN term -> backward -> bucketed gradient all-reduce ->
fused AdamW (both param groups) -> LR schedule.  Nothing is skipped inside the timed region.

Workload (BASELINE.json): metric "images/sec full train step (enc+dec+LPIPS+disc+bwd), 256x256 f=8";
default = configs[2] "vae_ch=128 ch_mult=1,2,4,4 f=8, batch=16 256x256, LPIPS + PatchDiscriminator +
GradNorm" per GPU (the configuration the metric's "disc" names; `--workload c2` drops the GAN branch =
configs[1]).  N>1: one process per GPU (torchrun), the batch dimension shards data-parallel, weak scaling.

Extra objects on the JSON line:
  roofline     — dominant kernel family (implicit-GEMM conv fwd/dgrad on MFMA): algorithmic FLOPs of
                 every launch in the timed region / their HIP-event durations, vs the dense bf16 MFMA peak;
                 `wgrad` = the same for the weight-gradient family, `conv3x3` = the 3x3 conv GEMMs with >= 64 channels on
                 both sides in all three passes (the sub-metric BASELINE.json's north_star states its 40 % target on).
  cpu_baseline — the oracle's restated reference step (oracle/model_ref.py, plain PyTorch CPU fp32) timed
                 on this box's host cores on a bounded sample (rank 0, N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

PEAK_BF16_TFLOPS = 2500.0     # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c3", choices=["c2", "c3", "c5"],
                    help="BASELINE.json configs[1] / configs[2] (default: the one the metric is quoted on) / configs[4] per-GPU share")
    ap.add_argument("--batch", type=int, default=0, help="per GPU (default: 16; 8 for c5)")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32", "fp32x3"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--conv-table", default="", help="write the per-layer-shape conv timing table to this path")
    ap.add_argument("--cpu-baseline-res", type=int, default=256)
    return ap.parse_args()


class ConvTimer:
    """HIP-event timing of every conv launch on the stream it is launched on (torch's current stream)."""

    def __init__(self):
        self.records = []     # (kind, flops, start_event, end_event)
        self.enabled = False

    def launch(self, kind, flops, fn, tag=""):
        if not self.enabled:
            return fn()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        fn()
        e.record()
        self.records.append((kind, flops, s, e, tag))

    def table(self, steps):
        """Per layer-shape rows: launches/step, ms/step, achieved TFLOP/s — sorted by time."""
        agg = {}
        for kind, flops, s, e, tag in self.records:
            d = agg.setdefault(tag, [0, 0.0, 0.0])
            d[0] += 1
            d[1] += flops
            d[2] += s.elapsed_time(e)
        rows = sorted(agg.items(), key=lambda kv: -kv[1][2])
        tot = sum(v[2] for _, v in rows) or 1.0
        lines = [f"{'layer':46s} {'n/step':>6s} {'ms/step':>8s} {'share':>6s} {'TFLOP/s':>8s}"]
        for tag, (n, fl, ms) in rows:
            lines.append(f"{tag:46s} {n / steps:6.1f} {ms / steps:8.3f} {ms / tot:6.1%} {fl / ms / 1e9:8.1f}")
        return "\n".join(lines)

    def family(self, pred):
        """(launches, FLOPs, seconds) over the records whose (what, cin, cout, k, stride) parsed from the layer tag satisfy
        pred — e.g. the 3x3 convolution GEMMs the north-star target is stated on."""
        import re
        pat = re.compile(r"^(\w+) (\d+)->(\d+) in \d+x\d+x\d+ k(\d+) s(\d+) up")
        n, fl, sec = 0, 0.0, 0.0
        for _kind, flops, s, e, tag in self.records:
            m = pat.match(tag)
            if m and pred(m.group(1), int(m.group(2)), int(m.group(3)), int(m.group(4)), int(m.group(5))):
                n += 1
                fl += flops
                sec += s.elapsed_time(e) * 1e-3
        return n, fl, sec

    def summary(self):
        out = {}
        for kind, flops, s, e, _tag in self.records:
            d = out.setdefault(kind, [0, 0.0, 0.0])
            d[0] += 1
            d[1] += flops
            d[2] += s.elapsed_time(e) * 1e-3
        return out


def cpu_baseline(args, cfg):
    """Oracle = restated reference step on CPU fp32 (kind 'port'); bounded to ~one step at B=1."""
    from oracle import model_ref as M
    import vqgan_training_amd as vq
    torch.manual_seed(42)
    res = args.cpu_baseline_res
    vae = vq.ae.VAE(res, 3, cfg["ch"], 3, list(cfg["ch_mult"]), 2, cfg["z"], False, False, False)
    lp = vq.utils.LPIPS(pretrained_path=None)
    disc = vq.utils.PatchDiscriminator() if cfg["gan"] else None
    st = M.RefState(vae.state_dict(), lp.state_dict(), None if disc is None else disc.state_dict())
    del vae, lp, disc
    kw = dict(do_ganloss=cfg["gan"], disc_type="hinge", learning_rate_vae=1e-5, vae_ch=cfg["ch"], max_steps=1000)
    # PyTorch's CPU convolutions stop scaling (and collapse when oversubscribed) long before the 256
    # hardware threads of the GPU box: try a few intra-op thread counts, keep the fastest.
    ncpu = os.cpu_count() or 1
    best = None
    for thr in sorted({t for t in (8, 16, 32, 64) if t <= ncpu} or {ncpu}):
        torch.set_num_threads(thr)
        M.train_step_ref(st, torch.rand(1, 3, 64, 64) * 2 - 1, **kw)        # tiny warm-up (thread pool, allocator)
        x = torch.rand(2, 3, res, res) * 2 - 1
        t0 = time.time()
        M.train_step_ref(st, x, **kw)
        dt = time.time() - t0
        if best is None or dt < best[0]:
            best = (dt, thr)
        if dt > 40:
            break
    dt, thr = best
    return {"value": round(2.0 / dt, 4), "unit": "images/sec", "cores": thr, "kind": "port",
            "sample": f"1 full train step of the restated reference loop (oracle/model_ref.py) at batch 2, {res}x{res}, "
                      f"same model config, CPU fp32, best of thread counts 8..64: {dt:.1f} s on {thr} threads "
                      f"({ncpu} logical CPUs on the box)"}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs"
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world} (launch with torch.distributed.run)"
    device = torch.device(f"cuda:{local_rank}")
    torch.cuda.set_device(device)
    if world > 1:
        dist.init_process_group("nccl", device_id=device)

    import vqgan_training_amd as vq
    from vqgan_training_amd import ops
    vq._lib.lib()                                         # fail loudly if libvqhip.so is missing
    if os.environ.get("VQ_TILE"):                         # A/B knob for kernel experiments (tools/): never set by the driver
        vq._lib.lib().dll.vq_debug_set_conv_tile(int(os.environ["VQ_TILE"]))
    ops.set_default_precision(args.precision)
    cfg = {"ch": 128, "ch_mult": (1, 2, 4, 4), "z": 16, "res": 256, "gan": args.workload == "c3", "vq": None}
    if args.workload == "c5":   # configs[4]: VQ codebook 16384 x 32, 512x512, f=16 (ch=128 assumed), full loss
        cfg = {"ch": 128, "ch_mult": (1, 2, 4, 4, 4), "z": 32, "res": 512, "gan": True, "vq": (16384, 32)}
    if not args.batch:
        args.batch = 8 if args.workload == "c5" else 16

    torch.manual_seed(42)                                 # vae_trainer.py:374-378: same seed on every rank
    vae = vq.ae.VAE(cfg["res"], 3, cfg["ch"], 3, list(cfg["ch_mult"]), 2, cfg["z"], False, False, False).to(device)
    disc = vq.utils.PatchDiscriminator().to(device) if cfg["gan"] else None
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        lpips = vq.utils.LPIPS().to(device)               # train mode: Dropout(0.5) live, as in the reference (F3)
    vq.distributed.broadcast_parameters(vae)
    if disc is not None:
        vq.distributed.broadcast_parameters(disc)
    quant = None
    if cfg["vq"]:
        quant = vq.quantizer.VectorQuantizer(cfg["vq"][0], cfg["vq"][1]).to(device)
        vq.distributed.broadcast_parameters(quant)
    step = vq.vae_trainer.VAETrainStep(vae, lpips, disc, do_ganloss=cfg["gan"], disc_type="hinge",
                                       learning_rate_vae=1e-5, vae_ch=cfg["ch"], max_steps=1000, quantizer=quant)
    timer = ConvTimer()
    ops.set_launch_hook(timer.launch)

    gen = torch.Generator(device=device).manual_seed(42 + rank)
    B = args.batch
    batches = [vq.vae_trainer.synthetic_batch(B, cfg["res"], device, gen) for _ in range(4)]   # resident in HBM

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(batches[i % len(batches)])
    barrier()
    timer.enabled = True
    t0 = time.perf_counter()
    last = None
    for i in range(args.steps):
        last = step(batches[i % len(batches)])
    barrier()
    elapsed = time.perf_counter() - t0
    timer.enabled = False
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    loss = float(last["overall_vae_loss"])
    assert loss == loss, "non-finite loss"

    if rank == 0:
        summ = timer.summary()
        if args.conv_table:
            with open(args.conv_table, "w") as f:
                f.write(timer.table(args.steps) + "\n")
        roof = None
        if "conv_igemm" in summ:
            n, fl, sec = summ["conv_igemm"]
            ach = fl / sec / 1e12
            roof = {"bound": "mfma", "kernel": "conv_igemm_kernel (implicit-GEMM conv fwd + dgrad)",
                    "achieved": round(ach, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(ach / PEAK_BF16_TFLOPS, 4), "traffic": None,   # per-layer PMC passes: profiles/r1_pmc_hbm_traffic_v14.txt
                    "launches": n, "avg_launch_ms": round(sec / n * 1e3, 4),
                    "algorithmic_gflop_per_launch": round(fl / n / 1e9, 2),
                    "share_of_step_time": round(sec / elapsed, 3)}
            if "conv_wgrad" in summ:
                n2, fl2, sec2 = summ["conv_wgrad"]
                roof["wgrad"] = {"achieved": round(fl2 / sec2 / 1e12, 2), "launches": n2,
                                 "frac": round(fl2 / sec2 / 1e12 / PEAK_BF16_TFLOPS, 4),
                                 "share_of_step_time": round(sec2 / elapsed, 3)}
            try:   # the sub-metric BASELINE.json's north_star names: the 3x3 conv GEMMs (>= 64 channels both sides), all three passes
                n3, fl3, sec3 = timer.family(lambda what, ci, co, k, st: k == 3 and ci >= 64 and co >= 64)
                if n3 and sec3 > 0:
                    roof["conv3x3"] = {"achieved": round(fl3 / sec3 / 1e12, 2), "frac": round(fl3 / sec3 / 1e12 / PEAK_BF16_TFLOPS, 4),
                                       "launches": n3, "share_of_step_time": round(sec3 / elapsed, 3)}
            except Exception as exc:   # an auxiliary field must never cost the bench line
                roof["conv3x3"] = {"error": repr(exc)}
        ips = args.steps * B * world / elapsed
        line = {
            "metric": ("images/sec full train step (enc+VQ+dec+LPIPS+disc+bwd), 512x512 f=16" if cfg["vq"] else
                       "images/sec full train step (enc+dec+LPIPS+disc+bwd), 256x256 f=8"),
            "value": round(ips, 3), "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": {"bf16": "bf16", "fp32": "bf16 MFMA operands / fp32 storage", "fp32x3": "bf16x3-split (fp32-class)"}[args.precision],
            "data": f"synthetic uniform [-1,1] {cfg['res']}x{cfg['res']} RGB resident in HBM; random-init VAE (seed 42); random-init VGG16 "
                    "weights for LPIPS / PatchDiscriminator (no network for ImageNet weights)",
            "config": {"workload": ("configs[4] (one GPU's share): VQ 16384x32, vae_ch=128 ch_mult=1,2,4,4,4 f=16 z=32, 512x512, LPIPS + PatchDiscriminator(hinge) + GradNorm, full step incl. AdamW"
                                    if cfg["vq"] else "configs[2]: vae_ch=128 ch_mult=1,2,4,4 f=8 z=16, 256x256, LPIPS + PatchDiscriminator(hinge) + GradNorm, full step incl. AdamW"
                                    if cfg["gan"] else
                                    "configs[1]: vae_ch=128 ch_mult=1,2,4,4 f=8 z=16, 256x256, LPIPS only, full step incl. AdamW"),
                       "per_gpu_batch": B, "global_batch": B * world, "parallelism": f"dp{world}", "precision": args.precision,
                       "final_loss": round(loss, 5)},
            "roofline": roof,
        }
        if world == 1 and not args.no_cpu_baseline and not cfg["vq"]:
            line["cpu_baseline"] = cpu_baseline(args, cfg)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
