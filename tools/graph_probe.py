"""Round-6 probe: what would HIP-graph capture of the step buy, and does the path capture at all?

The library never synchronises, never allocates and launches everything with hipLaunchKernel on the caller's stream, so its launches
are capturable by construction.  This probe measures on the GPU box, for configs[2]'s VAE at B = 16 under policy `ref`:
  (1) forward only (encoder + decoder, no autograd): eager vs torch.cuda.CUDAGraph replay;
  (2) forward + backward of the VAE (gradient sinks, weight gradients on the side stream, no optimizer): eager vs replay.
Both report ms per iteration and the number of kernel nodes; a failing capture prints the failing call.  What a captured STEP would
need on top (not built — see DESIGN.md section 6): device-resident learning rate / Adam step count / dropout counter (host scalars
today: baked into a graph), static input / output buffers, and the range-event polls kept outside the graph.
"""
import os
import sys
import time
import traceback

os.environ.setdefault("VQ_SIDE_KEEP_PRUNE", "0")        # (event queries are not allowed while a stream is capturing)
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import vqgan_training_amd as vq  # noqa: E402
from vqgan_training_amd import ops  # noqa: E402


def timeit(fn, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def main():
    dev = torch.device("cuda:0")
    torch.manual_seed(42)
    B = 16
    vae = vq.ae.VAE(256, 3, 128, 3, [1, 2, 4, 4], 2, 16, False, False, False).to(dev)
    vq.vae_trainer.apply_precision_policy("ref", vae, None, None)
    opt = vq.optim.FusedAdamW([{"params": list(vae.parameters())}], lr=1e-5)      # flat buffers + gradient sinks, like the trainer
    x = vq.vae_trainer.synthetic_batch(B, 256, dev, torch.Generator(device=dev).manual_seed(1))
    launches = [0]
    ops.set_launch_hook(lambda kind, flops, fn, tag="": (launches.__setitem__(0, launches[0] + 1), fn())[1])

    def fwd():
        with torch.no_grad():
            recon, z = vae(x)
        return recon

    def fwd_bwd():
        recon, z = vae(x)
        (recon.float().mean() * 1024.0).backward()
        opt.zero_grad()

    side = torch.cuda.Stream(device=dev)
    for name, fn, n in (("forward (no autograd)", fwd, 10), ("forward + backward (no optimizer)", fwd_bwd, 6)):
        with torch.cuda.stream(side):
            for _ in range(3):
                fn()
            launches[0] = 0
            fn()
            per_iter = launches[0]
            eager = timeit(fn, n)
        row = f"{name}: eager {eager:.2f} ms / iteration, {per_iter} library launches"
        try:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side):
                fn()
            replay = timeit(g.replay, n)
            row += f"; graph replay {replay:.2f} ms ({replay / eager - 1:+.1%})"
        except Exception as exc:       # the failing call is the finding
            row += f"; CAPTURE FAILED: {exc!r}\n" + "".join(traceback.format_exc().splitlines(True)[-6:])
            try:
                torch.cuda.synchronize()
            except Exception:
                pass
        print(row, flush=True)


if __name__ == "__main__":
    main()
