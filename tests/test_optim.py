"""FusedAdamW (vq_adamw_multi: one launch per param group on flat fp32 buffers, 16 bytes per lane) against torch.optim.AdamW —
the optimizer the reference builds (vae_trainer.py:455-475: two param groups, weight decay 1e-3, betas (0.9, 0.95)) — over several
steps with a changing learning rate (the cosine schedule rewrites param_groups[i]["lr"] every step, vae_trainer.py:486-490,704)."""
import pytest
import torch

import vqgan_training_amd as vq
from oracle import weights as W


@pytest.mark.parametrize("grad_scale", [1.0, 0.5])
def test_fused_adamw_matches_torch_adamw(backend, grad_scale):
    dev = backend.device
    shapes = [(64, 3, 3, 3), (7,), (33, 5), (1,), (128, 64, 3, 3), (3,), (255,), (4, 4, 4, 4)]     # odd sizes: unaligned chunk tails
    ref_p = [W.uniform_tensor(s, 40 + i).requires_grad_() for i, s in enumerate(shapes)]
    hip_p = [torch.nn.Parameter(p.detach().clone().to(dev)) for p in ref_p]
    groups = lambda ps: [{"params": ps[:5], "lr": 3e-3}, {"params": ps[5:], "lr": 1e-3}]           # noqa: E731
    kw = dict(weight_decay=1e-3, betas=(0.9, 0.95))
    ref = torch.optim.AdamW(groups(ref_p), **kw)
    opt = vq.optim.FusedAdamW(groups(hip_p), grad_scale=grad_scale, **kw)
    for step in range(4):
        for g_ref, g_hip in zip(ref.param_groups, opt.param_groups):        # a schedule: new learning rates every step
            g_ref["lr"] = g_hip["lr"] = g_ref["lr"] * (0.7 + 0.2 * step)
        for i, (a, b) in enumerate(zip(ref_p, hip_p)):
            g = W.uniform_tensor(tuple(a.shape), 1000 * step + i) * (10.0 ** (i % 3 - 1))
            a.grad = g * grad_scale                                           # what the kernel sees after its grad_scale factor
            b.grad.copy_(g.to(dev))                                           # gradients live in the flat buffer (views)
        ref.step()
        opt.step()
        opt.zero_grad()
        for a, b in zip(ref_p, hip_p):
            assert b.grad is not None and float(b.grad.abs().max()) == 0.0
            err = (b.detach().cpu() - a.detach()).abs().max().item()
            assert err <= 2e-6 * max(1.0, a.detach().abs().max().item()), (step, tuple(a.shape), err)
    # moments too: a fifth step from the same state must still agree (m, v enter it)
    for a, b in zip(ref_p, hip_p):
        a.grad = torch.ones_like(a) * grad_scale
        b.grad.fill_(1.0)
    ref.step(); opt.step()
    for a, b in zip(ref_p, hip_p):
        assert (b.detach().cpu() - a.detach()).abs().max().item() <= 2e-6 * max(1.0, a.detach().abs().max().item())
