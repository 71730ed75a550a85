"""Pins oracle/vq_oracle.c (the definition the HIP lookup is bit-exact against) to arithmetic OUTSIDE this repository.

The reference has no quantizer, so "the oracle defines it" used to mean the C file was compared with nothing
but the kernel that mirrors it.  Here it is checked against
  (1) the mathematical definition in fp64, brute force:  idx_i = argmin_j sum_k (z_ik - e_jk)^2, lowest index on ties;
  (2) the published VQGAN quantizer arithmetic (taming-transformers `VectorQuantizer.forward`):
        d = sum(z^2, 1, keepdim) + sum(e^2, 1) - 2 z @ e^T ;  torch.argmin(d, 1)      in fp32, as the public code runs it;
  (3) the published loss  beta * mean((z_q.detach() - z)^2) + mean((z_q - z.detach())^2)  and its straight-through gradients,
      by fp64 autograd.
On well-separated data (1) and (2) must agree with the oracle EXACTLY.  On adversarial near-tie data fp32 evaluation orders
legitimately disagree with each other; what is asserted there is that every choice of the oracle is an fp64 near-minimiser
(within a few fp32 roundings of the true minimum), and the disagreement rates are reported.
CPU only — this is about the checker, not about the kernel."""
import json
import os

import torch

from oracle import vq_oracle
from oracle import weights as W


SEP = 1e-4          # >= 20 x the fp32 evaluation error of either formula at dim 64


def _brute_fp64(z, cb, chunk=256):
    """argmin_j sum_k (z_ik - e_jk)^2 in fp64 with the difference formed BEFORE squaring (no cancellation) -> (idx, d64 [n, K])."""
    z64, e64 = z.double(), cb.double()
    rows = []
    for i in range(0, z64.shape[0], chunk):
        diff = z64[i:i + chunk, None, :] - e64[None, :, :]
        rows.append((diff * diff).sum(-1))
    d = torch.cat(rows)
    return torch.argmin(d, 1), d


def _published_fp32(z, cb):
    """taming-transformers VectorQuantizer: d = |z|^2 + |e|^2 - 2 z.e^T, argmin over the codes (fp32 tensors, library matmul)."""
    d = (z ** 2).sum(1, keepdim=True) + (cb ** 2).sum(1) - 2.0 * z @ cb.t()
    return torch.argmin(d, 1)


def _margins(d64, z, cb):
    """Gap between the best and the second-best code relative to the magnitude of the terms the fp32 formula cancels
    (|z|^2 + |e|^2): fp32 evaluation error is ~ dim * 2^-24 of that, whatever the distance itself is."""
    two = torch.topk(d64, 2, dim=1, largest=False).values
    scale = (z.double() ** 2).sum(1) + (cb.double() ** 2).sum(1).max()
    return (two[:, 1] - two[:, 0]) / scale


def test_oracle_equals_fp64_bruteforce_and_published_formula_on_separated_data():
    checked = 0
    for n, k, d, seed in ((2000, 1024, 32, 1), (1500, 777, 8, 2), (800, 4096, 32, 3), (512, 64, 4, 4), (300, 1000, 64, 5)):
        z = W.uniform_tensor((n, d), 1000 + seed, -1, 1)
        cb = W.uniform_tensor((k, d), 2000 + seed, -1, 1)
        want, d64 = _brute_fp64(z, cb)
        sep = _margins(d64, z, cb) > SEP
        assert sep.float().mean().item() > 0.97, "the random data is supposed to be well separated"
        got, md = vq_oracle.nearest(z, cb)
        pub = _published_fp32(z, cb)
        assert torch.equal(got[sep], want[sep])
        assert torch.equal(pub[sep], want[sep])
        # the reported minimum distance is the fp32 value of the true squared distance
        true_min = d64.gather(1, got[:, None]).squeeze(1)
        assert ((md.double() - true_min).abs() <= 1e-5 * (1.0 + true_min)).all()
        checked += int(sep.sum())
    assert checked > 4500


def test_exact_ties_resolve_to_the_lowest_index_like_torch_argmin():
    d, k = 32, 300
    cb = W.uniform_tensor((k, d), 11, -1, 1)
    cb[200] = cb[17]
    cb[250] = cb[17]
    cb[40] = cb[39]
    z = torch.cat([cb[17:18], cb[17:18] + 1e-3, cb[39:40], W.uniform_tensor((50, d), 12, -1, 1)])
    got, _ = vq_oracle.nearest(z, cb)
    want, d64 = _brute_fp64(z, cb)             # torch.argmin (CPU): first occurrence of the minimum
    assert got[0].item() == 17 and got[1].item() == 17 and got[2].item() == 39
    assert torch.equal(got[:3], want[:3])
    assert torch.equal(got[_margins(d64, z, cb) > SEP], want[_margins(d64, z, cb) > SEP])


def test_near_ties_oracle_choice_is_always_an_fp64_near_minimiser_and_rates_are_reported(tmp_path):
    """Adversarial tokens: exactly half-way between two codes, and between codes one ulp apart.  fp32 orders disagree there (that is
    why the kernel follows ONE fixed order); each disagreement must be a genuine near-tie."""
    d, k = 32, 2048
    cb = W.uniform_tensor((k, d), 21, -1, 1)
    nxt = torch.nextafter(cb[100:140], torch.full((40, d), 2.0))
    cb[140:180] = nxt                                                      # 40 pairs one ulp apart
    a, b = cb[torch.arange(0, 1000, 2)], cb[torch.arange(1, 1000, 2)]
    z = torch.cat([(a + b) / 2, cb[100:140], (cb[100:140] + cb[140:180]) / 2, W.uniform_tensor((500, d), 22, -1, 1)])
    got, _ = vq_oracle.nearest(z, cb)
    want, d64 = _brute_fp64(z, cb)
    pub = _published_fp32(z, cb)
    chosen = d64.gather(1, got[:, None]).squeeze(1)
    best = d64.min(1).values
    scale = (z.double() ** 2).sum(1) + (cb.double() ** 2).sum(1).max()     # magnitude of the terms the fp32 formula cancels
    eps = 2.0 ** -23
    assert ((chosen - best) <= 8 * d * eps * scale).all(), "an oracle choice is further from the fp64 minimum than fp32 round-off allows"
    rates = {"tokens": int(z.shape[0]), "oracle_vs_fp64": float((got != want).float().mean()),
             "published_fp32_vs_fp64": float((pub != want).float().mean()), "oracle_vs_published_fp32": float((got != pub).float().mean())}
    print("VQ near-tie disagreement rates (adversarial data, documented not asserted):", json.dumps(rates))
    out = os.environ.get("VQ_NEAR_TIE_REPORT")
    if out:
        with open(out, "w") as f:
            json.dump(rates, f)
    # the random tail of the batch is well separated: exact there
    sep = _margins(d64, z, cb) > SEP
    assert torch.equal(got[sep], want[sep]) and int(sep.sum()) >= 450


def test_quantizer_losses_and_gradients_match_the_published_formulas_in_fp64():
    b, d, h, w, k, beta = 2, 8, 6, 5, 64, 0.25
    z = W.uniform_tensor((b, d, h, w), 31, -1, 1).requires_grad_()
    cb = W.uniform_tensor((k, d), 32, -1, 1).requires_grad_()
    out, loss, idx = vq_oracle.quantize(z, cb, beta)
    (out * W.uniform_tensor(tuple(out.shape), 33)).sum().backward(retain_graph=True)
    gz_st = z.grad.clone()
    z.grad = None
    loss.backward()
    # fp64 restatement of the published module (taming VectorQuantizer, legacy=False ordering of beta)
    z64 = z.detach().double().requires_grad_()
    e64 = cb.detach().double().requires_grad_()
    tok = z64.permute(0, 2, 3, 1).reshape(-1, d)
    i64, _ = _brute_fp64(tok.detach().float(), e64.detach().float())
    assert torch.equal(i64.reshape(b, h, w), idx)
    zq = e64[i64]
    l64 = beta * ((zq.detach() - tok) ** 2).mean() + ((zq - tok.detach()) ** 2).mean()
    l64.backward()
    assert abs(loss.item() - l64.item()) <= 1e-6 * abs(l64.item())
    assert torch.allclose(z.grad.double(), z64.grad, rtol=1e-5, atol=1e-9)
    assert torch.allclose(cb.grad.double(), e64.grad, rtol=1e-5, atol=1e-9)
    # straight-through: the output's gradient reaches z unchanged and the codebook not at all
    assert torch.equal(gz_st, W.uniform_tensor(tuple(out.shape), 33))
    st = (zq.detach().float().reshape(b, h, w, d).permute(0, 3, 1, 2))
    assert torch.allclose(out.detach(), st, rtol=0, atol=1e-6)
