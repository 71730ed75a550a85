"""VQ codebook quantizer on the HIP path (BASELINE.json north_star / config 5).

Not part of the reference (no quantizer/codebook exists in cloneofsimo/vqgan-training; its
`reg` is the identity DiagonalGaussian).  Semantics are the standard VQGAN ones and are pinned by
oracle/vq_oracle.{c,py}: nearest code under |z|^2 - 2 z.e + |e|^2 in a fixed fp32 evaluation order
(bit-exact indices, lowest index on ties), straight-through output, commitment + codebook loss.
"""
from __future__ import annotations

import torch
from torch import nn

from ._lib import lib, ptr, stream_of, workspace


class _VQLookup(torch.autograd.Function):
    @staticmethod
    def forward(ctx, tokens, codebook, beta, lookup_tokens=None):
        tokens = tokens.contiguous().float()
        cb = codebook.contiguous().float()
        n, d = tokens.shape
        k = cb.shape[0]
        L = lib()
        ws = workspace(tokens.device, L.size("vq_vq_workspace", n, k))
        idx = torch.empty(n, dtype=torch.int64, device=tokens.device)
        zq = torch.empty_like(tokens)
        md = torch.empty(n, dtype=torch.float32, device=tokens.device)
        # `lookup_tokens`: the tokens the nearest-code search READS when they are not the ones the gradient flows through (policy
        # ref_vq: an fp32-class evaluation of the encoder for the integer work, the binary16 one for the gradients); zq = rows of the
        # codebook either way, the losses and the straight-through output are formed with `tokens`
        look = tokens if lookup_tokens is None else lookup_tokens.contiguous().float()
        L.call("vq_vq_nearest_fwd", ptr(look), ptr(cb), n, k, d, ptr(idx), ptr(zq), ptr(md), ptr(ws), ws.numel(),
               stream_of(tokens))
        ctx.save_for_backward(tokens, zq, idx)
        ctx.beta, ctx.k = float(beta), k
        ctx.mark_non_differentiable(idx)
        diff2 = (zq - tokens).pow(2).mean()                  # [n,D] glue on a few hundred KB
        loss = (1.0 + beta) * diff2                          # value of beta*|sg(zq)-z|^2 + |zq-sg(z)|^2
        return zq, loss, idx                                 # zq doubles as the straight-through output

    @staticmethod
    def backward(ctx, g_out, g_loss, _):
        tokens, zq, idx = ctx.saved_tensors
        n, d = tokens.shape
        scale = 2.0 / (n * d)
        diff = zq - tokens
        # straight-through: d out / d z = I ; commitment: beta * 2 (z - zq) / N ; codebook: 2 (zq - z) / N
        gz = g_out - (ctx.beta * scale) * g_loss * diff
        gq = (scale * g_loss * diff).contiguous()
        dcb = torch.zeros(ctx.k, d, dtype=torch.float32, device=tokens.device)
        L = lib()
        ws = workspace(tokens.device, L.size("vq_vq_scatter_workspace", ctx.k, d), slot=2)   # (slot 1 belongs to the side stream)
        L.call("vq_vq_scatter_add", ptr(gq), ptr(idx), n, ctx.k, d, ptr(dcb), ptr(ws), ws.numel(), stream_of(tokens))
        return gz, dcb, None, None


class VectorQuantizer(nn.Module):
    """forward(z [B,D,h,w]) -> (z_q with straight-through gradient, loss, indices [B,h,w])."""

    def __init__(self, n_codes: int = 16384, dim: int = 32, beta: float = 0.25):
        super().__init__()
        self.n_codes, self.dim, self.beta = n_codes, dim, beta
        self.embedding = nn.Embedding(n_codes, dim)
        self.embedding.weight.data.uniform_(-1.0 / n_codes, 1.0 / n_codes)

    def forward(self, z, lookup_from=None):
        b, d, h, w = z.shape
        tokens = z.permute(0, 2, 3, 1).reshape(-1, d)
        look = None if lookup_from is None else lookup_from.detach().permute(0, 2, 3, 1).reshape(-1, d)
        zq, loss, idx = _VQLookup.apply(tokens, self.embedding.weight, self.beta, look)
        return zq.reshape(b, h, w, d).permute(0, 3, 1, 2), loss, idx.reshape(b, h, w)
