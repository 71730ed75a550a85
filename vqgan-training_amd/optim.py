"""Fused AdamW over flat parameter / gradient buffers (vae_trainer.py:455-475, 659, 702-704).

Every param group is flattened once: its parameters, gradients and both moments live in four
contiguous fp32 buffers (the nn.Parameters become views), so
  * the update is ONE vq_adamw_multi launch per group (HBM roofline: 28 B/param),
  * the gradient buffer is what the data-parallel reducer all-reduces in place (distributed.py),
  * lr can change every step (cosine schedule) without touching device tables.
torch.optim.Optimizer is subclassed only for its param_groups / LR-scheduler protocol.
"""
from __future__ import annotations

import weakref

import torch

from . import ops
from ._lib import VqAdamTensor, lib, ptr, stream_of

_CHUNK = 65536


class FlatGroup:
    """Parameters of one group re-homed into flat buffers; p.data / p.grad are views."""

    def __init__(self, params):
        self.params = [p for p in params]
        assert self.params, "empty parameter group"
        dev = self.params[0].device
        self.numel = sum(p.numel() for p in self.params)
        self.flat_p = torch.empty(self.numel, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros(self.numel, dtype=torch.float32, device=dev)
        self.flat_m = torch.zeros(self.numel, dtype=torch.float32, device=dev)
        self.flat_v = torch.zeros(self.numel, dtype=torch.float32, device=dev)
        off = 0
        self.offsets = []
        for p in self.params:
            assert p.dtype == torch.float32 and p.device == dev
            n = p.numel()
            self.flat_p[off:off + n].copy_(p.detach().reshape(-1))
            p.data = self.flat_p[off:off + n].view(p.shape)
            p.grad = self.flat_g[off:off + n].view(p.shape)
            ops.register_grad_sink(p, p.grad)      # backward kernels accumulate straight into the flat buffer
            self.offsets.append(off)
            off += n
        # one-entry device table for vq_adamw_multi
        ent = VqAdamTensor(self.flat_p.data_ptr(), self.flat_g.data_ptr(), self.flat_m.data_ptr(),
                           self.flat_v.data_ptr(), self.numel)
        raw = bytes(memoryview(ent))
        self.table = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(dev)
        self.n_chunks = (self.numel + _CHUNK - 1) // _CHUNK
        self.chunk_offsets = torch.tensor([0, self.n_chunks], dtype=torch.int64, device=dev)
        self._ptrs = [p.data_ptr() for p in self.params]
        weakref.finalize(self, ops.unregister_grad_sinks, list(self._ptrs))   # never leave sinks of a freed buffer behind

    def rebind_grads(self):
        for p, off in zip(self.params, self.offsets):
            if p.grad is None or p.grad.data_ptr() != self.flat_g.data_ptr() + 4 * off:
                p.grad = self.flat_g[off:off + p.numel()].view(p.shape)
                ops.register_grad_sink(p, p.grad)


class FusedAdamW(torch.optim.Optimizer):
    """AdamW with torch.optim.AdamW's update rule, executed by the HIP kernel."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, grad_scale=1.0):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.grad_scale = float(grad_scale)
        self._flat = [FlatGroup(g["params"]) for g in self.param_groups]
        self._pack = [ops.PackPlan(f.params) for f in self._flat]      # one re-pack launch per group per step
        self._step = 0
        # device-side skip (include/vqhip.h vq_adamw_multi skip_flags): (int32 tensor, n_flags, stride in elements) — when any
        # flag is non-zero the launch changes nothing.  VAETrainStep points this at the saturation counters of the fp16 stacks
        # whose gradients feed this optimizer: a clipped gradient never reaches the parameters, and no host sync is needed.
        self.skip_flags = None

    def rewind(self, steps: int) -> None:
        """The host learnt (at its logging cadence) that `steps` earlier step() calls were skipped on the device: take them out of
        the bias-correction count, so that Adam's step number is the number of updates actually applied."""
        self._step = max(0, self._step - int(steps))

    def snapshot(self, moments: bool = False):
        """Copies of the groups' flat parameter buffers (for `restore`); moments=True: the AdamW moments and the step count too."""
        ops.join_side_stream()
        if not moments:
            return [f.flat_p.clone() for f in self._flat]
        return {"p": [f.flat_p.clone() for f in self._flat], "m": [f.flat_m.clone() for f in self._flat],
                "v": [f.flat_v.clone() for f in self._flat], "step": self._step}

    @torch.no_grad()
    def restore(self, snap) -> None:
        """Back to the parameters of `snapshot()` — with fresh AdamW state (moments zero, step 0), or with the moments and step count
        a `snapshot(moments=True)` holds; the packed conv operands follow."""
        ops.join_side_stream()
        full = isinstance(snap, dict)
        for i, (f, pack) in enumerate(zip(self._flat, self._pack)):
            f.flat_p.copy_(snap["p"][i] if full else snap[i])
            if full:
                f.flat_m.copy_(snap["m"][i]); f.flat_v.copy_(snap["v"][i])
            else:
                f.flat_m.zero_(); f.flat_v.zero_()
            f.flat_g.zero_()
            ops.bump_generation(f._ptrs)
            pack.run()
        self._step = snap["step"] if full else 0

    def flat_grad_buffers(self):
        return [f.flat_g for f in self._flat]

    @torch.no_grad()
    def step(self, closure=None):
        assert closure is None
        ops.join_side_stream()            # weight gradients are written on a second stream (ops._on_side_stream); no-op when it is idle
        self._step += 1
        L = lib()
        for group, flat, pack in zip(self.param_groups, self._flat, self._pack):
            b1, b2 = group["betas"]
            bc1 = 1.0 - b1 ** self._step
            bc2 = 1.0 - b2 ** self._step
            # 28 B / parameter: reads of p, g, m, v and writes of p, m, v
            sk, nsk, sks = self.skip_flags if self.skip_flags is not None else (None, 0, 1)
            ops._launch("hbm:adamw", 28.0 * flat.numel, lambda: L.call(
                "vq_adamw_multi", ptr(flat.table), ptr(flat.chunk_offsets), 1, flat.n_chunks, _CHUNK, float(group["lr"]),
                float(group["weight_decay"]), float(b1), float(b2), float(group["eps"]), float(bc1), float(bc2), self.grad_scale,
                ptr(sk), int(nsk), int(sks), stream_of(flat.flat_p)))
            ops.bump_generation(flat._ptrs)
            # bf16 GEMM operands of every conv weight of the group, one launch: 4 B read per weight + 2 B per packed copy
            # (the byte count is only needed when a launch hook measures; it is taken AFTER run() has re-built a stale plan)
            if ops._launch_hook is None:
                pack.run()
            else:
                pack.prepare()
                ops._launch("hbm:weight_pack", pack.algorithmic_bytes(), pack.run)

    def zero_grad(self, set_to_none: bool = False):
        """Gradients stay bound to the flat buffer (set_to_none is ignored on purpose)."""
        ops.join_side_stream()
        for flat in self._flat:
            flat.flat_g.zero_()
            flat.rebind_grads()
