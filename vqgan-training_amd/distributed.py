"""Data-parallel gradient exchange: bucketed all-reduce over RCCL/xGMI overlapped with backward.

The reference wraps the VAE in DDP (vae_trainer.py:438) but calls `vae.module.encoder/.reg/.decoder`
(vae_trainer.py:538,563,624), so DDP's reducer is never armed and VAE gradients are NOT averaged
across ranks.  This module implements the intended exchange:

  * gradients live in the optimizer's flat fp32 buffers (optim.FlatGroup) — buckets are contiguous
    slices, reduced IN PLACE (no bucket copies);
  * buckets are cut in reverse parameter order (decoder tail first ~ the order backward produces
    them); a post-accumulate-grad hook per parameter counts readiness and launches
    `dist.all_reduce(bucket, async_op=True)` as soon as a bucket is complete, so the RCCL kernels run
    on the communicator's stream underneath the remaining backward convolutions;
  * buckets go on the wire STRICTLY IN INDEX ORDER on every rank: a bucket that completes early waits for its
    predecessors.  Collectives of one communicator are matched by issue order, and the order in which gradients become
    ready is a property of each rank's host (the autograd engine's ready queue, sink callbacks vs hooks) — launching
    "whichever bucket completes first" could pair bucket 3 of one rank with bucket 2 of another;
  * `finish()` waits for the outstanding handles before the optimizer step; the 1/world averaging is
    folded into the AdamW kernel (`grad_scale`), not a separate pass.

xGMI sizing: 326.6 MB of VAE gradients per step; ring all-reduce is per-link bound
(~1.75*S through one ~76.8 GB/s link direction => ~7.4 ms), far below the backward time, so
~32 MB buckets keep several collectives in flight without fragmenting the links.
`sync_vae_grads=False` reproduces the reference's (unsynchronised) behaviour.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


class BucketedGradReducer:
    def __init__(self, flat_groups, bucket_bytes: int = 32 << 20, group=None, enabled: bool = True, overlap: bool = True,
                 single_rank_ok: bool = False):
        """overlap=False: all buckets are launched in finish() (for modules whose parameters receive several
        gradient contributions per backward, e.g. the discriminator's real + fake passes).
        single_rank_ok=True keeps the whole machinery (hooks, async collectives, waits) alive in a 1-rank process group —
        used to exercise the RCCL path on a single-GPU box (tests/test_distributed.py)."""
        self.overlap = overlap
        self.group = group
        self.world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
        self.enabled = enabled and (self.world > 1 or (single_rank_ok and dist.is_available() and dist.is_initialized()))
        self.buckets = []          # (flat tensor slice, [param indices])
        self._pending = []
        self._handles = []
        self._hooks = []
        self._reported = {}           # id(parameter) -> channel of its first gradient report in the current backward
        self._started = False
        self._ready = []              # bucket complete (all its parameters reported), not necessarily launched yet
        self._next = 0                # first bucket not yet on the wire: launches happen in index order only
        self.launch_log = []          # bucket indices in launch order of the current backward (tests)
        self.last_launch_log = []     # ... of the backward the last finish() closed
        if not self.enabled:
            return
        cap = max(1, bucket_bytes // 4)
        for fg in flat_groups:
            # walk parameters from the last to the first; a bucket is a contiguous range of the flat buffer
            hi = fg.numel
            cur_lo = hi
            members = []
            for idx in range(len(fg.params) - 1, -1, -1):
                lo = fg.offsets[idx]
                members.append(fg.params[idx])
                cur_lo = lo
                if hi - cur_lo >= cap or idx == 0:
                    self._add_bucket(fg.flat_g[cur_lo:hi], members)
                    hi = cur_lo
                    members = []

    def _add_bucket(self, view, members):
        b = len(self.buckets)
        live = [p for p in members if p.requires_grad]
        self.buckets.append((view, len(live)))
        self._pending.append(len(live))
        self._ready.append(False)
        if not self.overlap:
            return
        from . import ops
        for p in live:
            hook = self._make_hook(b)
            # parameters whose gradients are written in place by the kernels (ops gradient sinks) report through the sink
            # callback at write time, gradients that arrive through autograd (the VQ codebook, torch glue) through the
            # post-accumulate hook: BOTH are installed.  (torch >= 2.x also fires the post-accumulate hook of a parameter whose
            # backward returned None — every sink parameter echoes there once its node is done; that echo is not a contribution.)
            ops.set_grad_ready_callback(p, lambda h=hook, q=p: h(q, "sink"))
            self._hooks.append(p.register_post_accumulate_grad_hook(hook))

    def _make_hook(self, b):
        def hook(param, channel="autograd"):
            key = id(param)
            first = self._reported.get(key)
            if first is not None:
                if channel == "autograd" and first == "sink":
                    return                        # the echo of a sink parameter
                # a second contribution to the same parameter in one backward: harmless while its bucket is still waiting for
                # others, wrong once the bucket is on the wire (the all-reduce would race the write)
                if b < self._next:
                    raise RuntimeError("BucketedGradReducer(overlap=True): a parameter received a second gradient contribution after "
                                       "its bucket was launched — build the reducer with overlap=False for modules that are "
                                       "applied more than once per backward")
                return
            self._reported[key] = channel
            self._pending[b] -= 1
            if self._pending[b] == 0:
                self._ready[b] = True
                self._launch_ready()
        return hook

    def _launch_ready(self):
        """Every complete bucket whose predecessors are all on the wire goes out, in index order."""
        while self._next < len(self.buckets) and self._ready[self._next]:
            self._launch(self._next)
            self._next += 1

    def _launch(self, b):
        view, _ = self.buckets[b]
        self.launch_log.append(b)
        if view.is_cuda:                  # weight gradients are written on ops' side stream: the collective must not start before them
            from . import ops
            with ops.collective_after_side_stream(view.device):
                self._handles.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            return
        self._handles.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def grad_scale(self) -> float:
        """Factor the optimizer applies to the summed gradients (mean over ranks)."""
        return 1.0 / self.world if self.enabled else 1.0

    def start(self):
        """Launch every bucket that has not been launched yet (all of them when overlap=False; otherwise those whose
        parameters did not all receive a gradient, so that ranks stay in lock-step).  Returns immediately: the
        collectives run on the communicator's stream under whatever the caller enqueues next."""
        if not self.enabled or self._started:
            return
        for b in range(self._next, len(self.buckets)):    # whatever is not on the wire yet, complete or not, in index order
            self._launch(b)
        self._next = len(self.buckets)
        self._started = True

    def finish(self):
        """Wait for all bucket all-reduces of this backward (start() is implied)."""
        if not self.enabled:
            return
        self.start()
        self._started = False
        for h in self._handles:
            h.wait()
        self._handles = []
        self.last_launch_log, self.launch_log = self.launch_log, []
        self._pending = [n for (_, n) in self.buckets]
        self._ready = [False] * len(self.buckets)
        self._next = 0
        self._reported.clear()

    def remove(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []


def broadcast_parameters(module: torch.nn.Module, src: int = 0, group=None, bucket_bytes: int = 256 << 20):
    """What DDP's constructor does (vae_trainer.py:438,450): make every rank start from rank 0's weights.  Like DDP's
    `_sync_module_states` the tensors travel COALESCED — one flat buffer per dtype and <= 256 MB (the VAE: 224 tensors, 326.6 MB ->
    two broadcasts; the discriminator: one) instead of one collective per tensor (~270 of them, each a launch + a ring set-up on
    xGMI: round-5 verdict, item 7b).  -> number of collectives issued."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return 0
    tensors = [t.data for t in list(module.parameters()) + list(module.buffers())]
    rank = dist.get_rank(group)
    issued = 0
    by_type = {}
    for t in tensors:
        by_type.setdefault((t.dtype, t.device), []).append(t)
    for (_dt, _dev), ts in by_type.items():
        bucket, size = [], 0
        for i, t in enumerate(ts):
            bucket.append(t)
            size += t.numel() * t.element_size()
            if size >= bucket_bytes or i == len(ts) - 1:
                flat = torch.cat([b.reshape(-1) for b in bucket])
                dist.broadcast(flat, src=src, group=group)
                issued += 1
                if rank != src:
                    off = 0
                    for b in bucket:
                        b.copy_(flat[off:off + b.numel()].view_as(b))
                        off += b.numel()
                bucket, size = [], 0
    return issued
