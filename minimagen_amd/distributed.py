"""Data-parallel sampling over the GPUs of one node (one process per GPU, RCCL through torch.distributed).

The cascade has no cross-sample coupling (GroupNorm, attention, the dynamic-threshold quantile and the
schedule look-ups are all per sample), so each rank samples a contiguous slice of the batch with ZERO
communication and the only collective is one all_gather of the finished images.
Noise is keyed by the GLOBAL sample index, so the gathered result is bit-identical to a single-GPU run.
"""
from __future__ import annotations

import os
from typing import Optional, Tuple

import torch
import torch.distributed as dist


def _alone(ws: int) -> bool:
    """a group of one rank needs no collective -- unless MINIMAGEN_DIST_SINGLE=1 asks for them anyway (the single-GPU test tier runs the
    RCCL calls of this module that way: same arguments, streams and dtypes as with N ranks)"""
    return ws == 1 and os.environ.get("MINIMAGEN_DIST_SINGLE", "0") != "1"


def shard_bounds(batch: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous slice [lo, hi) of the batch owned by `rank` (first `batch % world_size` ranks get one extra row)."""
    base, rem = divmod(batch, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_samples(local: torch.Tensor, batch: int, group=None) -> torch.Tensor:
    """all_gather of per-rank image slices (possibly ragged by one row) into the full (batch, C, H, W) tensor."""
    ws = dist.get_world_size(group)
    if _alone(ws):
        return local
    sizes = [shard_bounds(batch, ws, r) for r in range(ws)]
    maxn = max(hi - lo for lo, hi in sizes)
    pad = local
    if local.shape[0] < maxn:
        pad = torch.cat((local, local.new_zeros(maxn - local.shape[0], *local.shape[1:])), 0)
    pad = pad.contiguous()
    if all(hi - lo == maxn for lo, hi in sizes):
        # even shards (the benchmark case): ONE collective straight into the preallocated (batch, C, H, W) result, no list + cat copy
        full = torch.empty(batch, *local.shape[1:], dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(full, pad, group=group)
        return full
    flat = torch.empty(ws * maxn, *local.shape[1:], dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(flat, pad, group=group)
    return torch.cat([flat[r * maxn:r * maxn + (hi - lo)] for r, (lo, hi) in enumerate(sizes)], 0)


def sample_distributed(imagen, *, text_embeds: torch.Tensor, text_masks: Optional[torch.Tensor] = None, gather: bool = True,
                       group=None, **sample_kwargs) -> torch.Tensor:
    """Every rank passes the SAME full-batch ``text_embeds``/``text_masks``; rank r samples rows shard_bounds(B, N, r)
    and (if ``gather``) all ranks return the full batch."""
    ws = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    batch = text_embeds.shape[0]
    lo, hi = shard_bounds(batch, ws, rank)
    seed_off = sample_kwargs.pop("_sample_offset", 0)
    if hi == lo:
        # more ranks than samples: this rank has nothing to sample but must still take part in the collective (a rank that raised
        # or returned early would leave the others hanging in all_gather)
        dev = next(imagen.parameters()).device
        size = imagen.image_sizes[-1]
        local = torch.zeros(0, imagen.channels, size, size, dtype=torch.float32, device=dev)
        return gather_samples(local, batch, group) if (gather and not _alone(ws)) else local
    local = imagen.sample(text_embeds=text_embeds[lo:hi].contiguous(),
                          text_masks=None if text_masks is None else text_masks[lo:hi].contiguous(),
                          _sample_offset=seed_off + lo, **sample_kwargs)
    if not gather or _alone(ws):
        return local
    return gather_samples(local, batch, group)


def allreduce_gradients(params, *, bucket_mb: float = 64.0, group=None, average: bool = True):
    """Data-parallel training (the reference trains on one device, train.py:99-103): sum the ranks' parameter
    gradients after ``loss.backward()``.  Gradients are copied into flat fp32 buckets of ``bucket_mb`` MiB (xGMI is point-to-point: a ring
    all-reduce is bound per link, so few large collectives, not one per tensor), every bucket's ``all_reduce`` (RCCL) is issued
    asynchronously and the results are scattered back after the last one was enqueued.  Parameters without a gradient on this rank
    contribute zeros (every rank must issue the same collectives).  Returns the number of collectives."""
    ws = dist.get_world_size(group) if dist.is_initialized() else 1
    params = [p for p in params if p.requires_grad]
    if _alone(ws) or not params:
        return 0
    limit = max(1, int(bucket_mb * (1 << 20) / 4))
    buckets, cur, n = [], [], 0
    for p in params:
        if cur and n + p.numel() > limit:
            buckets.append(cur)
            cur, n = [], 0
        cur.append(p)
        n += p.numel()
    buckets.append(cur)
    work = []
    for bucket in buckets:
        dev = bucket[0].device
        flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1).float() for p in bucket])
        assert flat.device == dev
        work.append((bucket, flat, dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group, async_op=True)))
    for bucket, flat, handle in work:
        handle.wait()
        if average:
            flat.div_(ws)
        off = 0
        for p in bucket:
            g = flat[off:off + p.numel()].view_as(p)
            if p.grad is None:
                p.grad = g.clone()
            else:
                p.grad.copy_(g)
            off += p.numel()
    return len(buckets)


class GradientBucketReducer:
    """``allreduce_gradients`` OVERLAPPED with the backward pass: parameters are grouped into flat fp32 buckets in reverse registration order
    (the order in which the backward produces their gradients); a post-accumulate hook on every parameter counts its bucket down, and a
    bucket whose gradients are all there is copied into its flat buffer and all-reduced asynchronously (RCCL on its own stream) while the
    backward continues below it.  Buckets are always launched in index order -- bucket k waits for bucket k - 1 even if it completes first --
    so every rank issues the same sequence of collectives whatever its autograd scheduling does.  ``finish()`` after ``loss.backward()``
    launches what is left (parameters that received no gradient contribute zeros), waits, averages and writes the results back into
    ``p.grad``.  Usage per step: ``loss.backward(); reducer.finish(); optimizer.step()``.  ``with reducer.no_sync():`` skips the reduction for
    gradient-accumulation steps."""

    def __init__(self, params, *, bucket_mb: float = 64.0, group=None, average: bool = True):
        self.group, self.average = group, average
        self.ws = dist.get_world_size(group) if dist.is_initialized() else 1
        self.params = [p for p in params if p.requires_grad]
        limit = max(1, int(bucket_mb * (1 << 20) / 4))
        self.buckets, cur, n = [], [], 0
        for p in reversed(self.params):
            if cur and n + p.numel() > limit:
                self.buckets.append(cur)
                cur, n = [], 0
            cur.append(p)
            n += p.numel()
        if cur:
            self.buckets.append(cur)
        self.where = {p: k for k, b in enumerate(self.buckets) for p in b}
        self.flat = [None] * len(self.buckets)
        self.enabled = True
        self.launched_in_backward = 0
        self._reset()
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self.params] if not _alone(self.ws) else []

    def _reset(self):
        self.missing = [len(b) for b in self.buckets]
        self.seen = set()
        self.next = 0
        self.work = []
        self.dirty = False           # a hook fired while enabled: this step has something to reduce

    def _launch(self, k):
        bucket = self.buckets[k]
        n = sum(p.numel() for p in bucket)
        flat = self.flat[k]
        if flat is None or flat.device != bucket[0].device:
            flat = self.flat[k] = torch.empty(n, dtype=torch.float32, device=bucket[0].device)
        off = 0
        for p in bucket:
            dst = flat[off:off + p.numel()]
            if p.grad is None or p not in self.seen:
                dst.zero_()                      # no gradient on this rank this step: zeros (every rank issues the same collectives)
            else:
                dst.copy_(p.grad.reshape(-1))
            off += p.numel()
        self.work.append((k, dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)))

    def _on_grad(self, p):
        if not self.enabled or p in self.seen:
            return
        self.dirty = True
        self.seen.add(p)
        self.missing[self.where[p]] -= 1
        while self.next < len(self.buckets) and self.missing[self.next] == 0:
            self._launch(self.next)
            self.next += 1
            self.launched_in_backward += 1

    def finish(self, force: bool = False) -> int:
        """-> number of collectives of this step (0 after a ``no_sync`` backward; ``force=True`` reduces even if no hook fired on this rank --
        a rank whose step produced no gradient at all must still take part when the others reduce)"""
        if _alone(self.ws) or not self.enabled or not (self.dirty or force):
            self._reset()
            return 0
        while self.next < len(self.buckets):
            self._launch(self.next)
            self.next += 1
        for k, handle in self.work:
            handle.wait()
            flat = self.flat[k]
            if self.average:
                flat.div_(self.ws)
            off = 0
            for p in self.buckets[k]:
                g = flat[off:off + p.numel()].view_as(p)
                if p.grad is None:
                    p.grad = g.clone()
                else:
                    p.grad.copy_(g)
                off += p.numel()
        n = len(self.work)
        self._reset()
        return n

    def no_sync(self):
        from contextlib import contextmanager

        @contextmanager
        def ctx():
            prev, self.enabled = self.enabled, False
            try:
                yield
            finally:
                self.enabled = prev
                self._reset()
        return ctx()

    def remove(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []
