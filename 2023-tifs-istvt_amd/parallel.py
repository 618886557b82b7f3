"""Data parallelism for the ISTVT path: one process per GPU, clips sharded over the batch axis,
ONE all-reduce (RCCL over xGMI when the backend is "nccl", gloo on CPU for tests) of a flat
gradient bucket per step.

The reference's only multi-GPU mode is single-process nn.DataParallel (train_CNN.py:185-186):
per-replica BatchNorm statistics, gradients summed onto device 0.  The equivalent here keeps
per-rank BN statistics (no SyncBN, as the reference) and averages gradients: the loss is a
batch mean, so mean-of-rank-means == global mean for equal shards.

Only parameters that can receive a gradient go in the bucket: the Xception wrapper holds 19.7 M
parameters (block4..fc) that ``low_level_features`` never touches.
"""
from __future__ import annotations

from typing import Iterable, List, Tuple

import torch
import torch.distributed as dist

from . import stem as _stem


def live_named_parameters(model: torch.nn.Module) -> List[Tuple[str, torch.nn.Parameter]]:
    """Parameters on the hot path (everything except the never-executed Xception tail)."""
    keep = set('xcep.model.' + n for n in _stem.param_names())
    out = []
    for name, p in model.named_parameters():
        if not p.requires_grad:
            continue
        if name.startswith('xcep.') and name not in keep:
            continue
        out.append((name, p))
    return out


class GradBucket:
    """Flat fp32 gradient buffer; every live parameter's ``.grad`` is a view into it."""

    def __init__(self, params: Iterable[torch.nn.Parameter], fuse_accumulate: bool = False):
        """fuse_accumulate: let the HIP weight-gradient kernels add straight into the bucket
        (functional._target) instead of returning fresh tensors for autograd to add."""
        self.params = list(params)
        for p in self.params:
            p._istvt_fused_grad = bool(fuse_accumulate)
        if not self.params:
            raise ValueError('GradBucket needs at least one parameter')
        dev = self.params[0].device
        self.numel = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(self.numel, dtype=torch.float32, device=dev)
        off = 0
        for p in self.params:
            n = p.numel()
            p.grad = self.flat[off:off + n].view_as(p)
            off += n

    def zero(self):
        self.flat.zero_()

    def all_reduce(self, group=None, chunks: int = 1):
        """sum over ranks then divide by world size (== DataParallel's global-batch mean)."""
        if not (dist.is_available() and dist.is_initialized()):
            return
        world = dist.get_world_size(group)
        if world == 1:
            return
        if chunks <= 1:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
        else:
            works = [dist.all_reduce(c, op=dist.ReduceOp.SUM, group=group, async_op=True)
                     for c in self.flat.chunk(chunks)]
            for w in works:
                w.wait()
        self.flat.mul_(1.0 / world)


def broadcast_parameters(model: torch.nn.Module, src: int = 0, group=None):
    """identical weights/buffers on every rank (DataParallel replicates module 0 each step)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    for t in list(model.parameters()) + list(model.buffers()):
        dist.broadcast(t.data, src, group=group)


def shard_batch(x: torch.Tensor, rank: int, world: int) -> torch.Tensor:
    """rank r gets clips [r*B/W, (r+1)*B/W)."""
    b = x.shape[0]
    if b % world != 0:
        raise ValueError('global batch %d is not divisible by world size %d' % (b, world))
    per = b // world
    return x[rank * per:(rank + 1) * per]
