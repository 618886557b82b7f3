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

import os
import weakref
from typing import Iterable, List, Tuple

import torch
import torch.distributed as dist

from . import stem as _stem


def _single_rank(group=None) -> bool:
    """True when the collectives can be skipped: a world of one rank.  ISTVT_FORCE_COLLECTIVES=1 runs them anyway -- on a
    one-GPU box that is the only way to drive RCCL through exactly the calls, streams and waits of the N > 1 path
    (`bench.py --rccl-rehearsal`); a one-rank all-reduce leaves the values unchanged."""
    return dist.get_world_size(group) == 1 and os.environ.get('ISTVT_FORCE_COLLECTIVES') != '1'


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

    def __init__(self, params: Iterable[torch.nn.Parameter], fuse_accumulate: bool = False,
                 flatten_params: bool = False):
        """fuse_accumulate: let the HIP weight-gradient kernels add straight into the bucket
        (functional._target) instead of returning fresh tensors for autograd to add.
        flatten_params: also move the parameters themselves into one flat fp32 buffer (``flat_params``; each
        ``p.data`` becomes a view of it, values preserved) -- what FusedSGD / FusedAdamW step over."""
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
        self.defer_scale = False        # set by the fused optimizers: all_reduce() then only sums, the step kernel scales
        self.grad_scale = 1.0           # what the next fused optimizer step multiplies the gradients by (1 / world size)
        self._early = None
        self._early_work = None
        # CUs left to the collective's kernels while the early all-reduce overlaps the stem backward (RCCL runs one
        # workgroup per channel; the persistent GEMM kernels need whole CUs).  ISTVT_RCCL_CU_RESERVE overrides.
        self.cu_reserve = int(os.environ.get('ISTVT_RCCL_CU_RESERVE', '32'))
        self._reserved = None           # (device index, value before ours) while the reserve is raised
        self._on_ready = None
        self.flat_params = None
        if flatten_params:
            if any(p.dtype != torch.float32 for p in self.params):
                raise TypeError('flatten_params needs float32 parameters')
            self.flat_params = torch.empty(self.numel, dtype=torch.float32, device=dev)
            off = 0
            for p in self.params:
                n = p.numel()
                view = self.flat_params[off:off + n].view_as(p)
                view.copy_(p.data)
                p.data = view
                off += n

    def zero(self):
        # weight gradients of a backward pass that aborted may still be queued on the side stream: they must land
        # before the zero fill, not after it; so must an early all-reduce that such a backward started and no
        # all_reduce() call collected (it also still holds the CU reserve)
        from . import functional as Fn
        Fn.flush_stale_joins()
        self._drop_early_work()
        self.flat.zero_()

    def enable_early_all_reduce(self, first_param: int, group=None):
        """Overlap the gradient all-reduce with the stem backward: the gradients of ``params[first_param:]`` (the
        transformer, when the parameters are in model order: stem first) are complete when the backward pass reaches
        the token assembly, long before the stem's are.  Their slice of the bucket is then all-reduced asynchronously
        (functional.grad_ready_hooks); ``all_reduce()`` later waits for it and reduces the rest.  Needs every gradient
        of the slice to be written by the kernels themselves (fuse_accumulate=True): a gradient handed back to autograd
        could still be accumulated after the hook.  No effect without an initialised process group of size > 1."""
        from . import functional as Fn
        if not all(getattr(p, '_istvt_fused_grad', False) for p in self.params[first_param:]):
            raise ValueError('enable_early_all_reduce needs GradBucket(..., fuse_accumulate=True)')
        lo = sum(p.numel() for p in self.params[:first_param])
        self._drop_early_work()
        self.disable_early_all_reduce()
        self._early = (lo, self.numel, group)
        ref = weakref.ref(self)          # the global hook list must not keep a dropped bucket (and its collectives) alive

        def on_ready(device):
            me = ref()
            if me is None:
                if on_ready in Fn.grad_ready_hooks:
                    Fn.grad_ready_hooks.remove(on_ready)
                return
            if not (dist.is_available() and dist.is_initialized()):
                return
            if _single_rank(group) or device != me.flat.device:
                return
            if me._early_work is not None:
                # a second backward before all_reduce() (gradient accumulation): its kernels would add into the slice
                # while the asynchronous collective is still reducing it, and its gradients would never be summed
                raise RuntimeError('GradBucket: early all-reduce still in flight; there must be exactly one backward per '
                                   'all_reduce() (call disable_early_all_reduce() for gradient accumulation)')
            if device.type == 'cuda':
                Fn.join_side_stream(device.index)       # the side-stream weight gradients are part of the slice
            me._early_work = (dist.all_reduce(me.flat[lo:me.numel], op=dist.ReduceOp.SUM, group=group, async_op=True), lo)
            # the collective's kernels hold CUs until it is done: the persistent GEMMs of the stem backward, which runs
            # beside it, are launched on the CUs that are left (ops.set_cu_reserve), reset in all_reduce()
            if device.type == 'cuda' and me.cu_reserve > 0:
                from . import ops
                me._reserved = (device.index, ops.set_cu_reserve(me.cu_reserve, device.index))

        self._on_ready = on_ready
        Fn.grad_ready_hooks.append(on_ready)

    def _release_cus(self):
        """put back the CU reserve that the early all-reduce hook raised (the value found then, not 0)"""
        if self._reserved is not None:
            from . import ops
            idx, prev = self._reserved
            self._reserved = None
            ops.set_cu_reserve(prev, idx)

    def _drop_early_work(self):
        """an early collective nobody will wait for through all_reduce() (re-arming, a backward that aborted after the
        hook): wait for it here so it cannot write into the bucket later, and give the CUs back"""
        if self._early_work is not None:
            work, _ = self._early_work
            self._early_work = None
            try:
                work.wait()
            except Exception:       # noqa: BLE001  (a torn-down process group)
                pass
        self._release_cus()

    def disable_early_all_reduce(self):
        """No early collective is started from the next backward on.  One already in flight stays owned by the
        all_reduce() call that follows (which also returns its CUs)."""
        from . import functional as Fn
        if self._on_ready is not None and self._on_ready in Fn.grad_ready_hooks:
            Fn.grad_ready_hooks.remove(self._on_ready)
        self._on_ready = None
        self._early = None
        if self._early_work is None:
            self._release_cus()

    def __del__(self):
        try:
            self.disable_early_all_reduce()
            self._release_cus()
        except Exception:           # noqa: BLE001  (interpreter shutdown)
            pass

    def _scale(self, world: int):
        if self.defer_scale:
            self.grad_scale = 1.0 / world           # folded into the fused optimizer step: no pass over the bucket
        else:
            self.flat.mul_(1.0 / world)

    def apply_deferred_scale(self):
        """multiply the bucket by the pending 1 / world-size factor now (one pass) instead of inside the next fused
        optimizer step: afterwards ``p.grad`` are the global-batch mean gradients, as after a plain all-reduce"""
        if self.grad_scale != 1.0:
            self.flat.mul_(self.grad_scale)
            self.grad_scale = 1.0

    def all_reduce(self, group=None, chunks: int = 1):
        """sum over ranks then divide by world size (== DataParallel's global-batch mean).

        With a fused optimizer attached (``defer_scale``) the division happens inside its step kernel: between this call
        and ``optimizer.step()`` every ``p.grad`` holds the cross-rank SUM (``world`` times the mean).  Code that reads
        gradients in between -- ``clip_grad_norm_``, gradient-norm logging, a non-fused optimizer over ``bucket.params``
        -- must call ``apply_deferred_scale()`` first."""
        if not (dist.is_available() and dist.is_initialized()):
            return
        world = dist.get_world_size(group)
        if _single_rank(group):
            return
        if self._early_work is not None:    # the transformer's slice is already in flight (enable_early_all_reduce)
            work, lo = self._early_work     # (lo travels with the work: disable_early_all_reduce() may have run since)
            self._early_work = None
            self._release_cus()                 # what is launched from here on runs after the collective (work.wait())
            if lo > 0:
                dist.all_reduce(self.flat[:lo], op=dist.ReduceOp.SUM, group=group)
            work.wait()
            self._scale(world)
            return
        if chunks <= 1:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
        else:
            works = [dist.all_reduce(c, op=dist.ReduceOp.SUM, group=group, async_op=True)
                     for c in self.flat.chunk(chunks)]
            for w in works:
                w.wait()
        self._scale(world)


class HostScalar:
    """The value of a device scalar on the host WITHOUT draining the launch queue.

    ``loss.item()`` (train_CNN.py:534) copies on the current stream: it returns when everything enqueued before it -- the
    whole backward pass and the optimizer step -- has run, and the next step then starts from an empty queue (measured at
    C2: +1.4 ... 2.1 ms per step; the copy itself is nothing).  ``HostScalar(loss)`` created where the value is PRODUCED
    (right after the loss / the accuracy count, before ``backward()``) records an event there and copies device -> pinned
    host memory on a stream of its own behind that event; ``.item()`` -- called where the loop needs the number, after
    ``optimizer.step()`` -- waits for that copy only, which finished while the backward pass was still running.  Same
    values at the same program points as the reference loop, no device-wide sync (tools/host_boundary_probe.py, round 6 with
    the probe's upload race fixed: 52.86 ms per step against 53.22 with ``.item()``, 52.33 with resident inputs and no
    readback; round 5, before the host work at the head of a step moved behind the optimizer step: 53.5 / 54.9 / 53.1)."""

    _streams: dict = {}

    def __init__(self, t: torch.Tensor):
        if t.numel() != 1:
            raise ValueError('HostScalar takes a one-element tensor')
        t = t.detach().reshape(1)
        if not t.is_cuda:                       # nothing to overlap: a host tensor is its own value
            self._value, self._done, self._buf = t[0].item(), None, None
            return
        idx = t.device.index
        side = HostScalar._streams.get(idx)
        if side is None:
            side = HostScalar._streams[idx] = torch.cuda.Stream(device=t.device)
        self._buf = torch.empty((1,), dtype=t.dtype).pin_memory()
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream(t.device))
        with torch.cuda.stream(side):
            side.wait_event(ready)
            self._buf.copy_(t, non_blocking=True)
            self._done = torch.cuda.Event()
            self._done.record(side)
        t.record_stream(side)                   # the allocator must not hand t's block out before the copy has read it
        self._value = None

    def item(self):
        if self._value is None:
            self._done.synchronize()
            self._value = self._buf[0].item()
        return self._value

    def __float__(self):
        return float(self.item())

    def __int__(self):
        return int(self.item())


def broadcast_parameters(model: torch.nn.Module, src: int = 0, group=None):
    """identical weights/buffers on every rank (DataParallel replicates module 0 each step)."""
    if not (dist.is_available() and dist.is_initialized()) or _single_rank(group):
        return
    for t in list(model.parameters()) + list(model.buffers()):
        dist.broadcast(t.data, src, group=group)
    # `.data` has its own version counter: the cached bf16 / transposed operand copies of any weight that was already
    # used in a forward would survive the overwrite.  (Anyone writing parameters through `.data` or raw pointers calls
    # ops.invalidate_weight_cache() likewise.)
    from . import ops
    ops.invalidate_weight_cache()


def shard_batch(x: torch.Tensor, rank: int, world: int) -> torch.Tensor:
    """rank r gets clips [r*B/W, (r+1)*B/W)."""
    b = x.shape[0]
    if b % world != 0:
        raise ValueError('global batch %d is not divisible by world size %d' % (b, world))
    per = b // world
    return x[rank * per:(rank + 1) * per]


class _FusedOptimizer(torch.optim.Optimizer):
    """Common part of the fused optimizers: one HIP launch over GradBucket's flat parameter / gradient buffers
    (reference optimizers: train_CNN.py:196-201).  ``zero_grad=True`` makes the step kernel
    write zeros over the gradients it has just consumed, so the training loop needs no zero-grad pass.

    They are torch.optim.Optimizer objects: ONE param group over the bucket's parameters whose hyper-parameters
    (``lr`` ...) are read at every step, so ``lr_scheduler.CosineAnnealingLR(optimizer, ...)`` (train_CNN.py:202)
    works; ``state_dict()`` / ``load_state_dict()`` carry the step count and the flat state buffers."""

    _state_names: Tuple[str, ...] = ()

    def __init__(self, bucket: GradBucket, defaults: dict, zero_grad: bool):
        if bucket.flat_params is None:
            raise ValueError('the fused optimizers need GradBucket(..., flatten_params=True)')
        if not bucket.flat.is_cuda:
            raise RuntimeError('the fused optimizers run on the GPU only (no CPU fallback)')
        super().__init__(bucket.params, defaults)
        self.bucket = bucket
        bucket.defer_scale = True           # all_reduce() leaves the sum; step() applies 1 / world size
        self.fused_zero_grad = bool(zero_grad)
        self.steps = 0

    @property
    def hyper(self) -> dict:
        return self.param_groups[0]

    def zero_grad(self, set_to_none: bool = False):
        if set_to_none:
            raise ValueError('gradients are views of the flat bucket: they cannot be set to None')
        # A backward pass that aborted behind the grad-ready hook leaves an early all-reduce nobody collected (and the CU
        # reserve raised): it is waited for and dropped HERE, on every call -- the fused step's re-zeroing skips
        # bucket.zero(), which used to be the only way to reach this recovery -- and the half-reduced bucket of
        # such a step is really zeroed.
        aborted = self.bucket._early_work is not None
        self.bucket._drop_early_work()
        if aborted or not (self.fused_zero_grad and self.steps > 0):
            self.bucket.zero()

    def _done(self):
        from . import ops
        self.steps += 1
        self.bucket.grad_scale = 1.0
        ops.invalidate_weight_cache()       # the kernel wrote parameters without bumping their version counters
        # Re-cast the bf16 operand copies NOW, right behind the update: the bookkeeping (~0.7 ms of Python for ~40 operands)
        # and the grouped cast launch then happen while the GPU is still working through the backward pass, instead of at
        # the head of the next forward pass -- where, in a loop that syncs every step (loss.item(), train_CNN.py:534), the
        # GPU sat idle for it (tools/sync_gap_probe.py: 0.65 ms per step).  The forward's own call finds nothing stale.
        ops.refresh_stale_operands()

    def _slices(self):
        off = 0
        for i, p in enumerate(self.bucket.params):
            n = p.numel()
            yield i, p, off, n
            off += n

    def _extra_state(self, i) -> dict:      # per-parameter entries beside the flat buffers (AdamW's step count)
        return {}

    def state_dict(self):
        """torch.optim's format -- ``{'state': {index: {name: tensor}}, 'param_groups': [{..., 'params': [indices]}]}``
        -- with the per-parameter tensors cut out of the flat state buffers, so a checkpoint written here loads into
        torch.optim.SGD / AdamW over the same parameters and the other way round.  ``fused_steps`` (ignored by torch)
        keeps the step count for SGD, whose torch state has none."""
        state = {}
        if self.steps > 0:                  # torch optimizers have no state before their first step either
            for i, p, off, n in self._slices():
                st = {name: getattr(self, name)[off:off + n].view_as(p).detach().clone() for name in self._state_names}
                st.update(self._extra_state(i))
                state[i] = st
        group = {k: v for k, v in self.hyper.items() if k != 'params'}
        group['params'] = list(range(len(self.bucket.params)))
        return {'state': state, 'param_groups': [group], 'fused_steps': self.steps}

    def load_state_dict(self, sd):
        groups = sd['param_groups']
        if len(groups) != 1 or len(groups[0]['params']) != len(self.bucket.params):
            raise ValueError('fused optimizer: expected ONE param group over %d parameters' % len(self.bucket.params))
        # every saved key, as torch.optim does (a checkpoint written with an LR scheduler attached carries
        # 'initial_lr', which CosineAnnealingLR(last_epoch=E) needs on resume); the kernels read only the keys they know
        self.hyper.update({k: v for k, v in groups[0].items() if k != 'params'})
        state = sd['state']
        steps = 0
        for i, p, off, n in self._slices():
            st = state.get(i, state.get(str(i)))
            for name in self._state_names:
                dst = getattr(self, name)[off:off + n]
                src = None if st is None else st.get(name)
                if src is None:
                    dst.zero_()
                else:
                    if src.numel() != n:
                        raise ValueError('optimizer state %r of parameter %d has %d elements, expected %d' % (name, i, src.numel(), n))
                    dst.copy_(src.reshape(-1))
            if st is not None:
                steps = max(steps, int(float(st['step'])) if 'step' in st else 1)
        self.steps = int(sd.get('fused_steps', steps))


class FusedSGD(_FusedOptimizer):
    """torch.optim.SGD(params, lr, momentum, dampening, weight_decay, nesterov) over the flat buffers."""

    _state_names = ('momentum_buffer',)

    def __init__(self, bucket: GradBucket, lr: float, momentum: float = 0.0, dampening: float = 0.0,
                 weight_decay: float = 0.0, nesterov: bool = False, zero_grad: bool = False):
        if nesterov and (momentum <= 0 or dampening != 0):
            raise ValueError('Nesterov momentum requires a momentum and zero dampening')
        super().__init__(bucket, dict(lr=float(lr), momentum=float(momentum), dampening=float(dampening),
                                      weight_decay=float(weight_decay), nesterov=bool(nesterov)), zero_grad)
        self.momentum_buffer = torch.zeros_like(bucket.flat)

    @torch.no_grad()
    def step(self, closure=None):
        from . import _lib, ops
        b, h = self.bucket, self.hyper
        _lib.check(_lib.lib().istvt_sgd_momentum(b.flat_params.data_ptr(), b.flat.data_ptr(), self.momentum_buffer.data_ptr(),
                                                 b.numel, h['lr'], h['momentum'], h['dampening'], h['weight_decay'],
                                                 int(h['nesterov']), int(self.steps == 0), int(self.fused_zero_grad),
                                                 float(b.grad_scale), ops._stream()), 'istvt_sgd_momentum')
        self._done()


class FusedAdamW(_FusedOptimizer):
    """torch.optim.AdamW(params, lr, betas, eps, weight_decay) (amsgrad off) over the flat buffers."""

    _state_names = ('exp_avg', 'exp_avg_sq')

    def _extra_state(self, i):
        return {'step': torch.tensor(float(self.steps))}

    def __init__(self, bucket: GradBucket, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 1e-2, zero_grad: bool = False):
        super().__init__(bucket, dict(lr=float(lr), betas=(float(betas[0]), float(betas[1])), eps=float(eps),
                                      weight_decay=float(weight_decay)), zero_grad)
        self.exp_avg = torch.zeros_like(bucket.flat)
        self.exp_avg_sq = torch.zeros_like(bucket.flat)

    @torch.no_grad()
    def step(self, closure=None):
        from . import _lib, ops
        b, h = self.bucket, self.hyper
        _lib.check(_lib.lib().istvt_adamw(b.flat_params.data_ptr(), b.flat.data_ptr(), self.exp_avg.data_ptr(),
                                          self.exp_avg_sq.data_ptr(), b.numel, h['lr'], h['betas'][0], h['betas'][1],
                                          h['eps'], h['weight_decay'], self.steps + 1, int(self.fused_zero_grad),
                                          float(b.grad_scale), ops._stream()), 'istvt_adamw')
        self._done()


# ------------------------------------------------------------------------------------------------------------------------
# HIP-graph replay of the training forward / backward (round 6)
# ------------------------------------------------------------------------------------------------------------------------
class _ReplayBackward(torch.autograd.Function):
    """Puts a replayed forward graph's output into the caller's autograd graph: its backward copies the incoming gradient
    into the static buffer and replays the captured backward graph (parameter gradients land in the flat bucket, as they
    do when the kernels are launched one by one)."""

    @staticmethod
    def forward(ctx, anchor, entry, token):
        ctx.entry = entry
        ctx.token = token            # alive until this node has run its backward or has been dropped: StepGraphs' "pending" flag
        return entry.static_out.detach().clone()

    @staticmethod
    def backward(ctx, g):
        from . import functional as Fn
        entry, ctx.entry, ctx.token = ctx.entry, None, None
        entry.static_gout.copy_(g)
        # A launch-by-launch backward pass may share this autograd pass (a second forward that fell back): its weight-gradient
        # launches sit on the side stream and read-modify-write the same .grad buffers as the graph's.  Order them: what the
        # side stream has so far, then the graph, then whatever it gets later.  (No-ops in the all-graph loop.)
        cur = torch.cuda.current_stream(g.device)
        side = Fn._overlap['streams'].get(g.device.index)
        if side is not None:
            cur.wait_stream(side)
        entry.g_bwd.replay()
        if side is not None:
            side.wait_stream(cur)
        return None, None, None


class _GraphEntry:
    __slots__ = ('static_x', 'static_out', 'static_gout', 'g_fwd', 'g_bwd', 'keep', 'fingerprint', 'calls', 'pool')


class StepGraphs:
    """The launch sequences of ``model(x)`` and of its backward pass, captured once per (input shape, mode) as HIP graphs
    (hipGraph via torch.cuda.CUDAGraph; the kernels are launched through the C ABI on torch's current stream, which is the
    capturing stream inside ``torch.cuda.graph``) and replayed from then on: ~1 400 launches per training step at C2 become
    two graph launches, the per-step host time falls from 14-23 ms of Python to ~3 ms, and a launch-bound configuration (C1,
    small batches) runs at the speed of its kernels.

    Transparent to the caller's loop (train_CNN.py:497-537 runs unchanged): ``model.enable_step_graphs()`` makes
    ``model(x)`` return logits that are part of the caller's autograd graph; ``loss.backward()`` replays the backward graph.
    A forward under ``torch.no_grad()`` (the validation loop, train_CNN.py:837-944) is a forward-only graph of its own.
    Falls back to the launch-by-launch path whenever the preconditions do not hold:
      * a CUDA input that does not require a gradient;
      * with gradients enabled: every live parameter's gradient is a view of a ``GradBucket(fuse_accumulate=True)`` (the
        kernels write the bucket directly; a torch optimizer's ``zero_grad(set_to_none=True)`` breaks that and is detected);
      * no early all-reduce hook (N > 1 with the overlapped schedule stays launch-by-launch: the collective's handle is host state);
      * no per-kernel instrumentation (bench.py's profiled step);
      * the switches that change the launch sequence (compute dtype, fp8 attention operands, dead-row elimination, the
        weight-gradient stream / grouping) are part of an entry's key: a toggle captures a new entry, it never replays the old one;
      * the previous graphed forward of this entry has had its backward (or was dropped): an entry owns ONE set of saved
        activations.
    Addresses: parameters, gradients and every cached operand copy must stay where they were at capture; the cached copies are
    refreshed in place (ops.set_static_addresses), parameters and gradients are fingerprinted (data_ptr of each) and a change
    drops the graphs and captures again.  The first ``warmup`` calls of an entry run launch by launch (they fill the operand
    caches, the statistics arena, the side stream, the kernels' one-time attribute calls).  Every entry has a memory pool of
    its own: replaying one never touches what another saved for its backward."""

    def __init__(self, model: torch.nn.Module, forward_eager, warmup: int = 2, max_entries: int = 4):
        self.model = weakref.ref(model)
        # (a bound method of the model would make model -> StepGraphs -> model a cycle only the garbage collector breaks:
        #  the graphs' memory pools, and static-address mode, would outlive the model for as long as that takes)
        self._fwd = weakref.WeakMethod(forward_eager) if hasattr(forward_eager, '__self__') else (lambda f=forward_eager: f)
        self.warmup = int(warmup)
        self.max_entries = int(max_entries)
        self.entries = {}
        self.seen = {}
        self.pending = {}               # entry key -> weakref to the token of a graphed forward whose backward has not run
        self.anchor = None
        self.live = None
        self._probe = None
        self.stats = {'captures': 0, 'replays': 0, 'eager': 0, 'recaptures': 0}
        self.last_reason = None
        from . import ops
        ops._static_holders.add(self)   # static-address mode is on exactly while some StepGraphs holds a captured graph

    def forward_eager(self, x):
        return self._fwd()(x)

    def _config(self, m):
        from . import functional as Fn
        if m is None:
            return None
        if self._probe is None:
            fp8 = [mod for mod in m.modules() if hasattr(mod, 'attn_fp8')]
            dre = [mod for mod in m.modules() if hasattr(mod, 'dead_row_elimination')]
            self._probe = (fp8, dre)
        fp8, dre = self._probe
        return (str(getattr(m, 'compute_dtype', None)), tuple(bool(q.attn_fp8) for q in fp8[:1]) + tuple(bool(q.attn_fp8) for q in fp8[-1:]),
                tuple(bool(q.dead_row_elimination) for q in dre), bool(Fn._overlap['on']), int(Fn._overlap['group']),
                bool(Fn.GELU_SAVE_DERIV[0]), os.environ.get('ISTVT_GEMM_TM', ''))

    # -- preconditions -----------------------------------------------------------------------------------------------
    def _live(self):
        if self.live is None:
            self.live = [p for _, p in live_named_parameters(self.model())]
        return self.live

    def _fingerprint(self, grads: bool):
        fp = []
        for p in self._live():
            fp.append(p.data_ptr())
            if grads:
                g = p.grad
                if g is None or not getattr(p, '_istvt_fused_grad', False):
                    return None
                fp.append(g.data_ptr())
        return tuple(fp)

    def why_not(self, x, key):
        from . import functional as Fn
        from . import ops
        if self.model() is None:
            return 'model gone'
        if not x.is_cuda or x.requires_grad:
            return 'input on the host or requiring a gradient'
        if ops.kernel_profile is not None or ops.gemm_profile is not None:
            return 'per-kernel instrumentation on'
        if torch.cuda.is_current_stream_capturing():
            return 'already inside a capture'
        if key[-1]:                     # gradients enabled
            if Fn.grad_ready_hooks:
                return 'early all-reduce hook registered'
            pend = self.pending.get(key)
            if pend is not None and pend() is not None:
                return 'a graphed forward is still waiting for its backward'
        return None

    # -- the call ----------------------------------------------------------------------------------------------------
    def __call__(self, x):
        from . import ops
        m = self.model()
        grads = torch.is_grad_enabled()
        # the key names everything that selects a launch sequence: the input, the mode, and the switches that change which
        # kernels a forward / backward issues (a toggle after a capture must not replay the old sequence)
        key = (tuple(x.shape), x.dtype, x.device.index, self._config(m), bool(m is not None and m.training), grads)
        reason = self.why_not(x, key)
        fp = None
        if reason is None:
            fp = self._fingerprint(grads)
            if fp is None:
                reason = 'a live gradient is not a fused-bucket view'
        if reason is None:
            n = self.seen.get(key, 0)
            self.seen[key] = n + 1
            if n < self.warmup:
                reason = 'warm-up'
            elif key not in self.entries and len(self.entries) >= self.max_entries:
                reason = 'more (shape, mode) entries than max_entries'
        if reason is not None:
            self.last_reason = reason
            self.stats['eager'] += 1
            return self.forward_eager(x)
        ent = self.entries.get(key)
        if ent is not None and ent.fingerprint != fp:
            # a parameter or gradient moved (a new bucket, load_state_dict into new storage, .to()): the captured pointers
            # are dead -- drop every graph and start over
            self.drop()
            self.stats['recaptures'] += 1
            ent = None
        if ent is None:
            ent = self._capture(x, key, fp, grads)
        # operand copies the optimizer step (or a load_state_dict) made stale: in place, before the graph reads them
        ops.refresh_stale_operands()
        if ent.static_x.data_ptr() != x.data_ptr():
            ent.static_x.copy_(x)
        ent.g_fwd.replay()
        ent.calls += 1
        self.stats['replays'] += 1
        if not grads:
            return ent.static_out.detach().clone()

        class _Token:
            __slots__ = ('__weakref__',)
        token = _Token()
        self.pending[key] = weakref.ref(token)
        return _ReplayBackward.apply(self.anchor, ent, token)

    def _capture(self, x, key, fp, grads):
        from . import functional as Fn
        from . import ops
        from . import stem as stem_mod
        Fn.flush_stale_joins()
        if self.anchor is None or self.anchor.device != x.device:
            self.anchor = torch.zeros((), device=x.device, requires_grad=True)
        ent = _GraphEntry()
        ent.calls = 0
        ent.fingerprint = fp
        ent.static_x = x.detach().clone()
        ent.static_gout = ent.g_bwd = None
        self.entries[key] = ent         # from here on ops.static_addresses() is true: every refresh is in place
        ops.refresh_stale_operands()
        ent.pool = torch.cuda.graph_pool_handle()
        ent.g_fwd = torch.cuda.CUDAGraph()
        # The statistics arena of the launch-by-launch path (stem._arena: one buffer, zeroed once per forward, the backward's
        # accumulators cut from it behind the forward's) must not be shared with a graph: a launch-by-launch forward between
        # this entry's forward and backward would zero and re-cut what the captured backward accumulates into.  The capture
        # starts from an empty arena: the captured forward then allocates one of its own from the graph's pool.
        arena_saved = dict(stem_mod._arena)
        stem_mod._arena.update(buf=None, off=0, need=max(arena_saved['need'], arena_saved['off']))
        try:
            self._capture_graphs(ent, grads)
        except BaseException:
            self.entries.pop(key, None)
            raise
        finally:
            arena_private = stem_mod._arena.get('buf')
            stem_mod._arena.clear()
            stem_mod._arena.update(arena_saved)
        # everything outside the graphs' pool whose ADDRESS they captured stays alive with them
        ent.keep = ([h[2] for h in ops._wcache.values()] + [d[2] for d in ops._derived.values()]
                    + [arena_private] + [v[1:] for v in ops._operands.values()])
        self.entries[key] = ent
        self.stats['captures'] += 1
        return ent

    def _capture_graphs(self, ent, grads):
        with torch.set_grad_enabled(grads):
            with torch.cuda.graph(ent.g_fwd, pool=ent.pool, capture_error_mode='thread_local'):
                out = self.forward_eager(ent.static_x)
            ent.static_out = out
            if grads:
                ent.static_gout = torch.zeros_like(out)
                ent.g_bwd = torch.cuda.CUDAGraph()
                with torch.cuda.graph(ent.g_bwd, pool=ent.pool, capture_error_mode='thread_local'):
                    # autograd.grad, not backward(): no AccumulateGrad node runs (those of the warm-up steps live on the
                    # default stream, and the engine's stream hand-over to them is illegal inside a capture -- it crashed
                    # the process); the kernels write the fused bucket themselves and the Functions return None for their
                    # parameters
                    got = torch.autograd.grad(out, self._live(), grad_outputs=ent.static_gout, allow_unused=True)
                    # a Function that does hand a gradient tensor back is accumulated here, inside the capture, as
                    # AccumulateGrad would have done
                    for p, g in zip(self._live(), got):
                        if g is not None:
                            p.grad.add_(g)
                del got

    def drop(self):
        """forget every captured graph (each entry's pool is released once the last reference is gone)"""
        self.entries.clear()
        self.seen.clear()
        self.pending.clear()
        self.live = None
