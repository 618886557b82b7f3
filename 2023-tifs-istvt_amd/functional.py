"""torch.autograd.Function glue: each Function is a thin autograd wrapper whose forward and
backward are sequences of HIP launches (ops.py).  No arithmetic is done with torch ops here
except zero-initialising gradient buffers.

Fused gradient accumulation: when a parameter was registered with
``parallel.GradBucket(..., fuse_accumulate=True)`` its ``.grad`` (a view into the flat fp32
bucket) is handed to the weight-gradient kernels as their accumulate target and the Function
returns ``None`` for it, which removes one zero-fill and one add pass per parameter per step.
"""
from __future__ import annotations

import os

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import ops

Tensor = torch.Tensor


def _target(p):
    """(buffer to accumulate this parameter's gradient into, value to return to autograd)"""
    if p is None:
        return None, None
    if getattr(p, '_istvt_fused_grad', False) and p.grad is not None:
        return p.grad, None
    z = torch.zeros(p.shape, dtype=torch.float32, device=p.device)
    return z, z


class BiasSink:
    """Names the bias parameter of the nn.Linear whose output a LayerNorm normalises.  That Linear's bias gradient is
    the column sum of the LayerNorm's input gradient, so the LayerNorm backward kernel accumulates it on the way
    (ops.layernorm_bwd(dcol=...)) and the Linear, called with defer_bias=True, skips its own pass over dy.  Only valid
    when the gradient lands straight in ``bias.grad`` (GradBucket(fuse_accumulate=True)): see ``usable``."""
    __slots__ = ('bias',)

    def __init__(self, bias):
        self.bias = bias

    @staticmethod
    def usable(bias) -> bool:
        return (bias is not None and torch.is_grad_enabled() and bias.requires_grad
                and getattr(bias, '_istvt_fused_grad', False) and bias.grad is not None)

    def buffer(self):
        return self.bias.grad


class LayerNormFn(Function):
    """nn.LayerNorm over the last dim (reference module.py:15-21).  fork=True also returns the input itself: a
    caller that uses x both as the LayerNorm input and as a residual takes the second output for the residual, and
    the gradient arriving through it is added inside the LayerNorm backward kernel (no separate add pass)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, fork=False, sink=None):
        y, mean, rstd = ops.layernorm_fwd(x, gamma, beta, eps, pad=True)
        ctx.save_for_backward(x, mean, rstd, gamma, beta)
        ctx.set_materialize_grads(False)
        ctx.sink = sink
        return (y, x.view_as(x)) if fork else y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy, dres=None):
        x, mean, rstd, gamma, beta = ctx.saved_tensors
        dcol = ctx.sink.buffer() if ctx.sink is not None else None
        if dy is None:                       # only the pass-through output was used
            if dcol is not None and dres is not None:
                ops.colsum(dres.reshape(-1, dres.shape[-1]), out=dcol)
            return dres, None, None, None, None, None
        dg, rg = _target(gamma)
        db, rb = _target(beta)
        dx = ops.layernorm_bwd(dy, x, mean, rstd, gamma, dg, db, dres=dres, pad=True, dcol=dcol, defer=_ln_defer(dy, (rg, rb)))
        return dx, rg, rb, None, None, None


class LayerNormDiffFn(Function):
    """LayerNorm + the frame difference of module.py:193 in one kernel (bfloat16 path of TemporalResidualAttention):
    returns (y, diff[, x]) with diff = cat(y[:, :2], y[:, 2:] - y[:, 1:-1]) over frames, taken in fp32 before the rounding
    to bfloat16.  y and diff are the two planes of one buffer (ops.layernorm_fwd_diff).  diff carries NO gradient of its
    own: its only consumer, LinearCatSelFn + TemporalAttnFn(diff=2), returns the gradient of the whole
    difference-project-attend chain with respect to y (the attention backward applies the difference's adjoint to dq and
    dk), so the backward here is the plain LayerNorm backward."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, geom, fork=False, sink=None):
        B, F, P = geom
        x2 = x.reshape(-1, x.shape[-1])
        y, yd, mean, rstd = ops.layernorm_fwd_diff(x2, gamma, beta, eps, B, F, P)
        ctx.save_for_backward(x, mean, rstd, gamma, beta)
        ctx.set_materialize_grads(False)
        ctx.sink = sink
        y, yd = y.view(*x.shape), yd.view(*x.shape)
        ctx.mark_non_differentiable(yd)         # the tensor that is RETURNED (a view made after the mark would not carry it)
        return (y, yd, x.view_as(x)) if fork else (y, yd)

    @staticmethod
    @once_differentiable
    def backward(ctx, dy, _dyd=None, dres=None):
        if _dyd is not None:
            # the difference plane is an operand, not a differentiable output: a consumer that sends a gradient into it
            # would have it dropped silently
            raise RuntimeError('LayerNormDiffFn: the frame-difference plane carries no gradient of its own (its consumer '
                               'returns the gradient of the whole difference-project-attend chain through y)')
        x, mean, rstd, gamma, beta = ctx.saved_tensors
        dcol = ctx.sink.buffer() if ctx.sink is not None else None
        if dy is None:
            if dcol is not None and dres is not None:
                ops.colsum(dres.reshape(-1, dres.shape[-1]), out=dcol)
            return dres, None, None, None, None, None, None
        dg, rg = _target(gamma)
        db, rb = _target(beta)
        dx = ops.layernorm_bwd(dy, x, mean, rstd, gamma, dg, db, dres=dres, pad=True, dcol=dcol, defer=_ln_defer(dy, (rg, rb)))
        return dx, rg, rb, None, None, None, None


class FrameDiffFn(Function):
    """residual = cat(x[:, :2], x[:, 2:] - x[:, 1:-1]) over frames (module.py:193) as a pass of its own.  The model does
    not use it (TemporalResidualAttention differences q and k inside the attention kernels); it is the reference-order
    form the fused path is tested against."""

    @staticmethod
    def forward(ctx, x, B, F, P):
        ctx.geom = (B, F, P)
        return ops.frame_diff(x, B, F, P)

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        return ops.frame_diff(g, *ctx.geom, adjoint=True), None, None, None


# Weight gradients on a second HIP stream.  Nothing downstream in the backward pass reads a weight gradient, so the
# TN GEMM + split-K reduce of a Linear are enqueued on a side stream that waits for the main stream at the point of
# the call (dy and x are complete there) and is joined back into the main stream by an end-of-backward callback, which also releases the operands.
# The hardware then fills the partial last round of the persistent input-gradient GEMMs, and the CUs the
# bandwidth-bound kernels leave idle, with weight-gradient workgroups.  Only used when the gradient lands directly
# in the parameter's .grad (GradBucket(fuse_accumulate=True)): a returned tensor would be consumed by autograd on
# the main stream.  ISTVT_WGRAD_STREAM=0 (or set_wgrad_overlap(False)) serialises everything on one stream.
_overlap = {'on': os.environ.get('ISTVT_WGRAD_STREAM', '1') != '0', 'streams': {}, 'pending': {}, 'keep': {},
            # Grouping: the weight gradients of up to 8 Linears with the same row count (one transformer layer) are queued
            # and launched as ONE (tile, split) grid (ops.linear_wgrad_group): 120 tiles need 2 reduction splits to fill
            # the chip instead of 7..42 per GEMM alone, so the fp32 partial slabs shrink from 512 MB to 60 MB per layer.
            'group': int(os.environ.get('ISTVT_WGRAD_GROUP', '8')), 'queue': {}, 'lnq': {}}


def set_wgrad_overlap(on: bool):
    _overlap['on'] = bool(on)


def set_wgrad_group(n: int):
    """how many weight gradients are launched together (1: each on its own, as they are produced)"""
    if not 1 <= int(n) <= ops.WGRAD_GROUP_MAX:
        raise ValueError('weight-gradient group size must be in 1..%d' % ops.WGRAD_GROUP_MAX)
    _overlap['group'] = int(n)


def join_side_stream(dev=None):
    """Make the current backward's main stream wait for the side stream now (instead of at the end of backward): what a
    caller needs before it hands weight gradients to a collective mid-backward.  Queued weight gradients are launched
    first."""
    if dev is None:
        dev = torch.cuda.current_device()
    _join_side(dev)


def no_data_parallel(self):
    """``nn.Module._replicate_for_data_parallel`` of every module in network/: the reference loop wraps the model in
    single-process ``nn.DataParallel`` when ``-d`` names more than one device (train_CNN.py:185-186).  That mode cannot
    work here -- replicas are shallow copies made per forward on worker THREADS, while the weight-operand cache
    (ops._wcache), the side-stream join state and the flat gradient bucket are per process and keyed by the original
    parameters -- so it fails at the first replication with the way out, instead of training on stale operands."""
    raise RuntimeError(
        '%s: torch.nn.DataParallel is not supported by the HIP path (per-process operand caches, side streams and the '
        'flat gradient bucket).  Run one process per GPU instead: `python -m torch.distributed.run --nproc-per-node N '
        '...` (torchrun) with istvt_amd.parallel.GradBucket.all_reduce() after backward, as bench.py --gpus N does; '
        'per-rank BatchNorm statistics then match what DataParallel would have computed.' % type(self).__name__)


# Callables run by TokensFn.backward once its own kernels are enqueued.  Token assembly is the first operation of the
# transformer, so at that point every gradient of the transformer (98.8 % of the bucket) has been enqueued -- on the main
# stream or the side stream -- while the whole stem backward is still to come: parallel.GradBucket hangs the
# all-reduce of that part of the bucket here to overlap it with the stem backward.
grad_ready_hooks = []


def _flush_group(dev):
    """launch the queued weight gradients (on the side stream when the overlap is on, else on the current stream)"""
    q = _overlap['queue'].pop(dev, None)
    lq = _overlap['lnq'].pop(dev, None)          # deferred LayerNorm parameter-gradient folds (_ln_defer)
    if not q and not lq:
        return

    def launch():
        if q:
            ops.linear_wgrad_group(q)
        for ent in lq or ():
            ops.layernorm_bwd_reduce(*ent)

    if _overlap['on']:
        side = _side_stream(torch.device('cuda', dev))
        side.wait_stream(torch.cuda.current_stream(torch.device('cuda', dev)))
        with torch.cuda.stream(side):
            launch()
    else:
        launch()
    if lq:
        _overlap['keep'].setdefault(dev, []).append(lq)      # the workspaces: alive until the join


def _side_stream(device):
    dev = device.index
    side = _overlap['streams'].get(dev)
    if side is None:
        side = _overlap['streams'][dev] = torch.cuda.Stream(device=device, priority=int(os.environ.get('ISTVT_WGRAD_PRIO', '0')))
    return side


def _join_side(dev, task=None):
    """task: the autograd graph task whose end-of-backward callback this is (None: join whatever is pending).  A
    callback of an EARLIER task whose entry has already been flushed and replaced must not touch the new entry."""
    ent = _overlap['pending'].get(dev)
    if ent is None or (task is not None and ent[1] != task):
        return
    del _overlap['pending'][dev]
    _flush_group(dev)
    side = _overlap['streams'].get(dev)
    if side is not None:
        ent[0].wait_stream(side)
    # The operands of the side-stream launches were kept alive until here instead of being handed to the allocator with
    # record_stream(): blocks parked behind side-stream events made the caching allocator grow by a run-dependent
    # 16-56 GB at C2 (reserved 47-87 GB against 31 GB single-stream).  Released after the join they go back to the main
    # stream's pool in order: ~10 GB more at the peak, the same every run.
    _overlap['keep'].pop(dev, None)


def _graph_task() -> int:
    return torch._C._current_graph_task_id()


def flush_stale_joins():
    """A backward pass that aborted (an exception in a later node, an out-of-memory error) never runs its
    end-of-backward callback: its join entry and the operands it keeps alive would stay behind, every later backward
    would skip registering a join and the optimizer would read weight gradients the side stream may still be writing.
    Called from the forward (outside any backward every entry is stale) and from _wgrad (an entry of another graph
    task is stale); joins the side stream into the CURRENT stream and drops the entry.  What happens to the weight
    gradients that pass still had QUEUED depends on who is asking:
      * from inside a backward (task != -1) with an entry of ANOTHER task: that entry need not be an aborted pass -- a
        re-entrant backward (torch.utils.checkpoint's recompute, a Function that calls backward) has a task id of its
        own while the outer pass's gradients are still queued, and those must land: they are LAUNCHED;
      * from a forward (task == -1): a pending entry can only be left by a backward that aborted.  The usual loop is
        zero_grad() -> forward -> backward (and the fused optimizer zeroes inside step()), so the gradients have been
        zeroed since: launching the aborted pass's dy^T x products now would add them into the NEXT step's gradients
        (an OOM-skip-batch loop would silently train on a batch it meant to skip).  They are DROPPED."""
    if not _overlap['pending']:
        return
    task = _graph_task()
    for dev in list(_overlap['pending']):
        if task == -1 or _overlap['pending'][dev][1] != task:
            if task == -1:
                _overlap['queue'].pop(dev, None)
                _overlap['lnq'].pop(dev, None)
            else:
                _flush_group(dev)
            side = _overlap['streams'].get(dev)
            if side is not None:
                torch.cuda.current_stream(torch.device('cuda', dev)).wait_stream(side)
            del _overlap['pending'][dev]
            _overlap['keep'].pop(dev, None)


def _ensure_join(device):
    """register the end-of-backward join of the side stream for the current backward pass (once per pass and device)"""
    dev = device.index
    main = torch.cuda.current_stream(device)
    task = _graph_task()
    ent = _overlap['pending'].get(dev)
    if ent is not None and ent[1] != task:
        flush_stale_joins()                      # left behind by a backward pass that never finished
        ent = None
    if ent is None:
        _overlap['pending'][dev] = (main, task)
        torch.autograd.Variable._execution_engine.queue_callback(lambda: _join_side(dev, task))
    return dev, main


_LN_DEFER = os.environ.get('ISTVT_LN_DEFER', '1') != '0'      # 0: every LayerNorm backward folds its partial rows itself


def _ln_defer(dy, rets):
    """The `defer` argument of ops.layernorm_bwd for a LayerNorm backward inside an autograd pass: the fold of the
    kernel's partial rows into the parameter gradients (one 5 us launch per LayerNorm, 38 per training step, that nothing
    on the main stream waits for) is queued and launched with the next group of weight gradients on their stream.  Only
    when every gradient lands straight in its .grad buffer (rets all None: GradBucket(fuse_accumulate=True)) -- a tensor
    handed back to autograd would be read before the fold has written it -- and the overlap is on."""
    if not (_overlap['on'] and _LN_DEFER and dy.is_cuda and all(r is None for r in rets) and _graph_task() != -1):
        return None
    dev, _ = _ensure_join(dy.device)

    def sink(ws, M, D, dgamma, dbeta, dcol):
        _overlap['lnq'].setdefault(dev, []).append((ws, M, D, dgamma, dbeta, dcol))
    return sink


def _wgrad(dy, x, weight):
    buf, ret = _target(weight)
    out = buf.view(weight.shape[0], -1)
    group = _overlap['group']
    if not (ret is None and dy.is_cuda and (_overlap['on'] or group > 1)):
        ops.linear_wgrad(dy, x, out=out)
        return ret
    dev, main = _ensure_join(dy.device)
    if group > 1 and out.is_contiguous() and ops.wgrad_groupable(dy, x):
        q = _overlap['queue'].get(dev)
        if q and (q[0][0].shape[0] != dy.shape[0] or any(o.data_ptr() == out.data_ptr() for _, _, o in q)):
            _flush_group(dev)                    # another row count (the row-pruned last layer) or a shared weight
            q = None
        if q is None:
            q = _overlap['queue'][dev] = []
        q.append((dy, x, out))
        if len(q) >= group:
            _flush_group(dev)
    elif _overlap['on']:
        side = _side_stream(dy.device)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            ops.linear_wgrad(dy, x, out=out)
    else:
        ops.linear_wgrad(dy, x, out=out)
    _overlap['keep'].setdefault(dev, []).append((dy, x))     # alive until the join (see _join_side)
    return None


def side_launch(device, fn, keep=()):
    """Run `fn` (launches whose results nothing later in this backward pass reads: weight gradients that land straight in
    .grad) on the weight-gradient side stream, joined at the end of the backward pass like _wgrad's; `keep`: the operand
    tensors, alive until the join.  Outside a backward pass, or with the overlap off, `fn` runs in place."""
    if not (_overlap['on'] and device.type == 'cuda' and _graph_task() != -1):
        fn()
        return
    dev, main = _ensure_join(device)
    side = _side_stream(device)
    side.wait_stream(main)
    with torch.cuda.stream(side):
        fn()
    _overlap['keep'].setdefault(dev, []).append(keep)


def _bgrad(dy, bias):
    if bias is None:
        return None
    buf, ret = _target(bias)
    side = _overlap['streams'].get(dy.device.index) if dy.is_cuda else None
    if os.environ.get('ISTVT_BGRAD_SIDE', '0') == '1' and _overlap['on'] and ret is None and side is not None \
            and dy.device.index in _overlap['pending']:
        side.wait_stream(torch.cuda.current_stream(dy.device))
        with torch.cuda.stream(side):
            ops.colsum(dy, out=buf)
        _overlap['keep'].setdefault(dy.device.index, []).append((dy,))
        return None
    ops.colsum(dy, out=buf)
    return ret


class LinearFn(Function):
    """y = x W^T (+ b) (+ residual) on [M, K] inputs (nn.Linear, module.py:74,77,182,183,186)."""

    @staticmethod
    def forward(ctx, x, weight, bias, residual, defer_bias=False):
        """defer_bias: the LayerNorm that consumes y accumulates the bias gradient (BiasSink); skip it here."""
        if _overlap['pending']:
            flush_stale_joins()
        w = ops.weight_as(weight, x.dtype, pad=True)
        y = ops.linear_fwd(x, w, bias, residual, pad=True)
        ctx.save_for_backward(x, weight, bias)
        ctx.has_res = residual is not None
        ctx.defer_bias = bool(defer_bias)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, weight, bias = ctx.saved_tensors
        w = ops.weight_as(weight, dy.dtype, pad=True)       # the forward's operand copy: W^T is derived from it once
        dx = ops.linear_dgrad(dy, w, pad=True) if ctx.needs_input_grad[0] else None
        dw = _wgrad(dy, x, weight) if ctx.needs_input_grad[1] else None
        db = _bgrad(dy, bias) if (bias is not None and ctx.needs_input_grad[2] and not ctx.defer_bias) else None
        dres = dy if ctx.has_res and ctx.needs_input_grad[3] else None
        return dx, dw, db, dres, None


class LinearCatFn(Function):
    """y = x [W_0; W_1; ...]^T: several bias-free nn.Linear layers over ONE input as one GEMM on the stacked operand
    (ops.weight_cat_as); the parameters stay separate.  TemporalResidualAttention's to_qk and to_v (module.py:182-183,
    195-196): one 728 -> 1536 projection, one packed input gradient, per-parameter weight gradients from the column
    ranges of dy."""

    @staticmethod
    def forward(ctx, x, *weights):
        if _overlap['pending']:
            flush_stale_joins()
        w = ops.weight_cat_as(weights, x.dtype)
        y = ops.linear_fwd(x, w, pad=True)
        ctx.save_for_backward(x, *weights)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, *weights = ctx.saved_tensors
        w = ops.weight_cat_as(weights, dy.dtype)
        dx = ops.linear_dgrad(dy, w, pad=True) if ctx.needs_input_grad[0] else None
        grads, n0 = [], 0
        for i, wi in enumerate(weights):
            n = wi.shape[0]
            grads.append(_wgrad(dy[:, n0:n0 + n], x, wi) if ctx.needs_input_grad[1 + i] else None)
            n0 += n
        return (dx, *grads)


class LinearCatSelFn(Function):
    """LinearCatFn whose leading `sel_col` output columns are projections of a SECOND input xd that sits one plane in
    front of x in memory (LayerNormDiffFn's two outputs): q | k = to_qk(diff), v = to_v(x) as ONE GEMM whose A operand is
    picked per column tile (istvt_gemm flags bit 1).  The backward is LinearCatFn's, with respect to x alone: its dy is
    TemporalAttnFn(diff=2)'s gradient, which is already expressed for projections of the un-differenced rows."""

    @staticmethod
    def forward(ctx, x, xd, sel_col, *weights):
        if _overlap['pending']:
            flush_stale_joins()
        M = x.shape[0]
        if (xd.shape != x.shape or xd.stride() != x.stride() or xd.dtype != x.dtype
                or xd.data_ptr() + M * x.stride(0) * x.element_size() != x.data_ptr()):
            raise RuntimeError('LinearCatSelFn: xd must be the plane directly in front of x (ops.layernorm_fwd_diff)')
        w = ops.weight_cat_as(weights, x.dtype)
        y = ops.linear_fwd(xd, w, pad=True, a_sel_col=sel_col)
        ctx.save_for_backward(x, *weights)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, *weights = ctx.saved_tensors
        w = ops.weight_cat_as(weights, dy.dtype)
        dx = ops.linear_dgrad(dy, w, pad=True) if ctx.needs_input_grad[0] else None
        grads, n0 = [], 0
        for i, wi in enumerate(weights):
            n = wi.shape[0]
            grads.append(_wgrad(dy[:, n0:n0 + n], x, wi) if ctx.needs_input_grad[3 + i] else None)
            n0 += n
        return (dx, None, None, *grads)


# The GELU pair exchanges gelu'(u) instead of u (round 6): the forward GEMM's epilogue has the fp32 pre-activation and the
# erf / exp of gelu in registers, so gelu'(u) costs it two more FMAs per element; the backward GEMM's epilogue then multiplies
# by the saved value instead of evaluating erf and exp again (27 % of that epilogue, profiles/r05_a_*).  u itself is used by
# nothing else.  float32: bit-identical (the same gelu'(fp32 u), computed one kernel earlier).  ISTVT_GELU_SAVE_DERIV=0: the u form.
GELU_SAVE_DERIV = [os.environ.get('ISTVT_GELU_SAVE_DERIV', '1') != '0']


class FeedForwardFn(Function):
    """Linear -> exact GELU -> Linear (+ residual) (FeedForward, module.py:23-34)."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, residual, defer_bias=False):
        ctx.gelu_d = GELU_SAVE_DERIV[0]
        u, g = ops.linear_fwd(x, ops.weight_as(w1, x.dtype, pad=True), b1, gelu=True, pad=True, gelu_d=ctx.gelu_d)
        y = ops.linear_fwd(g, ops.weight_as(w2, x.dtype, pad=True), b2, residual, pad=True)
        ctx.save_for_backward(x, u, g, w1, b1, w2, b2)
        ctx.has_res = residual is not None
        ctx.defer_bias = bool(defer_bias)           # b2's gradient comes from the next LayerNorm's backward (BiasSink)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, u, g, w1, b1, w2, b2 = ctx.saved_tensors
        # (dy W2) * gelu'(u); the hidden layer's bias gradient = its column sums, taken in that GEMM's epilogue
        buf1, db1 = _target(b1)
        du = ops.linear_dgrad(dy, ops.weight_as(w2, dy.dtype, pad=True), gelu_u=u, pad=True, csum=buf1, gelu_d=ctx.gelu_d)
        dw2 = _wgrad(dy, g, w2)
        db2 = None if ctx.defer_bias else _bgrad(dy, b2)
        dx = ops.linear_dgrad(du, ops.weight_as(w1, dy.dtype, pad=True), pad=True) if ctx.needs_input_grad[0] else None
        dw1 = _wgrad(du, x, w1)
        return dx, dw1, db1, dw2, db2, (dy if ctx.has_res else None), None


class SpatialAttnFn(Function):
    """softmax(q k^T / sqrt(d)) v per (frame, head) on packed qkv [BF*P, 3*inner] (module.py:84-91)."""

    @staticmethod
    def forward(ctx, qkv, BF, P, heads, dh, fp8=False):
        out, lse = ops.attn_spatial_fwd(qkv, BF, P, heads, dh, fp8=fp8)
        ctx.save_for_backward(qkv, out, lse)
        ctx.geom = (BF, P, heads, dh)
        ctx.fp8 = fp8
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        qkv, out, lse = ctx.saved_tensors
        return ops.attn_spatial_bwd(qkv, out, dout, lse, *ctx.geom, fp8=ctx.fp8), None, None, None, None, None


class TemporalAttnFn(Function):
    """softmax(q k^T / sqrt(d)) v per (position, head) over frames (module.py:197-205) on ONE packed projection
    qkv [M, 3 * inner] (q | k | v).  diff=True: q and k are differenced over frames inside the kernels (module.py:193:
    TemporalResidualAttention projects the un-differenced LayerNorm output once; exact, to_qk has no bias).  The
    backward returns ONE packed gradient with respect to the un-differenced projection."""

    @staticmethod
    def forward(ctx, qkv, B, F, P, heads, dh, diff):
        inner = heads * dh
        out = ops.attn_temporal_fwd(qkv[:, :2 * inner], qkv[:, 2 * inner:], B, F, P, heads, dh, diff=diff)
        ctx.save_for_backward(qkv)
        ctx.geom = (B, F, P, heads, dh, int(diff))
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        (qkv,) = ctx.saved_tensors
        B, F, P, heads, dh, diff = ctx.geom
        inner = heads * dh
        dqkv, _ = ops.attn_temporal_bwd(qkv[:, :2 * inner], qkv[:, 2 * inner:], dout, B, F, P, heads, dh, diff=diff, packed=True)
        return dqkv, None, None, None, None, None, None


class TokensFn(Function):
    """Token assembly of DSTTr.forward (vivit.py:133-142) on NHWC features [B,T,hw,D]."""

    @staticmethod
    def forward(ctx, feats, space, temporal, pos):
        x = ops.tokens_fwd(feats, space, temporal, pos, pad=True)
        ctx.save_for_backward(space, temporal, pos)
        ctx.geom = tuple(feats.shape)
        return x

    @staticmethod
    @once_differentiable
    def backward(ctx, dx):
        space, temporal, pos = ctx.saved_tensors
        B, T, hw, D = ctx.geom
        (ds, rs), (dt, rt), (dp, rp) = _target(space), _target(temporal), _target(pos)
        dfeats = ops.tokens_bwd(dx, B, T, hw, D, ds, dt, dp, ctx.needs_input_grad[0])
        for hook in list(grad_ready_hooks):
            hook(dx.device)
        return dfeats, rs, rt, rp


class TakeClsFn(Function):
    """x[:, 0, 0] of x viewed as (B, F, P, D): the temporal-token frame's space-token slot (vivit.py:144-146).  The
    backward writes the B rows into a zeroed buffer that keeps the line-aligned row layout."""

    @staticmethod
    def forward(ctx, x, B, F, P):
        ctx.geom = (B, F, P, x.shape[-1], x.dtype)
        return x.reshape(B, F, P, -1)[:, 0, 0].contiguous()

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        B, F, P, D, dtype = ctx.geom
        dx = ops.zeros_rows(B * F * P, D, dtype, g.device)
        dx.view(B, F, P, D)[:, 0, 0] = g
        return dx.view(B, F * P, D), None, None, None


def _new_seed() -> int:
    """63-bit Philox key from torch's default CPU generator (torch.manual_seed makes a run reproducible)"""
    return int(torch.empty((), dtype=torch.int64).random_()) & 0x7fffffffffffffff


def _dropout_fwd(x: Tensor, p: float, seed: int):
    from . import _lib
    x2, ldx = ops.rows(ops._req(x))
    M, D = x2.shape
    y = ops.empty_rows(M, D, x.dtype, x.device, pad=(ldx != D))
    mask = torch.empty((M, D), dtype=torch.uint8, device=x.device)
    _lib.check(_lib.lib().istvt_dropout_fwd(x2.data_ptr(), ldx, y.data_ptr(), y.stride(0) if M > 1 else D, mask.data_ptr(),
                                            M, D, float(p), int(seed), ops.dtype_code(x), ops._stream()), 'istvt_dropout_fwd')
    return y, mask


def _dropout_bwd(dy: Tensor, mask: Tensor, p: float) -> Tensor:
    from . import _lib
    dy2, ldy = ops.rows(dy)
    M, D = dy2.shape
    dx = ops.empty_rows(M, D, dy.dtype, dy.device, pad=(ldy != D))
    _lib.check(_lib.lib().istvt_dropout_bwd(dy2.data_ptr(), ldy, mask.data_ptr(), dx.data_ptr(), dx.stride(0) if M > 1 else D,
                                            M, D, float(p), ops.dtype_code(dy), ops._stream()), 'istvt_dropout_bwd')
    return dx


class DropoutFn(Function):
    """nn.Dropout(p) in training mode on [M, D] rows (module.py:29,31,78,187; models_copy.py:41-44).  The mask comes
    from a Philox4x32-10 stream keyed by `seed`; it cannot reproduce torch's own dropout mask bit for bit -- parity is
    distributional (keep rate, 1/(1-p) scaling) plus exact mask consistency between forward and backward."""

    @staticmethod
    def forward(ctx, x, p, seed):
        y, mask = _dropout_fwd(x, p, seed)
        ctx.save_for_backward(mask)
        ctx.p = float(p)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        (mask,) = ctx.saved_tensors
        return _dropout_bwd(dy, mask, ctx.p), None, None


def dropout(x: Tensor, p: float, training: bool = True) -> Tensor:
    """F.dropout(x, p, training) on the last dim's rows; identity when p == 0 or not training."""
    if p == 0.0 or not training:
        return x
    if not 0.0 <= p < 1.0:
        raise ValueError('dropout probability has to be in [0, 1), got %r' % (p,))
    if x.shape[-1] % 8:
        raise RuntimeError('dropout rows must be a multiple of 8 wide, got %d' % x.shape[-1])
    y = DropoutFn.apply(x.reshape(-1, x.shape[-1]), p, _new_seed())
    return y.reshape(*x.shape[:-1], x.shape[-1])


class AddFn(Function):
    """a + b on [M, D] rows (the residual add that cannot ride in a GEMM epilogue because a Dropout sits in between)."""

    @staticmethod
    def forward(ctx, a, b):
        from . import _lib
        (a2, lda), (b2, ldb) = ops.rows(ops._req(a)), ops.rows(ops._req(b))
        M, D = a2.shape
        out = ops.empty_rows(M, D, a.dtype, a.device, pad=(lda != D))
        _lib.check(_lib.lib().istvt_add(a2.data_ptr(), lda, b2.data_ptr(), ldb, out.data_ptr(), out.stride(0) if M > 1 else D,
                                        M, D, ops.dtype_code(a), ops._stream()), 'istvt_add')
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        return g, g


def add(a: Tensor, b: Tensor) -> Tensor:
    if a.shape != b.shape or a.dtype != b.dtype:
        raise RuntimeError('add: operands must match (%s %s vs %s %s)' % (tuple(a.shape), a.dtype, tuple(b.shape), b.dtype))
    y = AddFn.apply(a.reshape(-1, a.shape[-1]), b.reshape(-1, b.shape[-1]))
    return y.reshape(a.shape)


class FeedForwardDropFn(Function):
    """FeedForward with active dropouts (module.py:26-32 with dropout > 0, training): Linear -> GELU -> Dropout ->
    Linear -> Dropout.  The dropout masks are elementwise factors, so the backward still uses the GEMM whose epilogue
    multiplies by gelu'(u), and applies the first dropout's mask to its result."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, p1, p2, seed1, seed2):
        ctx.gelu_d = GELU_SAVE_DERIV[0]
        u, g = ops.linear_fwd(x, ops.weight_as(w1, x.dtype, pad=True), b1, gelu=True, pad=True, gelu_d=ctx.gelu_d)
        m1 = m2 = None
        if p1 > 0:
            g, m1 = _dropout_fwd(g, p1, seed1)
        y = ops.linear_fwd(g, ops.weight_as(w2, x.dtype, pad=True), b2, None, pad=True)
        if p2 > 0:
            y, m2 = _dropout_fwd(y, p2, seed2)
        ctx.save_for_backward(x, u, g, w1, b1, w2, b2, m1, m2)
        ctx.p = (float(p1), float(p2))
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, u, g, w1, b1, w2, b2, m1, m2 = ctx.saved_tensors
        p1, p2 = ctx.p
        if m2 is not None:
            dy = _dropout_bwd(dy, m2, p2)
        du = ops.linear_dgrad(dy, ops.weight_as(w2, dy.dtype, pad=True), gelu_u=u, pad=True, gelu_d=ctx.gelu_d)
        if m1 is not None:
            du = _dropout_bwd(du, m1, p1)
        dw2 = _wgrad(dy, g, w2)
        db2 = _bgrad(dy, b2)
        dx = ops.linear_dgrad(du, ops.weight_as(w1, dy.dtype, pad=True), pad=True) if ctx.needs_input_grad[0] else None
        dw1 = _wgrad(du, x, w1)
        db1 = _bgrad(du, b1)
        return dx, dw1, db1, dw2, db2, None, None, None, None


class PrependFn(Function):
    """Token assembly of the ablation transformers (vivit.py:60-67 ViViT space stage, :74-75 its temporal stage,
    :183-186 VanillaTr): out[s] = [tok | src[s]] (+ pos[s % period]).  src [S, n, D]; tok (1, 1, D) float32;
    pos float32 [period, pos_rows >= n + 1, D] or None."""

    @staticmethod
    def forward(ctx, src, tok, pos, period):
        from . import _lib
        src = ops._c(ops._req(src))
        S, n, D = src.shape
        if pos is not None and (pos.shape[-1] != D or pos.shape[-2] < n + 1):
            raise RuntimeError('The size of tensor a (%d) must match the size of tensor b (%d) at non-singleton dimension 1'
                               % (n + 1, pos.shape[-2]))
        out = ops.empty_rows(S * (n + 1), D, src.dtype, src.device, True)
        pos_rows = pos.shape[-2] if pos is not None else 0
        _lib.check(_lib.lib().istvt_prepend_fwd(src.data_ptr(), tok.data_ptr(), ops._ptr(pos), out.data_ptr(), out.stride(0),
                                                S, n, D, period, pos_rows, ops.dtype_code(src), ops._stream()), 'istvt_prepend_fwd')
        ctx.save_for_backward(tok, pos)
        ctx.geom = (S, n, D, period, pos_rows)
        return out.view(S, n + 1, D)

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        from . import _lib
        tok, pos = ctx.saved_tensors
        S, n, D, period, pos_rows = ctx.geom
        d2, ldd = ops.rows(dout.reshape(S * (n + 1), D))
        dsrc = torch.empty((S, n, D), dtype=dout.dtype, device=dout.device) if ctx.needs_input_grad[0] else None
        (dt, rt), (dp, rp) = _target(tok), _target(pos)
        lib = _lib.lib()
        ws = torch.empty((lib.istvt_prepend_bwd_ws_rows(S, period, int(dp is not None)), D), dtype=torch.float32, device=dout.device)
        _lib.check(lib.istvt_prepend_bwd(d2.data_ptr(), ldd, ops._ptr(dsrc), dt.data_ptr(), ops._ptr(dp), ws.data_ptr(), S, n, D,
                                         period, pos_rows, ops.dtype_code(d2), ops._stream()), 'istvt_prepend_bwd')
        return dsrc, rt, rp, None


class TakeFirstFn(Function):
    """x[:, 0] of [S, n, D] (the cls rows, vivit.py:71,79,189); backward scatters into a zeroed, line-aligned buffer."""

    @staticmethod
    def forward(ctx, x):
        ctx.geom = (tuple(x.shape), x.dtype)
        return x[:, 0].contiguous()

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        (S, n, D), dtype = ctx.geom
        dx = ops.zeros_rows(S * n, D, dtype, g.device)
        dx.view(S, n, D)[:, 0] = g
        return dx.view(S, n, D)


class SeqMeanFn(Function):
    """x.mean(dim=1) of [S, n, D] (ViViT pool='mean', vivit.py:79)."""

    @staticmethod
    def forward(ctx, x):
        from . import _lib
        S, n, D = x.shape
        x2, ldx = ops.rows(ops._req(x).reshape(S * n, D))
        out = torch.empty((S, D), dtype=x.dtype, device=x.device)
        _lib.check(_lib.lib().istvt_seq_mean_fwd(x2.data_ptr(), ldx, out.data_ptr(), S, n, D, ops.dtype_code(x), ops._stream()),
                   'istvt_seq_mean_fwd')
        ctx.geom = (S, n, D)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        from . import _lib
        S, n, D = ctx.geom
        g = ops._c(g)
        dx = ops.empty_rows(S * n, D, g.dtype, g.device, True)
        _lib.check(_lib.lib().istvt_seq_mean_bwd(g.data_ptr(), dx.data_ptr(), dx.stride(0), S, n, D, ops.dtype_code(g),
                                                 ops._stream()), 'istvt_seq_mean_bwd')
        return dx.view(S, n, D)


class TakeFrameFn(Function):
    """x viewed as (B, F, P, D) -> frame 0 of every clip as (B, P, D) with line-aligned rows (the temporal-token frame:
    the only frame whose last-layer spatial attention reaches the output, vivit.py:144-146).  Backward scatters into a
    zeroed full-size buffer."""

    @staticmethod
    def forward(ctx, x, B, F, P):
        D = x.shape[-1]
        ctx.geom = (B, F, P, D, x.dtype)
        out = ops.empty_rows(B * P, D, x.dtype, x.device)
        out.view(B, P, D).copy_(x.reshape(B, F, P, D)[:, 0])
        return out.view(B, P, D)

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        B, F, P, D, dtype = ctx.geom
        dx = ops.zeros_rows(B * F * P, D, dtype, g.device)
        dx.view(B, F, P, D)[:, 0] = g
        return dx.view(B, F * P, D), None, None, None


def layer_norm(x, gamma, beta, eps=1e-5, fork=False, sink=None):
    return LayerNormFn.apply(x, gamma, beta, eps, fork, sink)


def layer_norm_diff(x, gamma, beta, eps, geom, fork=False, sink=None):
    """-> (y, diff[, x]); see LayerNormDiffFn"""
    return LayerNormDiffFn.apply(x, gamma, beta, eps, geom, fork, sink)


def linear(x: Tensor, weight: Tensor, bias=None, residual=None) -> Tensor:
    """nn.Linear on the last dim of x (any leading shape)."""
    lead = x.shape[:-1]
    x2 = x.reshape(-1, x.shape[-1])
    r2 = residual.reshape(-1, weight.shape[0]) if residual is not None else None
    y = LinearFn.apply(x2, weight, bias, r2)
    return y.view(*lead, weight.shape[0])
