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
        dx = ops.layernorm_bwd(dy, x, mean, rstd, gamma, dg, db, dres=dres, pad=True, dcol=dcol)
        return dx, rg, rb, None, None, None


class LayerNormDiffFn(Function):
    """LayerNorm that also returns the frame difference of its output (module.py:193); fork as in LayerNormFn."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, B, F, P, fork=False, sink=None):
        y, diff, mean, rstd = ops.layernorm_fwd_diff(x, gamma, beta, eps, B, F, P, pad=True)
        ctx.save_for_backward(x, mean, rstd, gamma, beta)
        ctx.geom = (F, P)
        ctx.set_materialize_grads(False)
        ctx.sink = sink
        return (y, diff, x.view_as(x)) if fork else (y, diff)

    @staticmethod
    @once_differentiable
    def backward(ctx, dy, ddiff, dres=None):
        x, mean, rstd, gamma, beta = ctx.saved_tensors
        F, P = ctx.geom
        dg, rg = _target(gamma)
        db, rb = _target(beta)
        if dy is None:
            dy = torch.zeros_like(ddiff if ddiff is not None else x)
        dcol = ctx.sink.buffer() if ctx.sink is not None else None
        dx = ops.layernorm_bwd(dy, x, mean, rstd, gamma, dg, db, dy2=ddiff, dres=dres, F=F, P=P, pad=True, dcol=dcol)
        return dx, rg, rb, None, None, None, None, None, None


class FrameDiffFn(Function):
    """residual = cat(x[:, :2], x[:, 2:] - x[:, 1:-1]) over frames (module.py:193)."""

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
_overlap = {'on': os.environ.get('ISTVT_WGRAD_STREAM', '1') != '0', 'streams': {}, 'pending': {}, 'keep': {}}


def set_wgrad_overlap(on: bool):
    _overlap['on'] = bool(on)


def join_side_stream(dev=None):
    """Make the current backward's main stream wait for the side stream now (instead of at the end of backward): what a
    caller needs before it hands weight gradients to a collective mid-backward."""
    if dev is None:
        dev = torch.cuda.current_device()
    _join_side(dev)


# Callables run by TokensFn.backward once its own kernels are enqueued.  Token assembly is the first operation of the
# transformer, so at that point every gradient of the transformer (98.8 % of the bucket) has been enqueued -- on the main
# stream or the side stream -- while the whole stem backward is still to come: parallel.GradBucket hangs the
# all-reduce of that part of the bucket here to overlap it with the stem backward.
grad_ready_hooks = []


def _join_side(dev):
    main = _overlap['pending'].pop(dev, None)
    if main is not None:
        main.wait_stream(_overlap['streams'][dev])
    # The operands of the side-stream launches were kept alive until here instead of being handed to the allocator with
    # record_stream(): blocks parked behind side-stream events made the caching allocator grow by a run-dependent
    # 16-56 GB at C2 (reserved 47-87 GB against 31 GB single-stream).  Released after the join they go back to the main
    # stream's pool in order: ~10 GB more at the peak, the same every run.
    _overlap['keep'].pop(dev, None)


def _wgrad(dy, x, weight):
    buf, ret = _target(weight)
    out = buf.view(weight.shape[0], -1)
    if not (_overlap['on'] and ret is None and dy.is_cuda):
        ops.linear_wgrad(dy, x, out=out)
        return ret
    dev = dy.device.index
    side = _overlap['streams'].get(dev)
    if side is None:
        side = _overlap['streams'][dev] = torch.cuda.Stream(device=dy.device, priority=int(os.environ.get('ISTVT_WGRAD_PRIO', '0')))
    main = torch.cuda.current_stream(dy.device)
    if dev not in _overlap['pending']:
        _overlap['pending'][dev] = main
        torch.autograd.Variable._execution_engine.queue_callback(lambda: _join_side(dev))
    side.wait_stream(main)
    with torch.cuda.stream(side):
        ops.linear_wgrad(dy, x, out=out)
    _overlap['keep'].setdefault(dev, []).append((dy, x))     # alive until the join (see _join_side)
    return None


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


class FeedForwardFn(Function):
    """Linear -> exact GELU -> Linear (+ residual) (FeedForward, module.py:23-34)."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, residual, defer_bias=False):
        u, g = ops.linear_fwd(x, ops.weight_as(w1, x.dtype, pad=True), b1, gelu=True, pad=True)
        y = ops.linear_fwd(g, ops.weight_as(w2, x.dtype, pad=True), b2, residual, pad=True)
        ctx.save_for_backward(x, u, g, w1, b1, w2, b2)
        ctx.has_res = residual is not None
        ctx.defer_bias = bool(defer_bias)           # b2's gradient comes from the next LayerNorm's backward (BiasSink)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, u, g, w1, b1, w2, b2 = ctx.saved_tensors
        du = ops.linear_dgrad(dy, ops.weight_as(w2, dy.dtype, pad=True), gelu_u=u, pad=True)       # (dy W2) * gelu'(u)
        dw2 = _wgrad(dy, g, w2)
        db2 = None if ctx.defer_bias else _bgrad(dy, b2)
        dx = ops.linear_dgrad(du, ops.weight_as(w1, dy.dtype, pad=True), pad=True) if ctx.needs_input_grad[0] else None
        dw1 = _wgrad(du, x, w1)
        db1 = _bgrad(du, b1)
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
    """softmax(q k^T / sqrt(d)) v per (position, head) over frames (module.py:197-205)."""

    @staticmethod
    def forward(ctx, qk, v, B, F, P, heads, dh):
        out = ops.attn_temporal_fwd(qk, v, B, F, P, heads, dh)
        ctx.save_for_backward(qk, v)
        ctx.geom = (B, F, P, heads, dh)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        qk, v = ctx.saved_tensors
        dqk, dv = ops.attn_temporal_bwd(qk, v, dout, *ctx.geom)
        return dqk, dv, None, None, None, None, None


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


def layer_norm(x, gamma, beta, eps=1e-5, fork=False, sink=None):
    return LayerNormFn.apply(x, gamma, beta, eps, fork, sink)


def linear(x: Tensor, weight: Tensor, bias=None, residual=None) -> Tensor:
    """nn.Linear on the last dim of x (any leading shape)."""
    lead = x.shape[:-1]
    x2 = x.reshape(-1, x.shape[-1])
    r2 = residual.reshape(-1, weight.shape[0]) if residual is not None else None
    y = LinearFn.apply(x2, weight, bias, r2)
    return y.view(*lead, weight.shape[0])
