"""Xception building blocks executed by themselves (reference network/xception.py:39-49 SeparableConv2d.forward,
:52-101 Block.forward, :185-190 the exit flow's conv3/bn3/relu/conv4/bn4, :208-213 logits) on the stem's HIP kernels.

``istvt_amd.stem.StemFn`` stays the fast path of ``Xception.low_level_features`` (one Function over conv1..block3 with
its block-1 special fusions).  Everything else -- a ``Block`` called on its own, blocks 4-12, the exit flow -- runs
through ``RepChainFn``: a chain of [ReLU?] SeparableConv2d BatchNorm units with one of three tails

    'pool'   MaxPool2d(3,2,1) of the last BatchNorm + BatchNorm(1x1 stride-2 conv(input))        (Block, strides = 2)
    'add'    last BatchNorm + (BatchNorm(1x1 conv(input)) or the input itself)                    (Block, strides = 1)
    'plain'  the last BatchNorm's output                                                         (exit flow)

with the same fusion plan as the stem: every BatchNorm is {statistics, finalize} and its apply (+ReLU) rides in the
consumer's load; in backward the ReLU masks, the skip-path gradient add and the BatchNorm-backward sums ride in the
depthwise input-gradient kernel's epilogue.  Activations are NHWC ``[frames*H*W, C]`` in float32 or bfloat16.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import _lib, ops
from . import stem as _stem
from .ops import _c, _req, _stream, dtype_code
from .stem import bn_apply, bn_backward, bn_forward_stats, dwconv, dwconv_wgrad, new_stats, pointwise_bn

Tensor = torch.Tensor


def nhwc(x: Tensor) -> Tensor:
    """(n, C, H, W) tensor of any strides -> contiguous (n, H, W, C); free when x is channels-last in memory (which is
    what every module of this package returns)."""
    return _c(x.permute(0, 2, 3, 1))


def nchw_view(y: Tensor) -> Tensor:
    """(n, H, W, C) contiguous -> the same memory seen as (n, C, H, W) (channels-last strides): the reference's shape
    convention without a copy."""
    return y.permute(0, 3, 1, 2)


def _tap_major(w: Tensor) -> Tensor:
    """depthwise weight (C, 1, 3, 3) -> float32 [9][C] (the layout istvt_dwconv3x3 takes), cached per parameter version"""
    return _stem.tap_major(w)


def _fused_target(q: Tensor, shape):
    """the view of q.grad the kernels accumulate into, when q was registered with GradBucket(fuse_accumulate=True)"""
    if getattr(q, '_istvt_fused_grad', False) and q.grad is not None and q.grad.is_contiguous():
        return q.grad.view(shape)
    return None


# ------------------------------------------------------------------------------------------ SeparableConv2d
class SepConvFn(Function):
    """y = pointwise(depthwise3x3(x)) (xception.py:46-49), NHWC [M, Cin] -> [M, Cout]."""

    @staticmethod
    def forward(ctx, x, wdw, wpw, Fr, H, W):
        _req(x, 'SeparableConv2d input')
        cin, cout = wdw.shape[0], wpw.shape[0]
        w9 = _tap_major(wdw)
        wp = ops.weight_as(wpw, x.dtype, pad=True)
        d = dwconv(x, w9, Fr, H, W, cin)
        u = ops.linear_fwd(d, wp, blocked=False)
        ctx.save_for_backward(x, d, wdw, wpw)
        ctx.geom = (Fr, H, W, cin, cout)
        return u

    @staticmethod
    @once_differentiable
    def backward(ctx, du):
        x, d, wdw, wpw = ctx.saved_tensors
        Fr, H, W, cin, cout = ctx.geom
        du = _c(du)
        wp = ops.weight_as(wpw, du.dtype, pad=True)
        dd = ops.linear_dgrad(du, wp, blocked=False)
        t = _fused_target(wpw, (cout, cin))
        gpw = ops.linear_wgrad(du, d, out=t)
        t2 = _fused_target(wdw, (cin, 9))
        gdw = dwconv_wgrad(x, dd, Fr, H, W, cin, None, False, out=t2)
        dx = dwconv(dd, _tap_major(wdw), Fr, H, W, cin, flip=True) if ctx.needs_input_grad[0] else None
        return (dx, None if t2 is not None else gdw.view(wdw.shape), None if t is not None else gpw.view(wpw.shape),
                None, None, None)


# ------------------------------------------------------------------------------------------ the unit chain
class ChainSpec:
    """Static description of a chain: units [(cin, cout)], whether the first unit is preceded by a ReLU, the tail, and
    the parameter / buffer order RepChainFn.apply expects.

    params:  for each unit [dw weight, pw weight, bn weight, bn bias], then (has_skip) [skip weight, skipbn weight, bias]
    buffers: for each unit [running_mean, running_var], then (has_skip) the skip BatchNorm's pair"""

    def __init__(self, units: Sequence[Tuple[int, int]], start_with_relu: bool, tail: str, has_skip: bool):
        assert tail in ('pool', 'add', 'plain')
        assert tail != 'pool' or has_skip, 'a strided Block always has the 1x1 skip convolution'
        assert tail != 'plain' or not has_skip
        self.units = list(units)
        self.start_with_relu = bool(start_with_relu)
        self.tail = tail
        self.has_skip = bool(has_skip)
        self.cin = self.units[0][0]
        self.cout = self.units[-1][1]


class RepChainFn(Function):
    """out = tail(chain(inp)) -- see the module docstring.  inp: NHWC [Fr*H*W, Cin]."""

    @staticmethod
    def forward(ctx, inp, spec: ChainSpec, Fr, H, W, training, buffers, *params):
        _req(inp, 'Block input')
        inp = _c(inp)
        dtype, dev = inp.dtype, inp.device
        L = _lib.lib()
        dtc = dtype_code(inp)
        nu = len(spec.units)
        M = Fr * H * W
        if inp.shape != (M, spec.cin):
            raise RuntimeError('Block expects %d input channels on a %dx%dx%d map, got %s' % (spec.cin, Fr, H, W, tuple(inp.shape)))
        units = []
        X, in_bn = inp, None
        for i, (cin, cout) in enumerate(spec.units):
            wdw, wpw, g, b = params[4 * i: 4 * i + 4]
            rm, rv = buffers[2 * i], buffers[2 * i + 1]
            relu = spec.start_with_relu if i == 0 else True
            w9 = _tap_major(wdw)
            wp = ops.weight_as(wpw, dtype, pad=True)
            d = dwconv(X, w9, Fr, H, W, cin, in_bn=in_bn, in_relu=relu)
            u, bn = pointwise_bn(d, wp, M, cout, g, b, rm, rv, training)
            units.append(dict(X=X, in_bn=in_bn, relu=relu, d=d, u=u, bn=bn, w9=w9, wp=wp, cin=cin, cout=cout))
            X, in_bn = u, bn
        sk = None
        cout = spec.cout
        if spec.has_skip:
            wsk, gs, bs = params[4 * nu: 4 * nu + 3]
            rms, rvs = buffers[2 * nu], buffers[2 * nu + 1]
            ws = ops.weight_as(wsk, dtype, pad=True)
            if spec.tail == 'pool':
                Hs, Ws = (H - 1) // 2 + 1, (W - 1) // 2 + 1
                xs = torch.empty((Fr * Hs * Ws, spec.cin), dtype=dtype, device=dev)
                _lib.check(L.istvt_subsample2(inp.data_ptr(), xs.data_ptr(), Fr, H, W, spec.cin, dtc, _stream()), 'istvt_subsample2')
            else:
                Hs, Ws, xs = H, W, inp
            uS, bnS = pointwise_bn(xs, ws, Fr * Hs * Ws, cout, gs, bs, rms, rvs, training)
            sk = dict(xs=xs, uS=uS, bnS=bnS, ws=ws, Hs=Hs, Ws=Ws)
        amax = None
        if spec.tail == 'pool':
            out = torch.empty((Fr * sk['Hs'] * sk['Ws'], cout), dtype=dtype, device=dev)
            amax = torch.empty(out.shape, dtype=torch.uint8, device=dev)
            _lib.check(L.istvt_pool_add_fwd(X.data_ptr(), in_bn.ptr(), sk['uS'].data_ptr(), sk['bnS'].ptr(), out.data_ptr(),
                                            amax.data_ptr(), Fr, H, W, cout, dtc, _stream()), 'istvt_pool_add_fwd')
        elif spec.tail == 'add':
            if not spec.has_skip and spec.cin != cout:
                raise RuntimeError('identity skip needs in_filters == out_filters')
            out = torch.empty((M, cout), dtype=dtype, device=dev)
            _lib.check(L.istvt_bn_add_fwd(X.data_ptr(), in_bn.ptr(), (sk['uS'] if sk else inp).data_ptr(),
                                          sk['bnS'].ptr() if sk else None, out.data_ptr(), M, cout, dtc, _stream()),
                       'istvt_bn_add_fwd')
        else:
            out = bn_apply(X, in_bn, M, cout, False)
        ctx.sv = dict(spec=spec, Fr=Fr, H=H, W=W, units=units, sk=sk, amax=amax, inp=inp, params=params, training=training)
        ctx.need_dx = ctx.needs_input_grad[0]       # (not inp.requires_grad: a contiguous copy made in here never requires grad)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        sv = ctx.sv
        if sv is None:
            raise RuntimeError('RepChainFn: the saved activations were released by the first backward pass '
                               '(retain_graph=True is not supported for Xception blocks)')
        spec, Fr, H, W, units, sk, params, training = (sv[k] for k in ('spec', 'Fr', 'H', 'W', 'units', 'sk', 'params', 'training'))
        L = _lib.lib()
        dout = _c(dout)
        dtype, dev = dout.dtype, dout.device
        dtc = dtype_code(dout)
        nu = len(units)
        M = Fr * H * W
        grads: List[Optional[Tensor]] = [None] * len(params)

        def bn_bwd(dz, u, st, gi, Mx, C, stats=None):
            g, b = params[gi], params[gi + 1]
            tg, tb = _fused_target(g, (C,)), _fused_target(b, (C,))
            du, dg, db = bn_backward(dz, u, st, g, Mx, C, stats=stats, dg=tg, db=tb, training=training)
            grads[gi] = None if tg is not None else dg
            grads[gi + 1] = None if tb is not None else db
            return du

        def lin_wgrad(wi, dyv, xv):
            q = params[wi]
            t = _fused_target(q, (q.shape[0], -1))
            r = ops.linear_wgrad(dyv, xv, out=t)
            grads[wi] = None if t is not None else r.view(q.shape)

        # ---- skip path
        dskip, skip_full = None, False                  # gradient reaching the block input through the skip path
        if spec.has_skip:
            Ms = Fr * sk['Hs'] * sk['Ws']
            duS = bn_bwd(dout, sk['uS'], sk['bnS'], 4 * nu + 1, Ms, spec.cout)
            lin_wgrad(4 * nu, duS, sk['xs'])
            if ctx.need_dx:
                dskip = ops.linear_dgrad(duS, sk['ws'], blocked=False)
                skip_full = spec.tail != 'pool'
        elif spec.tail == 'add':
            dskip, skip_full = dout, True
        # ---- rep path, last unit first
        if spec.tail == 'pool':
            dz = torch.empty((M, spec.cout), dtype=dtype, device=dev)
            # train mode: the last unit's BatchNorm-backward sums are taken by the pooling backward as it writes dz
            last = units[nu - 1]
            stats = new_stats(spec.cout, dev) if training else None
            _lib.check(L.istvt_pool_bwd(dout.data_ptr(), sv['amax'].data_ptr(), dz.data_ptr(), Fr, H, W, spec.cout,
                                        last['u'].data_ptr() if training else None, last['bn'].ptr() if training else None,
                                        stats[0, 0].data_ptr() if training else None,
                                        stats[0, 1].data_ptr() if training else None, dtc, _stream()), 'istvt_pool_bwd')
        else:
            dz = dout
            stats = None
        dinp = None
        for i in reversed(range(nu)):
            un = units[i]
            cin, cout = un['cin'], un['cout']
            du = bn_bwd(dz, un['u'], un['bn'], 4 * i + 2, M, cout, stats=stats)
            del dz
            lin_wgrad(4 * i + 1, du, un['d'])
            dd = ops.linear_dgrad(du, un['wp'], blocked=False)
            del du
            q = params[4 * i]
            t = _fused_target(q, (cin, 9))
            r = dwconv_wgrad(un['X'], dd, Fr, H, W, cin, un['in_bn'], un['relu'], out=t)
            grads[4 * i] = None if t is not None else r.view(q.shape)
            if i > 0:
                # through the ReLU and into the previous BatchNorm: mask by relu'(bn(u_prev)), its backward sums on the way
                prev = units[i - 1]
                stats = new_stats(cin, dev)
                dz = dwconv(dd, un['w9'], Fr, H, W, cin, flip=True, msrc=prev['u'], m_bn=prev['bn'], mask_pre=True, stats=stats)
            elif ctx.need_dx:
                Ha, Wa = (H, W) if skip_full else ((H - 1) // 2 + 1, (W - 1) // 2 + 1)
                dinp = _dw_input_grad(dd, un['w9'], Fr, H, W, cin, sv['inp'] if un['relu'] else None, dskip, Ha, Wa)
            del dd
        ctx.sv = None
        return (dinp, None, None, None, None, None, None, *grads)


def _dw_input_grad(dd, w9, Fr, H, W, C, relu_src, addsrc, Ha, Wa):
    """d(block input) = relu'(input) * d(rep path) + d(skip path) -- the skip path at full resolution (stride-1 blocks)
    or scattered to the even pixels (stride-2 blocks)."""
    out = torch.empty((Fr * H * W, C), dtype=dd.dtype, device=dd.device)
    _lib.check(_lib.lib().istvt_dwconv3x3(
        dd.data_ptr(), w9.data_ptr(), out.data_ptr(), Fr, H, W, C, None, 0, 1,
        relu_src.data_ptr() if relu_src is not None else None, None, int(relu_src is not None), 0,
        addsrc.data_ptr() if addsrc is not None else None, Ha, Wa, None, None, dtype_code(dd), _stream()), 'istvt_dwconv3x3')
    return out


# ------------------------------------------------------------------------------------------ logits head
class ReluAvgPoolFn(Function):
    """adaptive_avg_pool2d(relu(x), (1, 1)).view(n, -1) on NHWC features [n, HW, C] (xception.py:208-213)."""

    @staticmethod
    def forward(ctx, x, n, HW, relu=True):
        x = _c(_req(x))
        C = x.shape[-1]
        out = torch.empty((n, C), dtype=x.dtype, device=x.device)
        _lib.check(_lib.lib().istvt_relu_avgpool_fwd(x.data_ptr(), out.data_ptr(), n, HW, C, int(relu), dtype_code(x), _stream()),
                   'istvt_relu_avgpool_fwd')
        ctx.save_for_backward(x)
        ctx.geom = (n, HW, C, relu)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        (x,) = ctx.saved_tensors
        n, HW, C, relu = ctx.geom
        dout = _c(dout)
        dx = torch.empty_like(x)
        _lib.check(_lib.lib().istvt_relu_avgpool_bwd(x.data_ptr(), dout.data_ptr(), dx.data_ptr(), n, HW, C, int(relu),
                                                     dtype_code(x), _stream()), 'istvt_relu_avgpool_bwd')
        return dx, None, None, None
