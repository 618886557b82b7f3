"""Xception entry flow (conv1 .. block3, reference network/xception.py:193-206) as one
autograd Function whose forward and backward are sequences of HIP launches.

Layout: activations are NHWC ``[frames, H, W, C]`` in the compute dtype, so the stem's output
``[B*T, h, w, 728]`` already is the transformer's ``(b, t, h*w, c)`` token layout (no permute).

Fusion plan: train-mode BatchNorm needs a grid-wide reduction, so every BN is
split into {statistics kernel -> tiny finalize kernel} and its apply(+ReLU) is folded into the
consumer's load: the depthwise conv's LDS tile load, conv2's im2col, or the maxpool+skip-add
kernel.  Only bn2's output is materialised (it has two consumers).  In backward the ReLU masks,
the stride-2 scatter of the skip-path gradient and the BN-backward reductions ride in the
depthwise input-gradient kernel's epilogue.
"""
from __future__ import annotations

from typing import List, Optional

import os

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import _lib, ops
from .ops import _c, _ptr, _req, _stream, dtype_code

Tensor = torch.Tensor
BN_EPS = 1e-5
BN_MOMENTUM = 0.1


def out_side(side: int) -> int:
    s = (side - 3) // 2 + 1 - 2
    for _ in range(3):
        s = (s - 1) // 2 + 1
    return s


# ------------------------------------------------------------------------------------------ launch wrappers
class BNState:
    """Per-BatchNorm statistics produced by the forward and reused by the backward."""
    __slots__ = ('pack', 'mean', 'rstd', 'scale', 'beta')

    def __init__(self, C, device):
        # one [4][C] pack {mean, rstd, scale = gamma*rstd, beta}: kernels take its base pointer and
        # apply z = (u - mean) * scale + beta
        self.pack = torch.empty((4, C), dtype=torch.float32, device=device)
        self.mean, self.rstd, self.scale, self.beta = self.pack[0], self.pack[1], self.pack[2], self.pack[3]

    def ptr(self):
        return self.pack.data_ptr()


_REPLICAS = None


# Statistics accumulators are consumed (reduced, finalised) by the launches that directly follow their producer, so
# they carry nothing from one training step to the next: instead of one torch.zeros() per accumulator (34 fill launches
# per step at C2, each shorter than the ~5 us a launch occupies the queue for) they are cut from ONE arena that
# stats_arena_reset() zeroes with a single fill at the start of a step.  Sized by the previous step's demand; anything
# that does not fit (first step, other models on the same device) falls back to its own zeroed tensor.
_arena = {'buf': None, 'off': 0, 'need': 0}


def stats_arena_reset(device):
    a = _arena
    a['need'] = max(a['need'], a['off'])
    a['off'] = 0
    if a['need'] == 0:
        return
    buf = a['buf']
    if buf is None or buf.device != torch.device(device) or buf.numel() < a['need']:
        a['buf'] = torch.zeros((a['need'],), dtype=torch.float64, device=device)
    else:
        buf.zero_()


def new_stats(C: int, device) -> Tensor:
    """zeroed per-channel statistics accumulator double[R][2][C]; kernels add into replica
    (workgroup % R) to spread same-address atomics, reduce_stats() folds them into replica 0."""
    global _REPLICAS
    if _REPLICAS is None:
        fn = _lib.lib().istvt_stats_replicas
        _REPLICAS = int(fn())
    a = _arena
    n = _REPLICAS * 2 * C
    off = a['off']
    a['off'] = off + n
    buf = a['buf']
    if buf is not None and buf.device == torch.device(device) and off + n <= buf.numel():
        return buf[off:off + n].view(_REPLICAS, 2, C)
    return torch.zeros((_REPLICAS, 2, C), dtype=torch.float64, device=device)


def reduce_stats(acc: Tensor, C: int):
    _lib.check(_lib.lib().istvt_stats_reduce(acc.data_ptr(), C, _stream()), 'istvt_stats_reduce')


def bn_forward_stats(u: Tensor, M: int, C: int, gamma: Tensor, beta: Tensor, rmean: Tensor, rvar: Tensor,
                     training: bool, acc: Optional[Tensor] = None) -> BNState:
    """acc: statistics already accumulated (not yet reduced) by the producer of u -- the 1x1-convolution GEMM's epilogue
    (pointwise_bn) -- instead of a pass over u here."""
    st = BNState(C, u.device)
    L = _lib.lib()
    if training:
        if acc is None:
            acc = new_stats(C, u.device)
            with ops.prof('bn_stats', M * C * u.element_size()):
                _lib.check(L.istvt_bn_stats(u.data_ptr(), acc[0, 0].data_ptr(), acc[0, 1].data_ptr(), M, C, dtype_code(u),
                                            _stream()), 'istvt_bn_stats')
        reduce_stats(acc, C)
        s0, s1 = acc[0, 0].data_ptr(), acc[0, 1].data_ptr()
    else:
        s0 = s1 = None
    _lib.check(L.istvt_bn_finalize(s0, s1, float(M), gamma.data_ptr(), beta.data_ptr(), rmean.data_ptr(), rvar.data_ptr(),
                                   BN_MOMENTUM, BN_EPS, st.ptr(), C, int(training), int(training), _stream()),
               'istvt_bn_finalize')
    return st


def pointwise_bn(x: Tensor, w: Tensor, M: int, C: int, gamma: Tensor, beta: Tensor, rmean: Tensor, rvar: Tensor,
                 training: bool):
    """u = x @ w.T (a 1x1 convolution on NHWC rows) and the BatchNorm that follows it (xception.py:44,57 -> :58,69,75):
    in train mode the batch statistics come out of the GEMM's epilogue where the kernel supports it (bf16, the
    persistent NT kernel), otherwise from the separate istvt_bn_stats pass.  -> (u, BNState)"""
    acc = None
    if training and ops.stats_fusable(x, w):
        acc = new_stats(C, x.device)
        u = ops.linear_fwd(x, w, stats=acc, blocked=False)
    else:
        u = ops.linear_fwd(x, w, blocked=False)
    return u, bn_forward_stats(u, M, C, gamma, beta, rmean, rvar, training, acc=acc)


def bn_apply(u: Tensor, st: BNState, M: int, C: int, relu: bool) -> Tensor:
    y = torch.empty_like(u)
    _lib.check(_lib.lib().istvt_bn_apply(u.data_ptr(), st.ptr(), y.data_ptr(), M, C, int(relu), dtype_code(u), _stream()),
               'istvt_bn_apply')
    return y


def bn_backward(dz: Tensor, u: Tensor, st: BNState, gamma: Tensor, M: int, C: int, stats: Optional[Tensor] = None,
                dg: Optional[Tensor] = None, db: Optional[Tensor] = None, training: bool = True):
    """-> (du, dgamma, dbeta).  training=False: the pack holds running statistics (constants), so du = gamma*rstd*dz
    while dgamma / dbeta keep their sums.  `stats` = new_stats() buffer already accumulated (not yet reduced) by a
    fused producer (the depthwise input-gradient kernel).  dg / db: float32 [C] buffers to accumulate into
    (a parameter's .grad); fresh zero tensors otherwise."""
    L = _lib.lib()
    if stats is None:
        stats = new_stats(C, u.device)
        with ops.prof('bn_bwd_stats', 2 * M * C * u.element_size()):
            _lib.check(L.istvt_bn_bwd_stats(dz.data_ptr(), u.data_ptr(), st.ptr(),
                                            stats[0, 0].data_ptr(), stats[0, 1].data_ptr(), M, C, dtype_code(u), _stream()),
                       'istvt_bn_bwd_stats')
    reduce_stats(stats, C)
    du = torch.empty_like(u)
    if dg is None:
        dg = torch.zeros((C,), dtype=torch.float32, device=u.device)
    if db is None:
        db = torch.zeros((C,), dtype=torch.float32, device=u.device)
    with ops.prof('bn_bwd_apply', 3 * M * C * u.element_size()):
        _lib.check(L.istvt_bn_bwd_apply(dz.data_ptr(), u.data_ptr(), st.ptr(), gamma.data_ptr(),
                                        stats[0, 0].data_ptr(), stats[0, 1].data_ptr(), du.data_ptr(), dg.data_ptr(),
                                        db.data_ptr(), M, C, int(training), dtype_code(u), _stream()), 'istvt_bn_bwd_apply')
    return du, dg, db


def dwconv(x: Tensor, w9: Tensor, Fr: int, H: int, W: int, C: int, *, in_bn: Optional[BNState] = None,
           in_relu: bool = False, flip: bool = False, msrc: Optional[Tensor] = None, m_bn: Optional[BNState] = None,
           mask_pre: bool = False, mask_post: bool = False, addsrc: Optional[Tensor] = None,
           stats: Optional[Tensor] = None) -> Tensor:
    """stats ([2][C] fp64, accumulated): fused BatchNorm-backward sums of the OUTPUT w.r.t. `m_bn`."""
    out = torch.empty((Fr * H * W, C), dtype=x.dtype, device=x.device)
    Ha, Wa = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    nel = Fr * H * W * C
    nbytes = (2 * nel + (nel if msrc is not None else 0) + (nel // 4 if addsrc is not None else 0)) * x.element_size()
    with ops.prof('dwconv3x3', nbytes, 18.0 * nel):
        _lib.check(_lib.lib().istvt_dwconv3x3(
            x.data_ptr(), w9.data_ptr(), out.data_ptr(), Fr, H, W, C,
            in_bn.ptr() if in_bn else None, int(in_relu), int(flip),
            _ptr(msrc), m_bn.ptr() if m_bn else None, int(mask_pre), int(mask_post),
            _ptr(addsrc), Ha, Wa,
            stats[0, 0].data_ptr() if stats is not None else None, stats[0, 1].data_ptr() if stats is not None else None,
            dtype_code(x), _stream()), 'istvt_dwconv3x3')
    return out


def dwconv_wgrad(x: Tensor, dout: Tensor, Fr: int, H: int, W: int, C: int, in_bn: Optional[BNState], in_relu: bool,
                 out: Optional[Tensor] = None) -> Tensor:
    """out: float32 [C][9] buffer to accumulate into (a depthwise weight's .grad viewed (C, 9))."""
    dw = out if out is not None else torch.zeros((C, 9), dtype=torch.float32, device=x.device)
    lib = _lib.lib()
    ws = torch.empty((lib.istvt_dwconv3x3_wgrad_ws_elems(Fr, H, W, C),), dtype=torch.float32, device=x.device)
    with ops.prof('dwconv3x3_wgrad', 2 * Fr * H * W * C * x.element_size()):
        _lib.check(lib.istvt_dwconv3x3_wgrad(x.data_ptr(), in_bn.ptr() if in_bn else None, int(in_relu), dout.data_ptr(),
                                             dw.data_ptr(), ws.data_ptr(), ws.numel(), Fr, H, W, C, dtype_code(x), _stream()),
                   'istvt_dwconv3x3_wgrad')
    return dw


def tap_major(w: Tensor) -> Tensor:
    """depthwise weight (C, 1, 3, 3) -> float32 [9][C] (the layout istvt_dwconv3x3 takes), cached until the parameter
    changes and rebuilt behind the optimizer step (ops.derived)"""
    return ops.derived((id(w), 'tap'), w, lambda q: q.detach().reshape(q.shape[0], 9).t().contiguous())


def _conv1_weight(w: Tensor, dtype) -> Tensor:
    """(32,3,3,3) [co][ci][dy][dx] -> [co][(dy,dx,ci) padded to 32] in the compute dtype."""
    def build(q):
        w2 = q.detach().permute(0, 2, 3, 1).reshape(q.shape[0], 27)
        return ops.cast(torch.nn.functional.pad(w2, (0, 5)).contiguous(), dtype)
    return ops.derived((id(w), 'conv1', dtype), w, build)


def _conv2_weight(w: Tensor, dtype) -> Tensor:
    return ops.derived((id(w), 'conv2', dtype), w,
                       lambda q: ops.cast(q.detach().permute(0, 2, 3, 1).reshape(q.shape[0], -1).contiguous(), dtype))


# ------------------------------------------------------------------------------------------ the Function
BLOCKS = (('block1', 64, 128, False), ('block2', 128, 256, True), ('block3', 256, 728, True))
# parameter order of StemFn.apply (after x): see param_names()


def param_names() -> List[str]:
    """State-dict names (relative to the Xception module) of the parameters on the path, in
    the order StemFn expects them."""
    names = ['conv1.weight', 'bn1.weight', 'bn1.bias', 'conv2.weight', 'bn2.weight', 'bn2.bias']
    for name, cin, cout, relu in BLOCKS:
        i0 = 1 if relu else 0
        names += ['%s.skip.weight' % name, '%s.skipbn.weight' % name, '%s.skipbn.bias' % name,
                  '%s.rep.%d.conv1.weight' % (name, i0), '%s.rep.%d.pointwise.weight' % (name, i0),
                  '%s.rep.%d.weight' % (name, i0 + 1), '%s.rep.%d.bias' % (name, i0 + 1),
                  '%s.rep.%d.conv1.weight' % (name, i0 + 3), '%s.rep.%d.pointwise.weight' % (name, i0 + 3),
                  '%s.rep.%d.weight' % (name, i0 + 4), '%s.rep.%d.bias' % (name, i0 + 4)]
    return names


def bn_names() -> List[str]:
    """BatchNorm module names in the order their (running_mean, running_var) buffers are passed."""
    names = ['bn1', 'bn2']
    for name, cin, cout, relu in BLOCKS:
        i0 = 1 if relu else 0
        names += ['%s.skipbn' % name, '%s.rep.%d' % (name, i0 + 1), '%s.rep.%d' % (name, i0 + 4)]
    return names


class StemFn(Function):
    """y = Xception.low_level_features(x).  x: (Fr,3,S,S) float32 NCHW; y: (Fr,h,w,728) NHWC in
    `dtype`.  `buffers` = [running_mean, running_var] * 11 in bn_names() order (updated in place
    when training)."""

    @staticmethod
    def forward(ctx, x, dtype, training, buffers, *params):
        _req(x, 'input clip')
        if x.dtype != torch.float32:
            raise TypeError('stem input must be float32, got %s' % x.dtype)
        x = _c(x)
        Fr, cin, S, S2 = x.shape
        if cin != 3 or S != S2:
            raise RuntimeError('stem expects (frames, 3, S, S) input, got %s' % (tuple(x.shape),))
        L = _lib.lib()
        dev = x.device
        P = dict(zip(param_names(), params))
        bufs = {n: (buffers[2 * i], buffers[2 * i + 1]) for i, n in enumerate(bn_names())}
        sv = {}                                             # everything backward needs

        def bn(name, u, M, C):
            rm, rv = bufs[name]
            return bn_forward_stats(u, M, C, P[name + '.weight'], P[name + '.bias'], rm, rv, training)

        def pw_bn(name, xin, w, M, C):             # 1x1 convolution + its BatchNorm, statistics from the GEMM epilogue
            rm, rv = bufs[name]
            return pointwise_bn(xin, w, M, C, P[name + '.weight'], P[name + '.bias'], rm, rv, training)

        # conv1 (3->32, 3x3, s2, p0) as im2col + GEMM
        H1 = (S - 3) // 2 + 1
        M1 = Fr * H1 * H1
        # conv1 directly from the fp32 NCHW clip (one thread per output pixel); w1 is the GEMM form for the backward
        w1 = _conv1_weight(P['conv1.weight'], dtype)
        if dtype == torch.bfloat16:
            u1 = torch.empty((M1, 32), dtype=dtype, device=dev)
            _lib.check(L.istvt_conv1_fwd(x.data_ptr(), P['conv1.weight'].detach().contiguous().data_ptr(), u1.data_ptr(),
                                         Fr, S, ops._DT[dtype], _stream()), 'istvt_conv1_fwd')
        else:
            # fp32 parity mode keeps im2col + GEMM: with the golden recipe's structured weights a different fp32
            # summation order flips ReLU masks at |z| ~ 1e-7 and moves early-layer gradients by 1e-2
            col1 = torch.empty((M1, 32), dtype=dtype, device=dev)
            _lib.check(L.istvt_im2col_conv1(x.data_ptr(), col1.data_ptr(), Fr, S, ops._DT[dtype], _stream()), 'istvt_im2col_conv1')
            u1 = ops.linear_fwd(col1, w1, blocked=False)
            del col1
        bn1 = bn('bn1', u1, M1, 32)
        # conv2 (32->64, 3x3, p0): im2col applies bn1 + ReLU on load
        H2 = H1 - 2
        M2 = Fr * H2 * H2
        w2 = _conv2_weight(P['conv2.weight'], dtype)
        if dtype == torch.bfloat16:         # MFMA convolution straight from u1 (bn1 + ReLU in registers)
            u2 = torch.empty((M2, 64), dtype=dtype, device=dev)
            _lib.check(L.istvt_conv2_fwd(u1.data_ptr(), bn1.ptr(), w2.data_ptr(), u2.data_ptr(), Fr, H1, H1, _stream()),
                       'istvt_conv2_fwd')
        else:                               # fp32 parity mode: im2col + GEMM
            col2 = torch.empty((M2, 288), dtype=dtype, device=dev)
            _lib.check(L.istvt_im2col3x3(u1.data_ptr(), bn1.ptr(), 1, col2.data_ptr(), Fr, H1, H1, 32, ops._DT[dtype], _stream()),
                       'istvt_im2col3x3')
            u2 = ops.linear_fwd(col2, w2, blocked=False)
            del col2
        bn2 = bn('bn2', u2, M2, 64)
        # relu(bn2(u2)), block1's input (xception.py:121-125), is NOT materialised (round 6: a 0.8 GB read + write pass at C2):
        # block1's first depthwise convolution applies bn2 + ReLU as it stages its input tile -- rounded to the storage type
        # there, exactly where the separate pass rounded it --, its weight gradient takes the same pair on load, and the
        # stride-2 skip path applies them to the quarter of the pixels it keeps.
        sv.update(x=x, S=S, Fr=Fr, H1=H1, H2=H2, u1=u1, bn1=bn1, u2=u2, bn2=bn2, w1=w1, w2=w2)

        fuse_in = os.environ.get('ISTVT_STEM_MATERIALISE_A2', '0') != '1'     # (1: the separate bn_apply pass, for A/B runs)
        X, H = (u2 if fuse_in else bn_apply(u2, bn2, M2, 64, True)), H2
        blocks = []
        for name, cin_, cout, pre_relu in BLOCKS:
            i0 = 1 if pre_relu else 0
            M = Fr * H * H
            Hs = (H - 1) // 2 + 1
            Ms = Fr * Hs * Hs
            wdwA = tap_major(P['%s.rep.%d.conv1.weight' % (name, i0)])               # tap-major [9][C], cached per version
            wpwA = ops.weight_as(P['%s.rep.%d.pointwise.weight' % (name, i0)], dtype, pad=True)
            wdwB = tap_major(P['%s.rep.%d.conv1.weight' % (name, i0 + 3)])
            wpwB = ops.weight_as(P['%s.rep.%d.pointwise.weight' % (name, i0 + 3)], dtype, pad=True)
            wsk = ops.weight_as(P[name + '.skip.weight'], dtype, pad=True)
            first = name == 'block1' and fuse_in          # its input X is u2: bn2 + ReLU ride on every load of it
            d1 = dwconv(X, wdwA, Fr, H, H, cin_, in_bn=bn2 if first else None, in_relu=pre_relu or first)
            uA, bnA = pw_bn('%s.rep.%d' % (name, i0 + 1), d1, wpwA, M, cout)
            d2 = dwconv(uA, wdwB, Fr, H, H, cout, in_bn=bnA, in_relu=True)
            uB, bnB = pw_bn('%s.rep.%d' % (name, i0 + 4), d2, wpwB, M, cout)
            xs = torch.empty((Ms, cin_), dtype=dtype, device=dev)
            _lib.check(L.istvt_subsample2(X.data_ptr(), xs.data_ptr(), Fr, H, H, cin_, ops._DT[dtype], _stream()),
                       'istvt_subsample2')
            if first:
                xs = bn_apply(xs, bn2, Ms, cin_, True)
            uS, bnS = pw_bn(name + '.skipbn', xs, wsk, Ms, cout)
            out = torch.empty((Ms, cout), dtype=dtype, device=dev)
            amax = torch.empty((Ms, cout), dtype=torch.uint8, device=dev)
            esz = uB.element_size()          # reads the full-resolution map and the skip, writes the pooled map + argmax bytes
            with ops.prof('pool_add_fwd', M * cout * esz + Ms * cout * (2 * esz + 1)):
                _lib.check(L.istvt_pool_add_fwd(uB.data_ptr(), bnB.ptr(), uS.data_ptr(), bnS.ptr(), out.data_ptr(),
                                                amax.data_ptr(), Fr, H, H, cout, ops._DT[dtype], _stream()), 'istvt_pool_add_fwd')
            blocks.append(dict(name=name, i0=i0, cin=cin_, cout=cout, pre_relu=pre_relu, H=H, Hs=Hs, X=X, d1=d1, uA=uA,
                               bnA=bnA, d2=d2, uB=uB, bnB=bnB, xs=xs, uS=uS, bnS=bnS, amax=amax, wdwA=wdwA, wpwA=wpwA,
                               wdwB=wdwB, wpwB=wpwB, wsk=wsk))
            X, H = out, Hs
        sv['blocks'] = blocks
        sv['P'] = P
        sv['training'] = training
        sv['dtype'] = dtype
        ctx.sv = sv
        ctx.need_dx = x.requires_grad
        return X.view(Fr, H, H, 728)

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        sv = ctx.sv
        training = sv['training']           # eval mode: every BatchNorm is an affine map with constant statistics
        L = _lib.lib()
        P, Fr, dtype = sv['P'], sv['Fr'], sv['dtype']
        dtc = ops._DT[dtype]
        grads = {}

        # A parameter registered with GradBucket(fuse_accumulate=True) receives its gradient straight in .grad (the
        # kernels accumulate) and autograd gets None for it: no zero fill, no add pass, no copy per parameter.
        def tgt(n, shape):
            q = P[n]
            if getattr(q, '_istvt_fused_grad', False) and q.grad is not None and q.grad.is_contiguous():
                grads[n] = None
                return q.grad.view(shape)
            return None

        def bn_bwd(dz, u, st, n, M, C, stats=None):
            tg, tb = tgt(n + '.weight', (C,)), tgt(n + '.bias', (C,))
            du, dg, db = bn_backward(dz, u, st, P[n + '.weight'], M, C, stats=stats, dg=tg, db=tb, training=training)
            if tg is None:
                grads[n + '.weight'] = dg
            if tb is None:
                grads[n + '.bias'] = db
            return du

        # Weight gradients that land straight in .grad are read by nothing later in this pass: like the transformer's
        # (functional._wgrad) they go to the side stream, beside the chain of memory-bound kernels of the stem backward
        # (ISTVT_STEM_WGRAD_SIDE=0: in place, for A/B runs).
        side_on = os.environ.get('ISTVT_STEM_WGRAD_SIDE', '1') != '0'

        def lin_wgrad(n, dyv, xv):
            q = P[n]
            t = tgt(n, (q.shape[0], -1))
            if t is not None and side_on:
                from . import functional as Fn
                Fn.side_launch(dyv.device, lambda: ops.linear_wgrad(dyv, xv, out=t), keep=(dyv, xv))
                return
            r = ops.linear_wgrad(dyv, xv, out=t)
            if t is None:
                grads[n] = r

        def dw_wgrad(n, xv, dv, H_, C_, bn_, relu_):
            t = tgt(n, (C_, 9))
            if t is not None and side_on:
                from . import functional as Fn
                Fn.side_launch(dv.device, lambda: dwconv_wgrad(xv, dv, Fr, H_, H_, C_, bn_, relu_, out=t), keep=(xv, dv, bn_))
                return
            r = dwconv_wgrad(xv, dv, Fr, H_, H_, C_, bn_, relu_, out=t)
            if t is None:
                grads[n] = r

        dOut = _c(dy).reshape(-1, 728)
        for blk in reversed(sv['blocks']):
            name, i0, cin, cout, H, Hs = blk['name'], blk['i0'], blk['cin'], blk['cout'], blk['H'], blk['Hs']
            M, Ms = Fr * H * H, Fr * Hs * Hs
            dev = dOut.device
            # skip path: skipbn -> 1x1 stride-2 conv
            duS = bn_bwd(dOut, blk['uS'], blk['bnS'], name + '.skipbn', Ms, cout)
            lin_wgrad(name + '.skip.weight', duS, blk['xs'])
            dxs = ops.linear_dgrad(duS, blk['wsk'], blocked=False)
            # rep path: maxpool -> BN_B -> pointwise_B -> depthwise_B -> ReLU -> BN_A -> pointwise_A -> depthwise_A
            dzB = torch.empty((M, cout), dtype=dtype, device=dev)
            # the BatchNorm-backward sums of BN_B are taken by the pooling backward as it writes dz (train mode)
            statsB = new_stats(cout, dev) if training else None
            with ops.prof('pool_bwd', ((2 if training else 1) * M + Ms) * cout * dzB.element_size() + Ms * cout):
                _lib.check(L.istvt_pool_bwd(dOut.data_ptr(), blk['amax'].data_ptr(), dzB.data_ptr(), Fr, H, H, cout,
                                            blk['uB'].data_ptr() if training else None, blk['bnB'].ptr() if training else None,
                                            statsB[0, 0].data_ptr() if training else None,
                                            statsB[0, 1].data_ptr() if training else None, dtc, _stream()), 'istvt_pool_bwd')
            nB = '%s.rep.%d' % (name, i0 + 4)
            duB = bn_bwd(dzB, blk['uB'], blk['bnB'], nB, M, cout, stats=statsB)
            del dzB
            sB = '%s.rep.%d' % (name, i0 + 3)
            lin_wgrad(sB + '.pointwise.weight', duB, blk['d2'])
            dd2 = ops.linear_dgrad(duB, blk['wpwB'], blocked=False)
            del duB
            dw_wgrad(sB + '.conv1.weight', blk['uA'], dd2, H, cout, blk['bnA'], True)
            statsA = new_stats(cout, dev)
            dzA = dwconv(dd2, blk['wdwB'], Fr, H, H, cout, flip=True, msrc=blk['uA'], m_bn=blk['bnA'], mask_pre=True,
                         stats=statsA)
            del dd2
            nA = '%s.rep.%d' % (name, i0 + 1)
            duA = bn_bwd(dzA, blk['uA'], blk['bnA'], nA, M, cout, stats=statsA)
            del dzA
            sA = '%s.rep.%d' % (name, i0)
            lin_wgrad(sA + '.pointwise.weight', duA, blk['d1'])
            dd1 = ops.linear_dgrad(duA, blk['wpwA'], blocked=False)
            del duA
            # (block1: X is u2 -- the convolution's input was relu(bn2(u2)), taken on load as in the forward)
            fused_in = not blk['pre_relu'] and blk['X'] is sv['u2']
            dw_wgrad(sA + '.conv1.weight', blk['X'], dd1, H, cin, sv['bn2'] if fused_in else None, True if fused_in else blk['pre_relu'])
            if blk['pre_relu']:
                # d(block input) = relu'(X) * d(rep path) + scatter(d skip path)
                dOut = dwconv(dd1, blk['wdwA'], Fr, H, H, cin, flip=True, msrc=blk['X'], mask_pre=True, addsrc=dxs)
            else:
                # block1: its input is relu(bn2(u2)); fold that ReLU mask and bn2's backward statistics in
                stats2 = new_stats(cin, dev)
                dOut = dwconv(dd1, blk['wdwA'], Fr, H, H, cin, flip=True, msrc=sv['u2'], m_bn=sv['bn2'], mask_post=True,
                              addsrc=dxs, stats=stats2)
            del dd1
        # bn2 -> conv2
        H1, H2, S = sv['H1'], sv['H2'], sv['S']
        M1, M2 = Fr * H1 * H1, Fr * H2 * H2
        du2 = bn_bwd(dOut, sv['u2'], sv['bn2'], 'bn2', M2, 64, stats=stats2)
        dz1 = torch.empty((M1, 32), dtype=dtype, device=du2.device)
        dW2 = None
        if dtype == torch.bfloat16:
            du2 = du2.contiguous()
            q2 = P['conv2.weight']

            def conv2_wgrad(du=du2, u1=sv['u1'], bn1=sv['bn1']):
                dW = torch.zeros((64, 288), dtype=torch.float32, device=du.device)
                slabs = torch.empty((L.istvt_conv2_wgrad_slabs(), 64 * 288), dtype=torch.float32, device=du.device)
                _lib.check(L.istvt_conv2_wgrad(du.data_ptr(), u1.data_ptr(), bn1.ptr(), slabs.data_ptr(), dW.data_ptr(),
                                               Fr, H1, H1, _stream()), 'istvt_conv2_wgrad')
                return dW.view(64, 3, 3, 32).permute(0, 3, 1, 2)
            if side_on and getattr(q2, '_istvt_fused_grad', False) and q2.grad is not None:
                # nothing later reads it: beside the input gradient, on the side stream, added into .grad there
                from . import functional as Fn
                Fn.side_launch(du2.device, lambda: q2.grad.add_(conv2_wgrad()), keep=(du2, sv['u1'], sv['bn1']))
                grads['conv2.weight'] = None
            else:
                dW2 = conv2_wgrad().contiguous()
            _lib.check(L.istvt_conv2_dgrad(du2.data_ptr(), sv['w2'].data_ptr(), sv['u1'].data_ptr(), sv['bn1'].ptr(),
                                           dz1.data_ptr(), Fr, H1, H1, _stream()), 'istvt_conv2_dgrad')
            del du2
        else:
            col2 = torch.empty((M2, 288), dtype=dtype, device=du2.device)
            _lib.check(L.istvt_im2col3x3(sv['u1'].data_ptr(), sv['bn1'].ptr(), 1, col2.data_ptr(), Fr, H1, H1, 32, dtc, _stream()),
                       'istvt_im2col3x3')
            dW2 = ops.linear_wgrad(du2, col2)                                    # [64][(dy,dx,ci)]
            dcol2 = ops.linear_dgrad(du2, sv['w2'], blocked=False)
            del du2, col2
            _lib.check(L.istvt_col2im3x3(dcol2.data_ptr(), sv['u1'].data_ptr(), sv['bn1'].ptr(), dz1.data_ptr(), Fr, H1, H1, 32,
                                         dtc, _stream()),
                       'istvt_col2im3x3')
            del dcol2
        if 'conv2.weight' not in grads:
            grads['conv2.weight'] = dW2 if dtype == torch.bfloat16 else dW2.view(64, 3, 3, 32).permute(0, 3, 1, 2).contiguous()
        du1 = bn_bwd(dz1, sv['u1'], sv['bn1'], 'bn1', M1, 32)
        del dz1
        if dtype == torch.bfloat16 and H1 <= 128:
            du1 = du1.contiguous()
            dW1 = torch.zeros((32, 32), dtype=torch.float32, device=du1.device)      # [co][(ci,dy,dx) + 5 unused]
            slabs = torch.empty((L.istvt_conv1_wgrad_slabs(), 1024), dtype=torch.float32, device=du1.device)
            _lib.check(L.istvt_conv1_wgrad(du1.data_ptr(), sv['x'].data_ptr(), slabs.data_ptr(), dW1.data_ptr(), Fr, S, dtc,
                                           _stream()), 'istvt_conv1_wgrad')
            grads['conv1.weight'] = dW1[:, :27].reshape(32, 3, 3, 3).contiguous()
            del slabs
        else:
            col1 = torch.empty((M1, 32), dtype=dtype, device=du1.device)
            _lib.check(L.istvt_im2col_conv1(sv['x'].data_ptr(), col1.data_ptr(), Fr, S, dtc, _stream()), 'istvt_im2col_conv1')
            dW1 = ops.linear_wgrad(du1, col1)                                    # [32][(dy,dx,ci) + 5 zero columns]
            grads['conv1.weight'] = dW1[:, :27].reshape(32, 3, 3, 3).permute(0, 3, 1, 2).contiguous()
            del col1
        dx = None
        if ctx.need_dx:
            dcol1 = ops.linear_dgrad(du1, sv['w1'], blocked=False)
            dx = torch.empty_like(sv['x'])
            _lib.check(L.istvt_col2im_conv1(dcol1.data_ptr(), dx.data_ptr(), Fr, S, dtc, _stream()), 'istvt_col2im_conv1')
        ctx.sv = None
        # The two dense convolutions' weight gradients come out of a slab reduce as fresh tensors.  With the fused bucket they
        # are added into .grad HERE instead of being handed to autograd: the same add, but no AccumulateGrad node takes part
        # (inside a HIP-graph capture the engine's stream hand-over to an AccumulateGrad node created by an earlier,
        # launch-by-launch step on the default stream is illegal -- parallel.StepGraphs).
        for n in ('conv1.weight', 'conv2.weight'):
            q = P[n]
            if grads[n] is not None and getattr(q, '_istvt_fused_grad', False) and q.grad is not None:
                q.grad.add_(grads[n].view(q.shape))
                grads[n] = None
        out = [None if grads[n] is None else grads[n].view(P[n].shape) for n in param_names()]
        return (dx, None, None, None, *out)


def stem_forward(x: Tensor, xcep: torch.nn.Module, dtype: torch.dtype) -> Tensor:
    """Run the HIP stem with the parameters/buffers of an ``Xception`` module (network/xception.py)."""
    # The tensors are looked up through their owning submodules every call (a load_state_dict / .to() may have replaced
    # them): the owners are cached on the module and re-validated by identity (get_submodule per distinct owner), the
    # parameters / buffers then come straight out of the owners' dicts.  get_parameter / get_buffer by dotted name, 55 per
    # step, were ~0.4 ms of host time at the very head of a step -- where, in a loop that syncs every step, the GPU waits.
    refs = xcep.__dict__.get('_istvt_stem_refs')
    if refs is None or any(xcep.get_submodule(path) is not owner for path, owner in refs[0]):
        owners = {}
        for n in list(param_names()) + [b + '.x' for b in bn_names()]:
            path = n.rpartition('.')[0]
            if path not in owners:
                owners[path] = xcep.get_submodule(path)
        refs = (list(owners.items()), [(owners[n.rpartition('.')[0]], n.rpartition('.')[2]) for n in param_names()],
                [owners[b] for b in bn_names()])
        xcep.__dict__['_istvt_stem_refs'] = refs
    params = [owner._parameters[leaf] for owner, leaf in refs[1]]
    buffers = []
    for owner in refs[2]:
        buffers += [owner._buffers['running_mean'], owner._buffers['running_var']]
    if x.is_cuda:
        stats_arena_reset(x.device)          # one fill for every statistics accumulator of this step
        ops.refresh_stale_operands()         # one grouped cast for every bf16 weight operand the optimizer invalidated
    y = StemFn.apply(x, dtype, xcep.training, buffers, *params)
    if xcep.training:
        torch._foreach_add_([owner._buffers['num_batches_tracked'] for owner in refs[2]], 1)     # one launch, not 11
    return y
