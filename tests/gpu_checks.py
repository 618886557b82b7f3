"""GPU parity checks of the individual HIP kernels (through the C ABI wrappers in
istvt_amd.ops) against fp64 PyTorch restatements of the same op.  Each check returns
(error, tolerance); tests/test_kernels_gpu.py asserts them, tools/devcheck.py prints them all.

Tolerances: float32 storage -> 2e-5 relative (fp32 accumulate, different summation order);
bfloat16 storage -> 2e-2 relative to the fp64 result computed from the SAME bf16-rounded inputs
(the error left is output rounding 2^-9 plus bf16 rounding of intermediates such as P in P@V).
Integer-valued GEMM checks are exact (tolerance 0) in both dtypes: they pin every operand
layout / fragment mapping.
"""
import math

import torch

import istvt_pkg

istvt_amd = istvt_pkg.load()
from istvt_amd import ops  # noqa: E402

DEV = 'cuda'
TOL = {torch.float32: 2e-5, torch.bfloat16: 2e-2}


def relerr(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def rnd(shape, dtype, seed, scale=1.0):
    g = torch.Generator(device='cpu').manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(DEV).to(dtype)


def ints(shape, dtype, seed, lo=-2, hi=3):
    g = torch.Generator(device='cpu').manual_seed(seed)
    return torch.randint(lo, hi, shape, generator=g).to(DEV).to(dtype)


# ------------------------------------------------------------------------------------------ GEMM
def gemm_exact(dtype, mode, M=200, N=136, K=104):
    """integer data -> exact; mode in fwd/dgrad/wgrad; covers M/N/K tails."""
    if mode == 'fwd':
        x, w = ints((M, K), dtype, 1), ints((N, K), dtype, 2)
        y = ops.linear_fwd(x, w)
        ref = x.double() @ w.double().t()
    elif mode == 'dgrad':
        dy, w = ints((M, N), dtype, 3), ints((N, K), dtype, 4)
        y = ops.linear_dgrad(dy, w)
        ref = dy.double() @ w.double()
    else:
        dy, x = ints((M, N), dtype, 5), ints((M, K), dtype, 6)
        y = ops.linear_wgrad(dy, x)
        ref = dy.double().t() @ x.double()
    if mode != 'wgrad':
        ref = ref.to(dtype)            # the exact sum, rounded once to the storage type
    return float((y.double() - ref.double()).abs().max()), 0.0


def gemm_real(dtype, mode):
    M, N, K = 1000, 728, 2912
    if mode == 'fwd_bias_res':
        x, w, b, r = rnd((M, K), dtype, 1), rnd((N, K), dtype, 2, K ** -0.5), rnd((N,), torch.float32, 3), rnd((M, N), dtype, 4)
        y = ops.linear_fwd(x, w, b, r)
        ref = x.double() @ w.double().t() + b.double() + r.double()
    elif mode == 'fwd_gelu':
        x, w, b = rnd((M, 728), dtype, 1), rnd((2912, 728), dtype, 2, 728 ** -0.5), rnd((2912,), torch.float32, 3)
        u, g = ops.linear_fwd(x, w, b, gelu=True)
        ru = x.double() @ w.double().t() + b.double()
        rg = torch.nn.functional.gelu(ru)
        return max(relerr(u, ru), relerr(g, rg)), TOL[dtype]
    elif mode == 'dgrad_gelu':
        dy, w, u = rnd((M, 728), dtype, 1), rnd((728, 2912), dtype, 2, 728 ** -0.5), rnd((M, 2912), dtype, 3)
        y = ops.linear_dgrad(dy, w, gelu_u=u)
        ud = u.double().requires_grad_(True)
        torch.nn.functional.gelu(ud).backward(dy.double() @ w.double())
        ref = ud.grad
    elif mode == 'wgrad':
        dy, x = rnd((5000, N), dtype, 5), rnd((5000, 512), dtype, 6)
        y = ops.linear_wgrad(dy, x)
        ref = dy.double().t() @ x.double()
    elif mode == 'head':            # N = 1 (Linear(dim, 1)) and its gradients
        x, w, b = rnd((32, 728), dtype, 1), rnd((1, 728), dtype, 2, 0.05), rnd((1,), torch.float32, 3)
        y = ops.linear_fwd(x, w, b)
        dy = rnd((32, 1), dtype, 4)
        dx = ops.linear_dgrad(dy, w)
        dw = ops.linear_wgrad(dy, x)
        db = ops.colsum(dy)
        e = max(relerr(y, x.double() @ w.double().t() + b.double()), relerr(dx, dy.double() @ w.double()),
                relerr(dw, dy.double().t() @ x.double()), relerr(db, dy.double().sum(0)))
        return e, TOL[dtype]
    return relerr(y, ref), TOL[dtype]


# ------------------------------------------------------------------------------------------ LayerNorm
def layernorm(dtype, D=728, M=1003):
    x, g, b = rnd((M, D), dtype, 1, 2.0), rnd((D,), torch.float32, 2, 0.2) + 1, rnd((D,), torch.float32, 3, 0.1)
    y, mean, rstd = ops.layernorm_fwd(x, g, b, 1e-5)
    xd = x.double().requires_grad_(True)
    gd, bd = g.double().requires_grad_(True), b.double().requires_grad_(True)
    ref = torch.nn.functional.layer_norm(xd, (D,), gd, bd, 1e-5)
    dy, dres = rnd((M, D), dtype, 4), rnd((M, D), dtype, 5)
    ref.backward(dy.double())
    dg, db = torch.zeros_like(g), torch.zeros_like(b)
    dx = ops.layernorm_bwd(dy, x, mean, rstd, g, dg, db, dres=dres)
    e = max(relerr(y, ref), relerr(dx, xd.grad + dres.double()), relerr(dg, gd.grad), relerr(db, bd.grad))
    return e, TOL[dtype]


def _diff_ref(y, B, F, P):
    yr = y.view(B, F, P, -1)
    return torch.cat((yr[:, :2], yr[:, 2:] - yr[:, 1:-1]), dim=1).reshape(B * F * P, -1)


def layernorm_diff(dtype, B=2, F=9, P=37, D=728):
    M = B * F * P
    x, g, b = rnd((M, D), dtype, 1, 2.0), rnd((D,), torch.float32, 2, 0.2) + 1, rnd((D,), torch.float32, 3, 0.1)
    y, diff, mean, rstd = ops.layernorm_fwd_diff(x, g, b, 1e-5, B, F, P)
    xd = x.double().requires_grad_(True)
    gd, bd = g.double().requires_grad_(True), b.double().requires_grad_(True)
    ry = torch.nn.functional.layer_norm(xd, (D,), gd, bd, 1e-5)
    rdiff = _diff_ref(ry, B, F, P)
    dy, dd = rnd((M, D), dtype, 4), rnd((M, D), dtype, 5)
    (ry * dy.double()).sum().add((rdiff * dd.double()).sum()).backward()
    dg, db = torch.zeros_like(g), torch.zeros_like(b)
    dx = ops.layernorm_bwd(dy, x, mean, rstd, g, dg, db, dy2=dd, F=F, P=P)
    e = max(relerr(y, ry), relerr(diff, rdiff), relerr(dx, xd.grad), relerr(dg, gd.grad), relerr(db, bd.grad))
    return e, TOL[dtype] * (3 if dtype == torch.bfloat16 else 1)


def frame_diff(dtype, B=2, F=7, P=13, D=64):
    x = rnd((B * F * P, D), dtype, 1)
    xd = x.double().requires_grad_(True)
    ref = _diff_ref(xd, B, F, P)
    g = rnd((B * F * P, D), dtype, 2)
    ref.backward(g.double())
    e = max(relerr(ops.frame_diff(x, B, F, P), ref), relerr(ops.frame_diff(g, B, F, P, adjoint=True), xd.grad))
    return e, TOL[dtype]


# ------------------------------------------------------------------------------------------ attention
def _attn_ref(q, k, v):
    s = (q @ k.transpose(-1, -2)) * q.shape[-1] ** -0.5
    return s.softmax(-1) @ v


def attn_spatial(dtype, BF=3, P=197, heads=8, dh=64):
    inner = heads * dh
    qkv = rnd((BF * P, 3 * inner), dtype, 1)
    out, lse = ops.attn_spatial_fwd(qkv, BF, P, heads, dh)
    qd = qkv.double().requires_grad_(True)
    q, k, v = (t.view(BF, P, heads, dh).transpose(1, 2) for t in qd.chunk(3, dim=-1))
    ref = _attn_ref(q, k, v).transpose(1, 2).reshape(BF * P, inner)
    dout = rnd((BF * P, inner), dtype, 2)
    ref.backward(dout.double())
    dqkv = ops.attn_spatial_bwd(qkv, out, dout, lse, BF, P, heads, dh)
    e_f = relerr(out, ref)
    e_b = max(relerr(a, b) for a, b in zip(dqkv.chunk(3, dim=-1), qd.grad.chunk(3, dim=-1)))
    return max(e_f, e_b), TOL[dtype]


def attn_temporal(dtype, B=2, F=9, P=37, heads=8, dh=64):
    inner = heads * dh
    M = B * F * P
    qk, v = rnd((M, 2 * inner), dtype, 1), rnd((M, inner), dtype, 2)
    out = ops.attn_temporal_fwd(qk, v, B, F, P, heads, dh)
    qkd, vd = qk.double().requires_grad_(True), v.double().requires_grad_(True)

    def split(t):                                    # (b f p) (h d) -> b h p f d
        return t.view(B, F, P, heads, dh).permute(0, 3, 2, 1, 4)
    q, k = (split(t) for t in qkd.chunk(2, dim=-1))
    ref = _attn_ref(q, k, split(vd)).permute(0, 3, 2, 1, 4).reshape(M, inner)
    dout = rnd((M, inner), dtype, 3)
    ref.backward(dout.double())
    dqk, dv = ops.attn_temporal_bwd(qk, v, dout, B, F, P, heads, dh)
    e = max(relerr(out, ref), relerr(dqk, qkd.grad), relerr(dv, vd.grad))
    return e, TOL[dtype]


# ------------------------------------------------------------------------------------------ misc
def tokens(dtype, B=3, T=4, hw=36, D=728):
    feats = rnd((B, T, hw, D), dtype, 1)
    space, temporal = rnd((1, 1, D), torch.float32, 2), rnd((1, 1, D), torch.float32, 3)
    pos = rnd((1, T, hw + 3, D), torch.float32, 4)          # declared grid larger than the input's
    x = ops.tokens_fwd(feats, space, temporal, pos)
    fd = feats.double().requires_grad_(True)
    sd, td, pd = (t.double().requires_grad_(True) for t in (space, temporal, pos))
    r = torch.cat((sd.view(1, 1, 1, D).expand(B, T, 1, D), fd), dim=2) + pd[:, :, :hw + 1]
    r = torch.cat((td.view(1, 1, 1, D).expand(B, 1, hw + 1, D), r), dim=1).reshape(B, -1, D)
    dx = rnd(tuple(r.shape), dtype, 5)
    r.backward(dx.double())
    ds, dt, dp = torch.zeros_like(space), torch.zeros_like(temporal), torch.zeros_like(pos)
    dfe = ops.tokens_bwd(dx, B, T, hw, D, ds, dt, dp, True)
    e = max(relerr(x, r), relerr(dfe, fd.grad), relerr(ds, sd.grad), relerr(dt, td.grad), relerr(dp, pd.grad))
    return e, TOL[dtype]


def colsum_cast(dtype):
    x = rnd((3001, 2912), dtype, 1)
    e = relerr(ops.colsum(x), x.double().sum(0))
    w = rnd((700, 728), torch.float32, 2)
    e = max(e, relerr(ops.cast(w, dtype), w.to(dtype)))
    return e, TOL[dtype]


def all_checks():
    """-> list of (name, callable)"""
    out = []
    for dt, tag in ((torch.float32, 'f32'), (torch.bfloat16, 'bf16')):
        for mode in ('fwd', 'dgrad', 'wgrad'):
            out.append(('gemm_exact_%s_%s' % (mode, tag), lambda dt=dt, mode=mode: gemm_exact(dt, mode)))
            out.append(('gemm_exact_big_%s_%s' % (mode, tag), lambda dt=dt, mode=mode: gemm_exact(dt, mode, 515, 728, 1544)))
        for mode in ('fwd_bias_res', 'fwd_gelu', 'dgrad_gelu', 'wgrad', 'head'):
            out.append(('gemm_%s_%s' % (mode, tag), lambda dt=dt, mode=mode: gemm_real(dt, mode)))
        out.append(('layernorm_%s' % tag, lambda dt=dt: layernorm(dt)))
        out.append(('layernorm_d64_%s' % tag, lambda dt=dt: layernorm(dt, 64, 77)))
        out.append(('layernorm_diff_%s' % tag, lambda dt=dt: layernorm_diff(dt)))
        out.append(('frame_diff_%s' % tag, lambda dt=dt: frame_diff(dt)))
        for P, heads, dh in ((197, 8, 64), (37, 8, 64), (362, 2, 32), (362, 8, 64), (128, 2, 64)):
            out.append(('attn_spatial_P%d_h%d_d%d_%s' % (P, heads, dh, tag),
                        lambda dt=dt, P=P, heads=heads, dh=dh: attn_spatial(dt, 3, P, heads, dh)))
        for F, heads, dh in ((9, 8, 64), (5, 2, 32), (17, 8, 64), (7, 8, 64), (17, 2, 32)):
            out.append(('attn_temporal_F%d_h%d_d%d_%s' % (F, heads, dh, tag),
                        lambda dt=dt, F=F, heads=heads, dh=dh: attn_temporal(dt, 2, F, 37, heads, dh)))
        out.append(('tokens_%s' % tag, lambda dt=dt: tokens(dt)))
        out.append(('colsum_cast_%s' % tag, lambda dt=dt: colsum_cast(dt)))
    return out
