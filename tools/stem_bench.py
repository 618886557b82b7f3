#!/usr/bin/env python3
"""Micro-benchmark of the HBM-bound stem / LayerNorm kernels at C2 sizes: achieved GB/s."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import istvt_pkg
istvt_pkg.load()
from istvt_amd import ops, stem as S, _lib

dt = torch.bfloat16
reps = 10

def timeit(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3

def rnd(*s): return torch.randn(*s, device='cuda').to(dt)

Fr = 256
for H, C in ((109, 64), (109, 128), (55, 256), (28, 728)):
    M = Fr * H * H
    x = rnd(M, C); w = torch.randn(9, C, device='cuda')
    bn = S.BNState(C, 'cuda'); bn.pack.normal_()
    t = timeit(lambda: S.dwconv(x, w, Fr, H, H, C))
    print('dwconv fwd plain   H=%3d C=%3d  %7.1f us  %6.0f GB/s' % (H, C, t * 1e6, 2 * M * C * 2 / t / 1e9), flush=True)
    t = timeit(lambda: S.dwconv(x, w, Fr, H, H, C, in_bn=bn, in_relu=True))
    print('dwconv fwd bn+relu H=%3d C=%3d  %7.1f us  %6.0f GB/s' % (H, C, t * 1e6, 2 * M * C * 2 / t / 1e9), flush=True)
    st = S.new_stats(C, 'cuda')
    t = timeit(lambda: S.dwconv(x, w, Fr, H, H, C, flip=True, msrc=x, m_bn=bn, mask_pre=True, stats=st))
    print('dwconv bwd mask+st H=%3d C=%3d  %7.1f us  %6.0f GB/s' % (H, C, t * 1e6, 3 * M * C * 2 / t / 1e9), flush=True)
    t = timeit(lambda: S.dwconv_wgrad(x, x, Fr, H, H, C, bn, True))
    print('dwconv wgrad       H=%3d C=%3d  %7.1f us  %6.0f GB/s' % (H, C, t * 1e6, 2 * M * C * 2 / t / 1e9), flush=True)
    g = torch.ones(C, device='cuda'); rm = torch.zeros(C, device='cuda'); rv = torch.ones(C, device='cuda')
    t = timeit(lambda: S.bn_forward_stats(x, M, C, g, g, rm, rv, True))
    print('bn stats+finalize  H=%3d C=%3d  %7.1f us  %6.0f GB/s' % (H, C, t * 1e6, M * C * 2 / t / 1e9), flush=True)
    t = timeit(lambda: S.bn_backward(x, x, bn, g, M, C))
    print('bn bwd stats+apply H=%3d C=%3d  %7.1f us  %6.0f GB/s' % (H, C, t * 1e6, 5 * M * C * 2 / t / 1e9), flush=True)
# MaxPool2d(3,2,1) + both BatchNorm applies + skip add, and its backward with the BatchNorm-backward sums (the three
# stride-2 blocks of the entry flow: 109 -> 55 at 128 channels, 55 -> 28 at 256, 28 -> 14 at 728)
L = _lib.lib()
for H, C in ((109, 128), (55, 256), (28, 728)):
    Ho = (H - 1) // 2 + 1
    M, Ms = Fr * H * H, Fr * Ho * Ho
    x = rnd(M, C); sk = rnd(Ms, C); out = torch.empty(Ms, C, device='cuda', dtype=dt)
    amax = torch.empty(Ms, C, device='cuda', dtype=torch.uint8)
    bnx = S.BNState(C, 'cuda'); bnx.pack.normal_(); bns = S.BNState(C, 'cuda'); bns.pack.normal_()
    st0 = torch.cuda.current_stream().cuda_stream
    def pf():
        _lib.check(L.istvt_pool_add_fwd(x.data_ptr(), bnx.ptr(), sk.data_ptr(), bns.ptr(), out.data_ptr(), amax.data_ptr(),
                                        Fr, H, H, C, ops._DT[dt], st0), 'pool_add_fwd')
    t = timeit(pf)
    print('pool_add_fwd       H=%3d C=%3d  %7.1f us  %6.0f GB/s' % (H, C, t * 1e6, (M * C * 2 + Ms * C * 5) / t / 1e9), flush=True)
    dz = torch.empty(M, C, device='cuda', dtype=dt); st = S.new_stats(C, 'cuda')
    def pb():
        _lib.check(L.istvt_pool_bwd(out.data_ptr(), amax.data_ptr(), dz.data_ptr(), Fr, H, H, C, x.data_ptr(), bnx.ptr(),
                                    st[0, 0].data_ptr(), st[0, 1].data_ptr(), ops._DT[dt], st0), 'pool_bwd')
    t = timeit(pb)
    print('pool_bwd + sums    H=%3d C=%3d  %7.1f us  %6.0f GB/s' % (H, C, t * 1e6, (2 * M * C * 2 + Ms * C * 3) / t / 1e9), flush=True)
M, D = 56736, 728
x = rnd(M, D); g = torch.ones(D, device='cuda'); b = torch.zeros(D, device='cuda')
t = timeit(lambda: ops.layernorm_fwd(x, g, b, 1e-5))
print('layernorm fwd   %7.1f us  %6.0f GB/s' % (t * 1e6, 2 * M * D * 2 / t / 1e9))
y, mean, rstd = ops.layernorm_fwd(x, g, b, 1e-5)
dg, db = torch.zeros_like(g), torch.zeros_like(b)
t = timeit(lambda: ops.layernorm_bwd(x, x, mean, rstd, g, dg, db))
print('layernorm bwd   %7.1f us  %6.0f GB/s' % (t * 1e6, 3 * M * D * 2 / t / 1e9))
t = timeit(lambda: ops.layernorm_bwd(x, x, mean, rstd, g, dg, db, dres=x))
print('layernorm bwd+r %7.1f us  %6.0f GB/s' % (t * 1e6, 4 * M * D * 2 / t / 1e9))
t = timeit(lambda: ops.colsum(x))
print('colsum          %7.1f us  %6.0f GB/s' % (t * 1e6, M * D * 2 / t / 1e9))
