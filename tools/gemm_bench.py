#!/usr/bin/env python3
"""Micro-benchmark of the GEMM kernels at the model's shapes (run on the GPU box)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import istvt_pkg  # noqa: E402

istvt_pkg.load()
from istvt_amd import _lib, ops  # noqa: E402

if os.environ.get('GB_LIB'):            # A/B a variant build of the library (same box, one process per variant)
    _lib.LIB_PATH = os.path.abspath(os.environ['GB_LIB'])

M = int(os.environ.get('GB_M', 56736))
reps = int(os.environ.get('GB_REPS', 10))
which = os.environ.get('GB_WHICH', 'all')
dt = torch.bfloat16


def timeit(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


def rnd(*s):
    """random bf16 [M, D] with line-aligned rows, as the model's activations and weight operands are (ops.empty_rows);
    GB_DENSE=1: dense rows (what a caller with plain contiguous tensors gets: slower in the LDS-DMA GEMMs)"""
    t = (torch.randn(*s, device='cuda') * 0.5).to(dt)
    if os.environ.get('GB_DENSE') == '1':
        return t
    v = ops.empty_rows(s[0], s[1], dt, t.device, True)
    v.copy_(t)
    return v


shapes = [(728, 2912), (2912, 728), (728, 1536), (512, 728), (728, 1024), (728, 512)]
for K, N in shapes:
    x, w = rnd(M, K), rnd(N, K)
    b = torch.randn(N, device='cuda')
    res = rnd(M, N)
    pad = os.environ.get('GB_DENSE') != '1'
    fl = 2.0 * M * N * K
    if which in ('all', 'fwd'):
        t = timeit(lambda: ops.linear_fwd(x, w, pad=pad))
        print('fwd   plain    M=%d K=%4d N=%4d  %7.1f us  %7.1f TF/s' % (M, K, N, t * 1e6, fl / t / 1e12), flush=True)
        t = timeit(lambda: ops.linear_fwd(x, w, b, res, pad=pad))
        print('fwd   bias+res M=%d K=%4d N=%4d  %7.1f us  %7.1f TF/s' % (M, K, N, t * 1e6, fl / t / 1e12), flush=True)
    if which in ('all', 'gelu') and N == 2912:
        t = timeit(lambda: ops.linear_fwd(x, w, b, gelu=True, pad=pad))
        print('fwd   gelu     M=%d K=%4d N=%4d  %7.1f us  %7.1f TF/s' % (M, K, N, t * 1e6, fl / t / 1e12), flush=True)
    if which in ('all', 'wgrad'):
        dy = rnd(M, N)
        out = torch.zeros(N, K, device='cuda')
        t = timeit(lambda: ops.linear_wgrad(dy, x, out))
        print('wgrad          M=%d K=%4d N=%4d  %7.1f us  %7.1f TF/s' % (M, K, N, t * 1e6, fl / t / 1e12), flush=True)


if os.environ.get('GB_DEFER_AB'):
    if not hasattr(ops, 'GEMM_DEFER'):
        raise SystemExit('the deferred-epilogue prototype is not in the product: git apply tools/gemm_defer_prototype.patch, rebuild, run again')
    # same process, alternating: the plain launches as 256-row tiles, as 224-row tiles, and with the deferred epilogue
    # (gemm256q.h DEFER, 224-row tiles)
    print('--- deferred epilogue A/B (alternating, %d rounds of %d launches; lib %s) ---' % (5, reps, os.path.basename(_lib.LIB_PATH)))
    for K, N in [(728, 1536), (2912, 728), (728, 728), (1536, 728), (728, 512), (512, 728), (728, 2912)]:
        x, w = rnd(M, K), rnd(N, K)
        fl = 2.0 * M * N * K
        ts = {0: [], 1: [], 2: []}
        ys = {}
        for r in range(5):
            for d in (0, 2, 1):
                ops.GEMM_DEFER[0] = d
                ts[d].append(timeit(lambda: ops.linear_fwd(x, w, pad=True)))
                if r == 0:
                    ys[d] = ops.linear_fwd(x, w, pad=True).clone()
        ops.GEMM_DEFER[0] = 0
        same = bool(torch.equal(ys[0], ys[1])) and bool(torch.equal(ys[0], ys[2]))
        a, c, b = sorted(ts[0])[2], sorted(ts[2])[2], sorted(ts[1])[2]
        print('plain K=%4d N=%4d  256-row %7.1f us (%6.1f TF/s)  224-row %7.1f us  deferred 224-row %7.1f us (%6.1f TF/s)  %+5.1f %% vs 256, %+5.1f %% vs 224  identical %s'
              % (K, N, a * 1e6, fl / a / 1e12, c * 1e6, b * 1e6, fl / b / 1e12, (b / a - 1) * 100, (b / c - 1) * 100, same), flush=True)
