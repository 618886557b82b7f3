#!/usr/bin/env python3
"""GEMM micro-benchmark with operand row strides padded to a multiple of 64 elements (128 B): what the LDS-DMA
staging gains from line-aligned rows (run on the GPU box)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import istvt_pkg  # noqa: E402

istvt_pkg.load()
from istvt_amd import ops  # noqa: E402

M = int(os.environ.get('GB_M', 56736))
reps = 10
dt = torch.bfloat16


def timeit(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


def pad64(n):
    return (n + 63) // 64 * 64


def rnd(r, c, ld):
    return (torch.randn(r, ld, device='cuda') * 0.5).to(dt)


for K, N in [(728, 2912), (2912, 728), (728, 1536), (512, 728), (728, 1024), (728, 512)]:
    fl = 2.0 * M * N * K
    for padded in (False, True):
        lk, ln = (pad64(K), pad64(N)) if padded else (K, N)
        x, w, y, res = rnd(M, K, lk), rnd(N, K, lk), rnd(M, N, ln), rnd(M, N, ln)
        b = torch.randn(N, device='cuda')
        t0 = timeit(lambda: ops.gemm_raw(x, lk, True, w, lk, True, y, ln, M, N, K))
        t1 = timeit(lambda: ops.gemm_raw(x, lk, True, w, lk, True, y, ln, M, N, K, bias=b, residual=res, ldr=ln))
        print('K=%4d N=%4d %s  plain %7.1f us %6.1f TF/s   bias+res %7.1f us %6.1f TF/s'
              % (K, N, 'padded  ' if padded else 'unpadded', t0 * 1e6, fl / t0 / 1e12, t1 * 1e6, fl / t1 / 1e12), flush=True)
