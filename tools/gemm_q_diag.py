#!/usr/bin/env python3
"""Diagnostic variants of gemm256q_kernel at the FF1 shape (build with ISTVT_EXTRA_HIPCC_FLAGS=-DISTVT_GEMM_DIAG;
ISTVT_GEMM_QDBG=n: 1 no DMA, 2 no MFMA, 4 no LDS reads, 8 stamps).  Run on the GPU box, one process per variant."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import istvt_pkg  # noqa: E402

istvt_pkg.load()
from istvt_amd import ops  # noqa: E402

M = 56736
K, N = int(os.environ.get('GB_K', 728)), int(os.environ.get('GB_N', 2912))
lk, ln = (K + 63) // 64 * 64, (N + 63) // 64 * 64
dt = torch.bfloat16
x = (torch.randn(M, lk, device='cuda') * 0.5).to(dt)
w = (torch.randn(N, lk, device='cuda') * 0.5).to(dt)
y = torch.empty(M, ln, device='cuda', dtype=dt)
dbg = torch.zeros(8 * 16 * 3, device='cuda', dtype=torch.int64)
fn = lambda: ops.gemm_raw(x, lk, True, w, lk, True, y, ln, M, N, K, C2=dbg)
fn(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    fn()
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) / 10 * 1e-3
print('QDBG=%s K=%d N=%d: %.1f us  %.1f TF/s' % (os.environ.get('ISTVT_GEMM_QDBG', '0'), K, N, t * 1e6, 2.0 * M * N * K / t / 1e12), flush=True)
if os.environ.get('ISTVT_GEMM_QDBG') == '8':
    d = dbg.cpu().view(8, 16, 3)
    for wv in (0, 4):
        for ti in range(0, 11):
            a, b, c = [int(v) for v in d[wv, ti]]
            if a:
                nxt = int(d[wv, ti + 1, 0]) if ti + 1 < 16 and int(d[wv, ti + 1, 0]) else 0
                print('wave %d tile %2d: K loop %6d cyc, epilogue(+drain) %6d cyc, gap to next tile %6d' % (wv, ti, b - a, c - b, (nxt - c) if nxt else -1))
