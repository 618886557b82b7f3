#!/usr/bin/env python3
"""Per-tile timeline of the two GELU epilogues of the persistent NT GEMM (gemm256q.h, EPI 1 / 2) at the feed-forward's
shape (M = 56 736, K = 728, N = 2912; module.py:27-28), from in-kernel s_memtime stamps.  VERDICT r4 item 6.

Needs a diagnostic build (the shipped library has no stamps), one process per variant:
    tools/build_variant.sh tmp_ab/lib_gelu.so gemm.hip -DISTVT_GEMM_DIAG -DISTVT_TUNE
    for q in 1024 1152 3072 3200; do ISTVT_LIB=tmp_ab/lib_gelu.so ISTVT_GEMM_QDBG=$q python tools/gemm_gelu_stamps.py; done
QDBG 1024 = stamps only (three per tile), 1152 = + output stores out of range (no store traffic), 3072 = + no GELU
arithmetic (the stores and the side load stay), 3200 = neither.  ISTVT_GEMM_QDBG unset = the shipped kernel, launch time only."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import istvt_pkg  # noqa: E402

istvt_pkg.load()
from istvt_amd import ops, stem  # noqa: E402

M, K, N = int(os.environ.get('GS_M', 56736)), 728, 2912
secs = float(os.environ.get('GS_SECONDS', 1.5))
dt = torch.bfloat16
qdbg = int(os.environ.get('ISTVT_GEMM_QDBG', '0'))
dbg = torch.zeros(256 * 8 * 24, device='cuda', dtype=torch.int64)
os.environ['ISTVT_GEMM_DBGPTR'] = hex(dbg.data_ptr())


def rnd(r, c, s=0.5):
    v = ops.empty_rows(r, c, dt, torch.device('cuda'), True)
    v.copy_((torch.randn(r, c, device='cuda') * s).to(dt))
    return v


x, w = rnd(M, K), rnd(N, K, 0.04)
b = torch.randn(N, device='cuda')
u, g = ops.empty_rows(M, N, dt, x.device, True), ops.empty_rows(M, N, dt, x.device, True)
dy, wt = rnd(M, K), rnd(N, K, 0.04)           # dgrad of FF2 as NT on W2^T [2912, 728]
du = ops.empty_rows(M, N, dt, x.device, True)
uu = rnd(M, N, 1.0)
acc = stem.new_stats(N, x.device)


def fwd():
    ops.gemm_raw(x, x.stride(0), True, w, w.stride(0), True, u, u.stride(0), M, N, K, bias=b, C2=g, epi=1)


def bwd():
    ops.gemm_raw(dy, dy.stride(0), True, wt, wt.stride(0), True, du, du.stride(0), M, N, K, C2=uu, epi=2, stats=acc, csum=True)


for name, fn in (('GELU forward  (EPI 1: stores u and gelu(u))', fwd), ('GELU backward (EPI 2: loads u, stores dy * gelu\'(u), column sums)', bwd)):
    fn(); torch.cuda.synchronize()
    t0 = time.time()
    while time.time() - t0 < secs:
        for _ in range(50):
            fn()
        torch.cuda.synchronize()
    dbg.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    line = 'QDBG %-5d %s: %.1f us per launch (%.0f TF/s)' % (qdbg, name, us, 2.0 * M * N * K / us / 1e6)
    if qdbg:
        d = dbg.cpu().view(256, 8, 24).double()
        d = d[d[:, 0, 15] > 0]
        tiles = d[:, :, 18].clamp(min=1)
        clk = float((d[:, :, 12] / d[:, :, 13] * 0.1).median())
        kl = float((d[:, :, 12] / tiles).median())
        ep = float((d[:, :, 16] / tiles).median())
        gap = float((d[:, :, 17] / (tiles - 1).clamp(min=1)).median())
        line += '; in-kernel clock %.2f GHz; per tile: K loop %.0f cycles, epilogue (to its last store issued) %.0f, gap to the next K loop %.0f' % (clk, kl, ep, gap)
    print(line, flush=True)
