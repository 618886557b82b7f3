#!/usr/bin/env python3
"""Slot-level timeline and in-kernel clock of the persistent NT GEMM (gemm256q.h, DBG 256) at the model's shapes.

Needs a diagnostic build of the library (the shipped one has no stamps):
    tools/build_variant.sh tmp_ab/lib_qslots.so gemm.hip -DISTVT_GEMM_DIAG -DISTVT_TUNE
    ISTVT_LIB=tmp_ab/lib_qslots.so ISTVT_GEMM_QDBG=256 python tools/gemm_slots.py > profiles/r04_gemm_slot_stamps.txt
Per shape: >= GS_SECONDS (default 2) of back-to-back launches on random operands, then the stamps of the last launch:
median over workgroups of every segment's cycles per K tile, per wave group, and the in-kernel clock
(delta s_memtime / delta s_memrealtime x 100 MHz, MI355X_MICROARCH.md 'DVFS give-back' item 6)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import istvt_pkg  # noqa: E402

istvt_pkg.load()
from istvt_amd import ops  # noqa: E402

SEG = ['A:dma issue', 'A:frag reads', 'A:vmcnt wait', 'A:barrier(L)', 'A:mfma issue', 'A:barrier(C)',
       'B:dma issue', 'B:frag reads', 'B:vmcnt wait', 'B:barrier(L)', 'B:mfma issue', 'B:barrier(C)']
M = int(os.environ.get('GS_M', 56736))
secs = float(os.environ.get('GS_SECONDS', 2.0))
shapes = [(728, 728), (728, 1536), (2912, 728), (512, 728), (728, 512)]          # (K, N)
if os.environ.get('GS_SHAPES'):
    shapes = [tuple(int(v) for v in s.split('x')) for s in os.environ['GS_SHAPES'].split(',')]
dt = torch.bfloat16


def rnd(r, c):
    v = ops.empty_rows(r, c, dt, torch.device('cuda'), True)
    v.copy_((torch.randn(r, c, device='cuda') * 0.5).to(dt))
    return v


for K, N in shapes:
    x, w = rnd(M, K), rnd(N, K)
    y = ops.empty_rows(M, N, dt, x.device, True)
    dbg = torch.zeros(256 * 8 * 24, device='cuda', dtype=torch.int64)
    lda, ldb, ldc = x.stride(0), w.stride(0), y.stride(0)

    def fn():
        ops.gemm_raw(x, lda, True, w, ldb, True, y, ldc, M, N, K, C2=dbg)
    fn(); torch.cuda.synchronize()
    t0 = time.time()
    n = 0
    while time.time() - t0 < secs:
        for _ in range(50):
            fn()
        torch.cuda.synchronize()
        n += 50
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    d = dbg.cpu().view(256, 8, 24).double()
    live = d[:, 0, 15] > 0
    d = d[live]
    G = d.shape[0]
    kt = d[:, :, 14:15]                                   # K tiles this workgroup ran
    per = d[:, :, :12] / kt                               # cycles per K tile and segment
    clk = (d[:, :, 12] / d[:, :, 13] * 0.1)               # GHz
    loop = d[:, :, 12] / kt[:, :, 0]                      # cycles per K tile (whole K loop incl. the stamps)
    print('==== M=%d K=%d N=%d: %.1f us per launch with stamps (%.0f TF/s), %d workgroups, %d launches warm' %
          (M, K, N, us, 2.0 * M * N * K / us / 1e6, G, n))
    print('in-kernel clock (median over workgroups and wavefronts): %.3f GHz   [min %.3f max %.3f]' %
          (float(clk.median()), float(clk.min()), float(clk.max())))
    print('K tile, stamped build: %.0f cycles = %.3f us' % (float(loop.median()), float(loop.median() / clk.median() * 1e-3)))
    tiles = d[:, :, 18].clamp(min=1)
    ep = (d[:, :, 16] / tiles).median()
    gap = (d[:, :, 17] / (tiles - 1).clamp(min=1)).median()
    kl = (d[:, :, 12] / tiles).median()
    print('per tile: K loop %.0f cycles, epilogue (to its last store issued) %.0f, epilogue end -> next K loop %.0f  [K loop share %.1f %%]'
          % (float(kl), float(ep), float(gap), 100.0 * float(kl / (kl + ep + gap))))
    for grp, name in ((slice(0, 4), 'waves 0-3 (leading group)'), (slice(4, 8), 'waves 4-7 (one slot behind)')):
        med = per[:, grp, :].reshape(-1, 12).median(0).values
        print('  %s: cycles per K tile, median' % name)
        for j in range(12):
            print('    %-14s %7.0f' % (SEG[j], float(med[j])))
        la, ca = float(med[0:4].sum()), float(med[4:6].sum())
        lb, cb = float(med[6:10].sum()), float(med[10:12].sum())
        print('    load slot A %5.0f | mfma slot A %5.0f | load slot B %5.0f | mfma slot B %5.0f | sum %6.0f' % (la, ca, lb, cb, la + ca + lb + cb))
    if os.environ.get('GS_PER_WAVE'):
        for wv in range(8):
            med = per[:, wv, :].median(0).values
            print('  wave %d: %s' % (wv, ' '.join('%5.0f' % float(v) for v in med)))
    sys.stdout.flush()
