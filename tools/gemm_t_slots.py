#!/usr/bin/env python3
"""Slot-level timeline of the weight-gradient (TN) GEMM, round-4 schedule (gemm256t.h, -DISTVT_T_STAMP build):
    tools/build_variant.sh tmp_ab/lib_t_stamp.so gemm.hip -DISTVT_T_STAMP
    ISTVT_LIB=tmp_ab/lib_t_stamp.so python tools/gemm_t_slots.py
One transformer layer's eight weight gradients as the grouped launch the model issues (M = 56 736 tokens)."""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import istvt_pkg  # noqa: E402

istvt_pkg.load()
from istvt_amd import _lib, ops  # noqa: E402

SEG = ['A:dma issue', 'A:frag reads', 'A:vmcnt wait', 'A:barrier(L)', 'A:mfma(+reads)', 'A:barrier(C)',
       'B:dma issue', 'B:frag reads', 'B:vmcnt wait', 'B:barrier(L)', 'B:mfma+wait', 'B:barrier(C)']
M = 56736
dt = torch.bfloat16
shapes = [(1024, 728), (512, 728), (728, 512), (1536, 728), (728, 512), (2912, 728), (728, 2912)]     # (N out, K in) of a layer


def rnd(r, c):
    v = ops.empty_rows(r, c, dt, torch.device('cuda'), True)
    v.copy_((torch.randn(r, c, device='cuda') * 0.5).to(dt))
    return v


ents = []
xs = {}
for N, K in shapes:
    if K not in xs:
        xs[K] = rnd(M, K)
    ents.append((rnd(M, N), xs[K], torch.zeros(N, K, device='cuda')))
buf = torch.zeros(256 * 8 * 16, dtype=torch.int64, device='cuda')
raw = ctypes.CDLL(_lib.LIB_PATH)
raw.istvt_diag_t_stamps.argtypes = [ctypes.c_void_p]
assert raw.istvt_diag_t_stamps(buf.data_ptr()) == 0
fn = lambda: ops.linear_wgrad_group(ents)      # noqa: E731
fn(); torch.cuda.synchronize()
t0 = time.time()
while time.time() - t0 < float(os.environ.get('GS_SECONDS', 1.5)):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    fn()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 10 * 1e3
fl = sum(2.0 * M * N * K for N, K in shapes)
d = buf.cpu().view(256, 8, 16).double()
d = d[d[:, 0, 15] > 0]
kt = d[:, :, 14:15]
per = d[:, :, :12] / kt
clk = d[:, :, 12] / d[:, :, 13] * 0.1
loop = d[:, :, 12] / kt[:, :, 0]
print('grouped weight gradients of one layer: %.1f us per launch incl. the reduce (%.0f TF/s), %d workgroups stamped' % (us, fl / us / 1e6, d.shape[0]))
print('in-kernel clock %.3f GHz; K tile (stamped build) %.0f cycles = %.3f us; K tiles per workgroup %d' %
      (float(clk.median()), float(loop.median()), float(loop.median() / clk.median() * 1e-3), int(kt.median())))
for grp, name in ((slice(0, 4), 'waves 0-3'), (slice(4, 8), 'waves 4-7')):
    med = per[:, grp, :].reshape(-1, 12).median(0).values
    print('  %s: ' % name + '  '.join('%s %.0f' % (SEG[j], float(med[j])) for j in range(12)))
    print('     load slot A %.0f | mfma slot A %.0f | load slot B %.0f | mfma slot B %.0f' %
          (float(med[0:4].sum()), float(med[4:6].sum()), float(med[6:10].sum()), float(med[10:12].sum())))
