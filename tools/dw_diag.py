#!/usr/bin/env python3
"""In-kernel stamps of the depthwise 3x3 forward (build with ISTVT_EXTRA_HIPCC_FLAGS=-DISTVT_DW_DIAG): where a
workgroup's time goes -- tile fetch + commit, barrier, convolution + store issue, store drain.  Run on the GPU box."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

dbg = torch.zeros(32 * 4 * 5, dtype=torch.int64, device='cuda')
os.environ['ISTVT_DW_DBGPTR'] = str(dbg.data_ptr())
import istvt_pkg  # noqa: E402

istvt_pkg.load()
from istvt_amd import stem as S  # noqa: E402

Fr, H, W, C = 256, 109, 109, int(os.environ.get('DW_C', 64))
x = torch.randn(Fr * H * W, C, device='cuda').to(torch.bfloat16)
w9 = torch.randn(9, C, device='cuda')
for _ in range(3):
    y = S.dwconv(x, w9, Fr, H, W, C)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    y = S.dwconv(x, w9, Fr, H, W, C)
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) / 5 * 1e-3
print('dwconv fwd C=%d: %.1f us  %.2f TB/s' % (C, t * 1e6, 2 * x.numel() * 2 / t / 1e12))
d = dbg.cpu().view(32, 4, 5)
import statistics
seg = [[], [], [], []]
for b in range(32):
    for wv in range(4):
        s = [int(v) for v in d[b, wv]]
        if s[0]:
            for i in range(4):
                seg[i].append(s[i + 1] - s[i])
for name, v in zip(('fetch+commit', 'barrier', 'convolve+store issue', 'store drain'), seg):
    if v:
        print('%-22s median %6d cycles  (min %d, max %d)' % (name, statistics.median(v), min(v), max(v)))
