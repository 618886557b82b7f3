#!/usr/bin/env python3
"""FeedForward forward + backward at C2's shape (M = 56 736), the GELU pair exchanging gelu'(u) (istvt_gemm flags bit 4)
against the u form; alternating in one process."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import istvt_pkg
istvt_pkg.load()
from istvt_amd import ops
M = 56736
dt = torch.bfloat16
def rnd(r, c, s=0.5):
    v = ops.empty_rows(r, c, dt, torch.device('cuda'), True); v.copy_((torch.randn(r, c, device='cuda') * s).to(dt)); return v
x, w1, w2t = rnd(M, 728), rnd(2912, 728, 728 ** -0.5), rnd(2912, 728, 2912 ** -0.5)      # w2t: W2^T [2912][728] as the dgrad operand... (dy [M,728] @ W2 [728,2912])
w2 = rnd(728, 2912, 2912 ** -0.5)
b1 = torch.randn(2912, device='cuda')
dy = rnd(M, 728)
def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
res = {False: ([], []), True: ([], [])}
for r in range(5):
    for d in (False, True):
        u, g = ops.linear_fwd(x, w1, b1, gelu=True, pad=True, gelu_d=d)
        res[d][0].append(timeit(lambda: ops.linear_fwd(x, w1, b1, gelu=True, pad=True, gelu_d=d)))
        res[d][1].append(timeit(lambda: ops.linear_dgrad(dy, w2, gelu_u=u, pad=True, gelu_d=d)))
for d in (False, True):
    f, b = sorted(res[d][0])[2], sorted(res[d][1])[2]
    print('%-28s FF1 + GELU forward %7.1f us   dX(FF2) x gelu\' backward %7.1f us   pair %7.1f us' % ('gelu\'(u) saved (flags bit 4)' if d else 'u saved', f, b, f + b))
