#!/usr/bin/env python3
"""FF1 (+bias, GELU: stores u and gelu(u)) followed by FF2 (+bias, residual) back to back, as FeedForward runs them
(module.py:27-30), at C2's shape: time per pair and per GEMM.  For A/Bs of the epilogue's store policies (ISTVT_LIB)."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import istvt_pkg  # noqa: E402
istvt_pkg.load()
from istvt_amd import ops  # noqa: E402

M, D, H = 56736, 728, 2912
dt = torch.bfloat16


def rnd(r, c, s=0.5):
    v = ops.empty_rows(r, c, dt, torch.device('cuda'), True)
    v.copy_((torch.randn(r, c, device='cuda') * s).to(dt))
    return v


x, w1, w2 = rnd(M, D), rnd(H, D, 0.04), rnd(D, H, 0.02)
b1, b2 = torch.randn(H, device='cuda'), torch.randn(D, device='cuda')
for _ in range(30):
    u, g = ops.linear_fwd(x, w1, b1, gelu=True, pad=True)
    y = ops.linear_fwd(g, w2, b2, x, pad=True)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
t1 = t2 = 0.0
R = 40
for _ in range(R):
    ev[0].record()
    u, g = ops.linear_fwd(x, w1, b1, gelu=True, pad=True)
    ev[1].record()
    y = ops.linear_fwd(g, w2, b2, x, pad=True)
    ev[2].record()
    torch.cuda.synchronize()
    t1 += ev[0].elapsed_time(ev[1]); t2 += ev[1].elapsed_time(ev[2])
print('%s: FF1+GELU %.1f us, FF2+residual %.1f us, pair %.1f us' % (os.environ.get('ISTVT_LIB', 'shipped'), t1 / R * 1e3, t2 / R * 1e3, (t1 + t2) / R * 1e3), flush=True)
