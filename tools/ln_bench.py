#!/usr/bin/env python3
"""Micro-benchmark of the LayerNorm / column-sum kernels at the model's shapes (run on the GPU box)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import istvt_pkg  # noqa: E402

istvt_pkg.load()
from istvt_amd import ops  # noqa: E402

B, F, P, D = 32, 9, 197, int(os.environ.get("LN_D", 728))      # LN_D: another row length (512 = one full instruction per row)
M = B * F * P
dt = torch.bfloat16
reps = 20


def timeit(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


def rnd(*s):
    return torch.randn(*s, device='cuda').to(dt)


x, dy, dy2, dres = rnd(M, D), rnd(M, D), rnd(M, D), rnd(M, D)
g, b = torch.randn(D, device='cuda'), torch.randn(D, device='cuda')
dg, db, dc = torch.zeros(D, device='cuda'), torch.zeros(D, device='cuda'), torch.zeros(D, device='cuda')
y, mean, rstd = ops.layernorm_fwd(x, g, b, 1e-5)
row = M * D * 2
for name, fn, nbytes in [
    ('ln_fwd', lambda: ops.layernorm_fwd(x, g, b, 1e-5), 2 * row),
    ('ln_bwd', lambda: ops.layernorm_bwd(dy, x, mean, rstd, g, dg, db), 3 * row),
    ('ln_bwd + dres', lambda: ops.layernorm_bwd(dy, x, mean, rstd, g, dg, db, dres=dres), 4 * row),
    ('ln_bwd + dres + dcol', lambda: ops.layernorm_bwd(dy, x, mean, rstd, g, dg, db, dres=dres, dcol=dc), 4 * row),
    ('colsum', lambda: ops.colsum(dy, db), row),
]:
    t = timeit(fn)
    print('%-26s %8.1f us   %6.2f TB/s (algorithmic %d MB)' % (name, t * 1e6, nbytes / t / 1e12, nbytes / 1e6), flush=True)
