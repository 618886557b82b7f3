#!/usr/bin/env python3
"""Where does a GEMM variant differ from the fp64 reference?  Prints the error pattern by 16-row / 64-column block
(run on the GPU box; used while bringing up a new kernel)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import istvt_pkg  # noqa: E402

istvt_pkg.load()
from istvt_amd import ops  # noqa: E402

dt = torch.bfloat16


def rnd(*s, scale=1.0):
    return (torch.randn(*s, device='cuda') * scale).to(dt)


def report(name, y, ref):
    y, ref = y.double(), ref.double()
    bad = ~torch.isfinite(y) | ((y - ref).abs() > 0.05 * ref.abs().max())
    print('%-28s relerr %.3e  nan %d  bad %d / %d' % (name, float((torch.nan_to_num(y) - ref).norm() / ref.norm()),
                                                      int((~torch.isfinite(y)).sum()), int(bad.sum()), bad.numel()), flush=True)
    if bad.any():
        rows = torch.nonzero(bad.any(1)).flatten()
        cols = torch.nonzero(bad.any(0)).flatten()
        print('   bad rows: %d in [%d, %d]  first %s' % (len(rows), rows[0], rows[-1], rows[:12].tolist()))
        print('   bad cols: %d in [%d, %d]  first %s' % (len(cols), cols[0], cols[-1], cols[:12].tolist()))
        blk = bad[: (bad.shape[0] // 16) * 16].reshape(-1, 16, bad.shape[1]).any(1)
        rb = torch.nonzero(blk.any(1)).flatten() % 16
        print('   bad 16-row blocks (mod 16 within 256): %s' % sorted(set(rb.tolist())))
        idx = torch.nonzero(bad)[:6]
        refb = ref.to(dt).double()
        for i, j in idx.tolist():
            val = float(y[i, j])
            m = torch.nonzero(refb == val)
            print('   y[%d,%d] = %r  ref %r   same value in ref at %s' % (i, j, val, float(ref[i, j]), m[:4].tolist()))


for (M, K, N) in [(1000, 728, 2912), (1000, 728, 728), (512, 256, 512), (5000, 2912, 728)]:
    x, w = rnd(M, K), rnd(N, K, scale=K ** -0.5)
    b = torch.randn(N, device='cuda')
    res = rnd(M, N)
    ref = x.double() @ w.double().t()
    report('plain %s' % ((M, K, N),), ops.linear_fwd(x, w), ref)
    report('bias %s' % ((M, K, N),), ops.linear_fwd(x, w, b), ref + b.double())
    report('bias+res %s' % ((M, K, N),), ops.linear_fwd(x, w, b, res), ref + b.double() + res.double())
    u, g = ops.linear_fwd(x, w, b, gelu=True)
    report('gelu u %s' % ((M, K, N),), u, ref + b.double())
    report('gelu g %s' % ((M, K, N),), g, torch.nn.functional.gelu(ref + b.double()))
    dy = rnd(M, N)
    uu = rnd(M, K)
    ud = uu.double().requires_grad_(True)
    torch.nn.functional.gelu(ud).backward(dy.double() @ w.double())
    report('dgrad gelu %s' % ((M, K, N),), ops.linear_dgrad(dy, w, gelu_u=uu), ud.grad)
    report('dgrad %s' % ((M, K, N),), ops.linear_dgrad(dy, w), dy.double() @ w.double())
