#!/usr/bin/env python3
"""What a user of the reference gets on this MI355X from stock PyTorch-ROCm ops (MIOpen / hipBLASLt / the library
attention kernels) against this package's kernels, op by op at C2's shapes (bf16, forward + backward where both exist).
Measurement only: nothing in the product path calls a torch op for these."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402
import istvt_pkg  # noqa: E402
istvt_pkg.load()
from istvt_amd import ops, stem  # noqa: E402

dt = torch.bfloat16
dev = torch.device('cuda', 0)


def timeit(fn, reps=20):
    for _ in range(4):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3          # us


def row(name, t_lib, t_own):
    print('%-58s stock torch %8.1f us | this package %8.1f us | x%.2f' % (name, t_lib, t_own, t_lib / t_own), flush=True)


# ---- LayerNorm over 728 (module.py:18), M = 56 736 rows
M, D = 56736, 728
x = (torch.randn(M, D, device=dev)).to(dt)
g, b = torch.ones(D, device=dev), torch.zeros(D, device=dev)
xp = ops.empty_rows(M, D, dt, dev, True); xp.copy_(x)
row('LayerNorm forward (56736 x 728)', timeit(lambda: F.layer_norm(x, (D,), g.to(dt), b.to(dt))),
    timeit(lambda: ops.layernorm_fwd(xp, g, b, 1e-5, pad=True)))
xr = x.clone().requires_grad_(True)
gw, bw = g.to(dt).requires_grad_(True), b.to(dt).requires_grad_(True)
y = F.layer_norm(xr, (D,), gw, bw)
dy = torch.randn_like(y)
y2, mean, rstd = ops.layernorm_fwd(xp, g, b, 1e-5, pad=True)
dyp = ops.empty_rows(M, D, dt, dev, True); dyp.copy_(dy)
dg, db = torch.zeros(D, device=dev), torch.zeros(D, device=dev)
row('LayerNorm backward', timeit(lambda: torch.autograd.grad(y, (xr, gw, bw), dy, retain_graph=True)),
    timeit(lambda: ops.layernorm_bwd(dyp, xp, mean, rstd, g, dg, db, pad=True)))

# ---- spatial attention (module.py:84-91): 288 frames x 8 heads, P = 197, d = 64
BF, P, H, dh = 288, 197, 8, 64
qkv = (torch.randn(BF * P, 3 * H * dh, device=dev)).to(dt)
q, k, v = (t.view(BF, P, H, dh).transpose(1, 2).contiguous() for t in qkv.chunk(3, dim=-1))
row('spatial attention forward (2304 x [197 x 64])', timeit(lambda: F.scaled_dot_product_attention(q, k, v)),
    timeit(lambda: ops.attn_spatial_fwd(qkv, BF, P, H, dh)))
qr, kr, vr = q.clone().requires_grad_(True), k.clone().requires_grad_(True), v.clone().requires_grad_(True)
o = F.scaled_dot_product_attention(qr, kr, vr)
do = torch.randn_like(o)
out, lse = ops.attn_spatial_fwd(qkv, BF, P, H, dh)
dop = torch.randn(BF * P, H * dh, device=dev).to(dt)
row('spatial attention backward', timeit(lambda: torch.autograd.grad(o, (qr, kr, vr), do, retain_graph=True)),
    timeit(lambda: ops.attn_spatial_bwd(qkv, out, dop, lse, BF, P, H, dh)))

# ---- depthwise 3x3 (xception.py:43): 256 frames, 109 x 109 x 128 and 55 x 55 x 256
for Hh, C in ((109, 128), (55, 256), (28, 728)):
    Fr = 256
    xn = torch.randn(Fr, C, Hh, Hh, device=dev).to(dt).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(C, 1, 3, 3, device=dev) * 0.3).to(dt)
    xh = xn.permute(0, 2, 3, 1).contiguous().view(Fr * Hh * Hh, C)
    w9 = stem.tap_major(torch.nn.Parameter(w.float()))
    t_own = timeit(lambda: stem.dwconv(xh, w9, Fr, Hh, Hh, C))
    row('depthwise 3x3 forward (256 x %d^2 x %d, channels-last)' % (Hh, C), timeit(lambda: F.conv2d(xn, w, None, 1, 1, 1, C)), t_own)
    xc = xn.contiguous()                      # NCHW, the layout the reference's tensors have (xception.py:193-206)
    row('depthwise 3x3 forward (the same, NCHW as in the reference)', timeit(lambda: F.conv2d(xc, w, None, 1, 1, 1, C)), t_own)

# ---- train-mode BatchNorm + ReLU and MaxPool(3, 2, 1) on 256 x 109^2 x 128 (channels-last)
Fr, Hh, C = 256, 109, 128
xn = torch.randn(Fr, C, Hh, Hh, device=dev).to(dt).contiguous(memory_format=torch.channels_last)
bn = torch.nn.BatchNorm2d(C).to(dev).to(dt).train()
xc = xn.contiguous()
row('BatchNorm2d (train) + ReLU (256 x 109^2 x 128), channels-last', timeit(lambda: F.relu(bn(xn))), float('nan'))
row('BatchNorm2d (train) + ReLU, NCHW', timeit(lambda: F.relu(bn(xc))), float('nan'))
row('MaxPool2d(3, 2, 1) (256 x 109^2 x 128), channels-last', timeit(lambda: F.max_pool2d(xn, 3, 2, 1)), float('nan'))
row('MaxPool2d(3, 2, 1), NCHW', timeit(lambda: F.max_pool2d(xc, 3, 2, 1)), float('nan'))
print('(BatchNorm apply + ReLU and the pooling have no stand-alone kernel here: the apply rides in the consumer\'s load, the statistics in '
      'the producing GEMM\'s epilogue, the pooling is fused with both BatchNorm applies and the skip add)')

# ---- exact-erf GELU feed-forward (module.py:27-30): Linear + GELU + Linear + residual
Dm, Hd = 728, 2912
w1, w2 = (torch.randn(Hd, Dm, device=dev) * 0.04).to(dt), (torch.randn(Dm, Hd, device=dev) * 0.02).to(dt)
b1, b2 = torch.randn(Hd, device=dev).to(dt), torch.randn(Dm, device=dev).to(dt)
w1p = ops.empty_rows(Hd, Dm, dt, dev, True); w1p.copy_(w1)
w2p = ops.empty_rows(Dm, Hd, dt, dev, True); w2p.copy_(w2)


def ff_lib():
    return F.linear(F.gelu(F.linear(x, w1, b1)), w2, b2) + x


def ff_own():
    u, gl = ops.linear_fwd(xp, w1p, b1.float(), gelu=True, pad=True)
    return ops.linear_fwd(gl, w2p, b2.float(), xp, pad=True)


row('FeedForward forward: Linear + GELU + Linear + residual', timeit(ff_lib), timeit(ff_own))
