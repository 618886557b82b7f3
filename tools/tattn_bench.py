#!/usr/bin/env python3
"""Temporal-attention micro-benchmark at the C2 (F=9) and C4 (F=17) shapes (run on the GPU box)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import istvt_pkg  # noqa: E402

istvt_pkg.load()
from istvt_amd import ops  # noqa: E402

dt = torch.bfloat16
B, P, heads, dh = 32, 197, 8, 64
DIFF = int(os.environ.get("TB_DIFF", 2))          # 2: pre-differenced operands (what the model runs since round 4), 1: differenced scores


def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


for F in (9, 17):
    M = B * F * P
    # the model's form: one packed, line-aligned q|k|v projection, frame difference in the kernels, packed gradient
    qkv = ops.empty_rows(M, 1536, dt, 'cuda')
    qkv.copy_(torch.randn(M, 1536, device='cuda'))
    qk, v = qkv[:, :1024], qkv[:, 1024:]
    do = ops.empty_rows(M, 512, dt, 'cuda')
    do.copy_(torch.randn(M, 512, device='cuda'))
    t = timeit(lambda: ops.attn_temporal_fwd(qk, v, B, F, P, heads, dh, diff=DIFF))
    print('F=%2d fwd %7.1f us  %5.2f TB/s' % (F, t * 1e6, M * 2048 * 2 / t / 1e12), flush=True)
    t = timeit(lambda: ops.attn_temporal_bwd(qk, v, do, B, F, P, heads, dh, diff=DIFF, packed=True))
    print('F=%2d bwd %7.1f us  %5.2f TB/s' % (F, t * 1e6, M * 3584 * 2 / t / 1e12), flush=True)
