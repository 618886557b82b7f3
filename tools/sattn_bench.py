#!/usr/bin/env python3
"""Spatial-attention micro-benchmark: time vs tokens per frame (run on the GPU box)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import istvt_pkg  # noqa: E402

istvt_pkg.load()
from istvt_amd import _lib, ops  # noqa: E402

if os.environ.get('GB_LIB'):            # A/B a variant build of the library (same box, one process per variant)
    _lib.LIB_PATH = os.path.abspath(os.environ['GB_LIB'])

dt = torch.bfloat16
BF, heads, dh = int(os.environ.get('SA_BF', 288)), 8, 64


def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


for P in [int(v) for v in os.environ.get('SA_PS', '64,128,197,224,256,362').split(',')]:
    qkv = torch.randn(BF * P, 1536, device='cuda').to(dt)
    do = torch.randn(BF * P, 512, device='cuda').to(dt)
    out, lse = ops.attn_spatial_fwd(qkv, BF, P, heads, dh)
    tf = timeit(lambda: ops.attn_spatial_fwd(qkv, BF, P, heads, dh))
    tb = timeit(lambda: ops.attn_spatial_bwd(qkv, out, do, lse, BF, P, heads, dh))
    fl = 4.0 * BF * heads * P * P * dh
    print('P=%3d fwd %7.1f us %6.1f TF/s   bwd %7.1f us %6.1f TF/s' % (P, tf * 1e6, fl / tf / 1e12, tb * 1e6, 2.5 * fl / tb / 1e12), flush=True)
