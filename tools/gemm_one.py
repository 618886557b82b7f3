#!/usr/bin/env python3
"""one GEMM shape, a few launches (for rocprofv3 --pmc)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import istvt_pkg
istvt_pkg.load()
from istvt_amd import ops
M, K, N = [int(v) for v in os.environ.get('GB_SHAPE', '56736,2912,728').split(',')]
x = (torch.randn(M, K, device='cuda') * .5).bfloat16(); w = (torch.randn(N, K, device='cuda') * .5).bfloat16()
for _ in range(4):
    y = ops.linear_fwd(x, w)
torch.cuda.synchronize()
print('done', y.float().abs().mean().item())
