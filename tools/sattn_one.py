#!/usr/bin/env python3
"""One spatial-attention forward + backward at the C2 shape (288 frames x 8 heads x 197 tokens), a few repetitions:
the workload for `rocprofv3 --pmc ... -- python3 tools/sattn_one.py`."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import istvt_pkg  # noqa: E402

istvt_pkg.load()
from istvt_amd import ops  # noqa: E402

BF, heads, dh, P = 288, 8, 64, int(os.environ.get('SA_P', 197))
qkv = torch.randn(BF * P, 1536, device='cuda').to(torch.bfloat16)
do = torch.randn(BF * P, 512, device='cuda').to(torch.bfloat16)
for _ in range(int(os.environ.get('SA_REPS', 4))):
    out, lse = ops.attn_spatial_fwd(qkv, BF, P, heads, dh)
    ops.attn_spatial_bwd(qkv, out, do, lse, BF, P, heads, dh)
torch.cuda.synchronize()
