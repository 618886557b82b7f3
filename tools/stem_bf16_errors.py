#!/usr/bin/env python3
"""bf16 stem against the bf16-storage emulation of the oracle: forward error and the per-parameter gradient errors
(1 - cosine) that tests/gpu_checks.py::stem_vs_oracle folds into one number (run on the GPU box)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'tests', 'golden')):
    sys.path.insert(0, p)
os.environ['STEM_TOPK'] = '40'
import torch  # noqa: E402
import gpu_checks as G  # noqa: E402

for side in (139, 224):
    e, t = G.stem_vs_oracle(torch.bfloat16, side, init='random', need_dx=(side == 139), tol=0.15)
    print(side, 'forward %.5f (bound %.3f); worst gradient 1-cos %.4f (tolerance %.2f)' % (G.stem_vs_oracle.last_y, G.Y_TOL_BF16, e, t))
    for k, v in G.stem_vs_oracle.last:
        print('    %-40s %.5f' % (k, v))
