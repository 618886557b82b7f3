#!/usr/bin/env python3
"""C1 (BASELINE.json configs[0]: 1 clip, T=4, 96x96, depth 2, float32, train-mode forward under no_grad) on the GPU box:
N forwards launch by launch and through the captured graph; run under `rocprofv3 --kernel-trace --stats` for the kernel list."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import istvt_pkg
istvt_pkg.load()
from istvt_amd import stem as stem_mod
from istvt_amd.network.vivit.vivit import XceptionVidTr
dt = torch.bfloat16 if (len(sys.argv) > 1 and sys.argv[1] == 'bf16') else torch.float32
torch.manual_seed(0)
model = XceptionVidTr(num_frames=4, grid=stem_mod.out_side(96), depth=2, compute_dtype=dt).cuda().train()
x = torch.randn(1, 4, 3, 96, 96).cuda()
for graphs in (False, True):
    model.enable_step_graphs(graphs)
    with torch.no_grad():
        for _ in range(5):
            y = model(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            y = model(x)
        torch.cuda.synchronize()
    print('C1 %s %s: %.3f ms per forward' % (str(dt).split('.')[-1], 'graph replay' if graphs else 'launch by launch', (time.perf_counter() - t0) / 50 * 1e3), flush=True)
