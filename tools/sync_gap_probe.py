#!/usr/bin/env python3
"""Where does the GPU idle when the loop syncs every step (loss.item())?  Run under `rocprofv3 --kernel-trace`: prints nothing
itself; tools/sync_gap_analyze.py reads the kernel trace.  SYNC=1: loss.item() per step; SYNC=0: none."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import istvt_pkg  # noqa: E402
istvt_pkg.load()
from istvt_amd import parallel, stem as stem_mod  # noqa: E402
from istvt_amd.network.vivit.vivit import XceptionVidTr  # noqa: E402
dev = torch.device('cuda', 0)
torch.manual_seed(0)
model = XceptionVidTr(num_frames=8, grid=stem_mod.out_side(224), depth=12, compute_dtype=torch.bfloat16).to(dev).train()
live = [p for _, p in parallel.live_named_parameters(model)]
bucket = parallel.GradBucket(live, fuse_accumulate=True, flatten_params=True)
opt = parallel.FusedSGD(bucket, lr=1e-3, momentum=0.9, zero_grad=True)
crit = torch.nn.BCEWithLogitsLoss()
x = torch.randn(32, 8, 3, 224, 224, device=dev)
y = (torch.rand(32, device=dev) > 0.5).float()
sync = os.environ.get('SYNC', '1') == '1'
for i in range(8):
    opt.zero_grad()
    loss = crit(model(x).view(-1), y)
    loss.backward()
    opt.step()
    if sync:
        loss.item()
torch.cuda.synchronize()
