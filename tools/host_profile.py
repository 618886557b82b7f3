#!/usr/bin/env python3
"""cProfile of the host side of one training step (run on the GPU box): where do the ~38 ms of Python per step go?"""
import cProfile, os, pstats, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import istvt_pkg
istvt_pkg.load()
from istvt_amd import parallel, stem as stem_mod
from istvt_amd.network.vivit.vivit import XceptionVidTr
torch.manual_seed(0)
model = XceptionVidTr(num_frames=8, grid=stem_mod.out_side(224), depth=12, compute_dtype=torch.bfloat16).cuda().train()
live = [p for _, p in parallel.live_named_parameters(model)]
bucket = parallel.GradBucket(live, fuse_accumulate=True, flatten_params=True)
opt = parallel.FusedSGD(bucket, lr=1e-3, momentum=0.9, zero_grad=True)
x = torch.randn(32, 8, 3, 224, 224).cuda()
y = (torch.rand(32) > 0.5).float().cuda()
crit = torch.nn.BCEWithLogitsLoss()
def step():
    opt.zero_grad()
    loss = crit(model(x).view(-1), y)
    loss.backward()
    opt.step()
for _ in range(3):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(28)
