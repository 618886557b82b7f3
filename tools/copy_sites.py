#!/usr/bin/env python3
"""Which Python lines issue the small torch copies / fills / adds of a training step?  (torch.profiler with stacks.)"""
import os, sys, collections
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
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
    step()
torch.cuda.synchronize()
cnt = collections.Counter()
for ev in prof.events():
    if ev.name in ('aten::copy_', 'aten::fill_', 'aten::zero_', 'aten::add_', 'aten::contiguous', 'aten::clone', 'aten::zeros', 'aten::mul_'):
        site = next((s for s in ev.stack if 'istvt' in s or 'repo' in s), ev.stack[0] if ev.stack else '?')
        cnt[(ev.name, site.strip()[:110])] += 1
for (name, site), n in cnt.most_common(30):
    print('%4d  %-16s %s' % (n, name, site))
