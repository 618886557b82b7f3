#!/usr/bin/env python3
"""C2 step time: {launch by launch, HIP graphs} x {weight gradients on the side stream, one stream}: does the captured graph keep the
side stream's concurrency?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import istvt_pkg
istvt_pkg.load()
from istvt_amd import parallel, stem as stem_mod, functional as Fn
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
    opt.zero_grad(); loss = crit(model(x).view(-1), y); loss.backward(); opt.step()
def timed(n=20):
    for _ in range(4): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): step()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for rnd in range(2):
    for overlap in (True, False):
        Fn.set_wgrad_overlap(overlap)
        for graphs in (False, True):
            model.enable_step_graphs(graphs)
            print('round %d  %-16s %-22s %.3f ms per step' % (rnd, 'HIP graphs' if graphs else 'launch by launch',
                                                              'side-stream wgrads' if overlap else 'one stream', timed()), flush=True)
            model.enable_step_graphs(False)
