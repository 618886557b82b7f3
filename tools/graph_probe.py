#!/usr/bin/env python3
"""parallel.StepGraphs on the GPU box: (1) the graphed training loop gives the BITS of the launch-by-launch loop (two models
from one seed, one of them with enable_step_graphs(), N steps of SGD: logits per step and the flat parameter buffer);
(2) host time to enqueue a step and the step time of the reference's literal loop (loss.item() every step), both ways."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import istvt_pkg
istvt_pkg.load()
from istvt_amd import parallel, stem as stem_mod
from istvt_amd.network.vivit.vivit import XceptionVidTr

cfg = sys.argv[1] if len(sys.argv) > 1 else 'C2'
B, T, S, depth, dtype = {'C2': (32, 8, 224, 12, torch.bfloat16), 'small': (2, 8, 224, 2, torch.bfloat16),
                         'C1': (1, 4, 96, 2, torch.float32)}[cfg]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6


def make(graphs):
    torch.manual_seed(0)
    model = XceptionVidTr(num_frames=T, grid=stem_mod.out_side(S), depth=depth, compute_dtype=dtype).cuda().train()
    live = [p for _, p in parallel.live_named_parameters(model)]
    bucket = parallel.GradBucket(live, fuse_accumulate=True, flatten_params=True)
    opt = parallel.FusedSGD(bucket, lr=1e-3, momentum=0.9, zero_grad=True)
    if graphs:
        model.enable_step_graphs(True)
    return model, bucket, opt


g = torch.Generator().manual_seed(1)
xs = [torch.randn(B, T, 3, S, S, generator=g).cuda() for _ in range(2)]
ys = [(torch.rand(B, generator=g) > 0.5).float().cuda() for _ in range(2)]
crit = torch.nn.BCEWithLogitsLoss()


def run(model, opt, n, item=False):
    outs = []
    for i in range(n):
        opt.zero_grad()
        logits = model(xs[i % 2])
        loss = crit(logits.view(-1), ys[i % 2])
        loss.backward()
        opt.step()
        outs.append(logits.detach().clone())
        if item:
            loss.item()
    return outs


res = {}
for graphs in (False, True):
    model, bucket, opt = make(graphs)
    outs = run(model, opt, steps)
    torch.cuda.synchronize()
    res[graphs] = (torch.stack(outs).cpu(), bucket.flat_params.detach().cpu().clone())
    if graphs:
        print('StepGraphs stats:', model._step_graphs.stats, 'last fallback reason:', model._step_graphs.last_reason)
    # timing: enqueue time of a step (no sync inside), and the literal loop with loss.item() per step
    for _ in range(3):
        run(model, opt, 2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(model, opt, 2)
    t_enq = (time.perf_counter() - t0) / 2
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(model, opt, 10)
    torch.cuda.synchronize()
    t_free = (time.perf_counter() - t0) / 10
    t0 = time.perf_counter()
    run(model, opt, 10, item=True)
    torch.cuda.synchronize()
    t_item = (time.perf_counter() - t0) / 10
    print('%s %-18s host enqueue %.3f ms / step; step %.3f ms free-running, %.3f ms with loss.item() every step; reserved %.1f GB'
          % (cfg, 'HIP graphs' if graphs else 'launch by launch', t_enq * 1e3, t_free * 1e3, t_item * 1e3,
             torch.cuda.memory_reserved() / 2**30), flush=True)
    del model, bucket, opt
    torch.cuda.empty_cache()

lo, hi = res[False], res[True]
print('logits identical per step:', [bool(torch.equal(a, b)) for a, b in zip(lo[0], hi[0])])
print('flat parameters after %d steps identical: %s (max |diff| %.3e)' % (steps, bool(torch.equal(lo[1], hi[1])), float((lo[1] - hi[1]).abs().max())))
assert torch.equal(lo[0], hi[0]) and torch.equal(lo[1], hi[1]), 'graphed loop differs from the launch-by-launch loop'
print('OK')
