#!/usr/bin/env python3
"""How many bf16 weight operands does ops.refresh_stale_operands() re-cast per call in a steady training loop?  (Expected: all
of them in the first call of a step -- one grouped launch -- and none in the second; a count of 0 every time would mean the
'held by a live graph' guard misfires and every operand is re-made lazily.)  Run on the GPU box."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, istvt_pkg
istvt_pkg.load()
from istvt_amd import ops, parallel
from istvt_amd.network.vivit import vivit
counts = []
orig = ops.refresh_stale_operands
def logged():
    n = orig(); counts.append(n); return n
ops.refresh_stale_operands = logged
vivit.ops.refresh_stale_operands = logged
import istvt_amd.stem as stem
stem.ops.refresh_stale_operands = logged
m = vivit.XceptionVidTr(num_frames=4, grid=7, depth=2, compute_dtype=torch.bfloat16).cuda().train()
live = [p for _, p in parallel.live_named_parameters(m)]
bucket = parallel.GradBucket(live, fuse_accumulate=True, flatten_params=True)
opt = parallel.FusedSGD(bucket, lr=1e-3, momentum=0.9) if hasattr(parallel, 'FusedSGD') else None
x = torch.randn(2, 4, 3, 112, 112, device='cuda'); lab = torch.tensor([1., 0.], device='cuda')
for i in range(4):
    bucket.zero()
    out = m(x)
    torch.nn.functional.binary_cross_entropy_with_logits(out.view(-1), lab).backward()
    if opt is not None: opt.step()
    else:
        with torch.no_grad():
            for p in live: p.add_(p.grad, alpha=-1e-3)
torch.cuda.synchronize()
print('refresh counts per call:', counts, ' operands cached:', len(ops._operands))
