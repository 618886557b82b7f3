#!/usr/bin/env python3
"""Which aten::copy_ / aten::clone / aten::contiguous calls does one C2 training step make on CUDA tensors (the
__amd_rocclr_copyBuffer launches of the trace), and from where?  Run on the GPU box."""
import os, sys, collections, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, istvt_pkg
istvt_pkg.load()
from istvt_amd import parallel
from istvt_amd.network.vivit import vivit
m = vivit.XceptionVidTr(num_frames=8, grid=14, depth=2, compute_dtype=torch.bfloat16).cuda().train()
live = [p for _, p in parallel.live_named_parameters(m)]
bucket = parallel.GradBucket(live, fuse_accumulate=True, flatten_params=True)
opt = parallel.FusedSGD(bucket, lr=1e-3, momentum=0.9)
x = torch.randn(4, 8, 3, 224, 224, device='cuda'); lab = torch.tensor([1., 0., 1., 0.], device='cuda')
def step():
    bucket.zero()
    out = m(x)
    torch.nn.functional.binary_cross_entropy_with_logits(out.view(-1), lab).backward()
    opt.step()
for _ in range(2): step()
sites = collections.Counter()
orig = torch.Tensor.copy_
def spy(self, src, *a, **k):
    if self.is_cuda:
        fr = traceback.extract_stack(limit=4)[:-1]
        sites[' <- '.join('%s:%d' % (os.path.basename(f.filename), f.lineno) for f in reversed(fr)) + ' %s' % (tuple(self.shape),)] += 1
    return orig(self, src, *a, **k)
torch.Tensor.copy_ = spy
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU]) as prof:
    step()
torch.Tensor.copy_ = orig
torch.cuda.synchronize()
for k, v in sites.most_common(20): print(v, k)
ev = collections.Counter(e.name for e in prof.events() if e.name in ('aten::copy_', 'aten::clone', 'aten::contiguous', 'aten::fill_', 'aten::zero_', 'aten::cat', 'aten::to', 'aten::_to_copy'))
print(dict(ev))
