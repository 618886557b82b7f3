"""One rank of tests/test_model_gpu.py::test_early_all_reduce_two_ranks (launched by torch.distributed.run, gloo backend,
every rank on cuda:0): gradients after GradBucket.all_reduce() with and without the early (overlapped) all-reduce of the
transformer's slice, and against the average of the two ranks' local gradients."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import torch.distributed as dist

import istvt_pkg

istvt_pkg.load()
from istvt_amd import parallel  # noqa: E402
from istvt_amd.network.vivit.vivit import XceptionVidTr  # noqa: E402
from istvt_amd import stem as stem_mod  # noqa: E402


def main():
    dist.init_process_group('gloo', init_method='env://')
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    torch.manual_seed(0)
    model = XceptionVidTr(num_frames=4, grid=stem_mod.out_side(96), depth=2, compute_dtype=torch.bfloat16).cuda().train()
    parallel.broadcast_parameters(model)
    named = parallel.live_named_parameters(model)
    live = [p for _, p in named]
    first_vit = next(i for i, (n, _) in enumerate(named) if n.startswith('vit.'))
    bucket = parallel.GradBucket(live, fuse_accumulate=True)
    g = torch.Generator().manual_seed(100 + rank)
    x = torch.randn((2, 4, 3, 96, 96), generator=g).cuda()
    y = (torch.rand((2,), generator=g) > 0.5).float().cuda()

    def grads(early):
        if early:
            bucket.enable_early_all_reduce(first_vit)
        bucket.zero()
        loss = torch.nn.functional.binary_cross_entropy_with_logits(model(x).view(-1), y)
        loss.backward()
        used_early = getattr(bucket, '_early_work', None) is not None
        local = None
        if not early:
            local = bucket.flat.clone()
        bucket.all_reduce()
        out = bucket.flat.clone()
        if early:
            bucket.disable_early_all_reduce()
        return out, local, used_early

    model.eval()            # running-stat BatchNorm would need identical statistics; eval stem has no backward -> train
    model.train()
    ref, local, _ = grads(False)
    # the plain path equals the mean of the ranks' local gradients
    gathered = [torch.empty_like(local) for _ in range(world)]
    dist.all_gather(gathered, local)
    mean = sum(gathered) / world
    err_mean = float((ref - mean).norm() / mean.norm())
    # the BatchNorm running statistics moved in the first pass; gradients do not depend on them in train mode
    got, _, used = grads(True)
    err_early = float((got - ref).norm() / ref.norm())
    print('rank %d: early all-reduce used=%s  |early - plain| / |plain| = %.3e  |plain - mean(local)| = %.3e'
          % (rank, used, err_early, err_mean), flush=True)
    ok = used and err_early < 1e-5 and err_mean < 1e-6
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == '__main__':
    main()
