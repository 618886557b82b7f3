"""One rank of tests/test_model_gpu.py::test_data_parallel_equals_single_process (torch.distributed.run, gloo, every rank
on cuda:0): SURVEY 8(e)'s equivalence test on the REAL model.

  A. train mode (per-rank BatchNorm statistics, as the reference's nn.DataParallel): the W-rank gradients after
     GradBucket.all_reduce() == the mean of single-process runs over the shards one after the other.
  B. stem in eval() (running statistics: nothing couples the clips any more): the W-rank gradients == ONE process on
     the concatenated batch, up to float32 reduction order -- the exact W == 1 test bed.
  C. the fused SGD step with the 1 / W folded in (GradBucket.defer_scale) gives the parameters of the single-process
     step on the concatenated batch.
  D. bfloat16, the configuration bench.py runs: transformer slice of the bucket all-reduced early (from the token-assembly
     backward) with the layer's weight gradients queued and launched as one grid on the side stream == the same ranks
     with everything on one stream, one weight gradient per launch, one blocking all-reduce after backward.
"""
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

T, SIDE, DEPTH, PER = 4, 96, 2, 2


def build(state=None, dtype=torch.float32):
    torch.manual_seed(0)
    m = XceptionVidTr(num_frames=T, grid=stem_mod.out_side(SIDE), depth=DEPTH, compute_dtype=dtype).cuda().train()
    if state is not None:
        m.load_state_dict(state)
    return m


def shard_data(r):
    g = torch.Generator().manual_seed(100 + r)
    return torch.randn((PER, T, 3, SIDE, SIDE), generator=g).cuda(), (torch.rand((PER,), generator=g) > 0.5).float().cuda()


def local_grads(model, x, y, fused_opt=False):
    named = parallel.live_named_parameters(model)
    bucket = parallel.GradBucket([p for _, p in named], fuse_accumulate=True, flatten_params=fused_opt)
    bucket.zero()
    torch.nn.functional.binary_cross_entropy_with_logits(model(x).view(-1), y).backward()
    return bucket


def rel(a, b):
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def main():
    dist.init_process_group('gloo', init_method='env://')
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    model = build()
    parallel.broadcast_parameters(model)
    state0 = {k: v.clone() for k, v in model.state_dict().items()}
    x, y = shard_data(rank)
    errs = {}

    # ---- A: train mode
    bucket = local_grads(model, x, y)
    bucket.all_reduce()
    got = bucket.flat.clone()
    if rank == 0:
        acc = None
        for r in range(world):
            b = local_grads(build(state0), *shard_data(r))
            acc = b.flat.clone() if acc is None else acc + b.flat
        errs['A'] = rel(got, acc / world)

    # ---- B: stem in eval mode -> W ranks == one process on the concatenated batch
    m2 = build(state0)
    m2.xcep.eval()
    b2 = local_grads(m2, x, y)
    b2.all_reduce()
    got2 = b2.flat.clone()
    if rank == 0:
        m1 = build(state0)
        m1.xcep.eval()
        xs, ys = zip(*(shard_data(r) for r in range(world)))
        b1 = local_grads(m1, torch.cat(xs), torch.cat(ys))
        errs['B'] = rel(got2, b1.flat)

    # ---- C: fused SGD step with the deferred 1 / W
    m3 = build(state0)
    m3.xcep.eval()
    b3 = local_grads(m3, x, y, fused_opt=True)
    opt = parallel.FusedSGD(b3, lr=0.05, momentum=0.9, zero_grad=True)
    b3.all_reduce()
    assert b3.grad_scale == 1.0 / world and b3.defer_scale
    opt.step()
    if rank == 0:
        m4 = build(state0)
        m4.xcep.eval()
        b4 = local_grads(m4, torch.cat(xs), torch.cat(ys), fused_opt=True)
        o4 = parallel.FusedSGD(b4, lr=0.05, momentum=0.9, zero_grad=True)
        o4.step()
        errs['C'] = rel(b3.flat_params, b4.flat_params)
        errs['C_moved'] = rel(b3.flat_params, torch.cat([p.detach().flatten() for _, p in parallel.live_named_parameters(build(state0))]))

    # ---- D: bf16, early all-reduce + grouped side-stream weight gradients vs the plain order
    from istvt_amd import functional as Fn
    res = []
    for fancy in (True, False):
        Fn.set_wgrad_overlap(fancy)
        Fn.set_wgrad_group(8 if fancy else 1)
        m5 = build(state0, torch.bfloat16)
        named = parallel.live_named_parameters(m5)
        b5 = parallel.GradBucket([p for _, p in named], fuse_accumulate=True)
        if fancy:
            b5.enable_early_all_reduce(next(i for i, (n, _) in enumerate(named) if n.startswith('vit.')))
        for it in range(2):                     # twice: the second step runs with recycled allocator blocks
            b5.zero()
            torch.nn.functional.binary_cross_entropy_with_logits(m5(x * (1.0 + 0.5 * it)).view(-1), y).backward()
            b5.all_reduce()
        res.append(b5.flat.clone())
        b5.disable_early_all_reduce()
    Fn.set_wgrad_overlap(True)
    Fn.set_wgrad_group(8)
    errs['D'] = rel(res[0], res[1])

    ok = True
    if rank == 0:
        print('data-parallel equivalence (W=%d): %s' % (world, errs), flush=True)
        ok = errs['A'] < 1e-5 and errs['B'] < 2e-5 and errs['C'] < 1e-6 and errs['C_moved'] > 1e-6
    ok = ok and errs['D'] < 1e-5 and float(res[1].norm()) > 0
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == '__main__':
    main()
