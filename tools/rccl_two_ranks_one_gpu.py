#!/usr/bin/env python3
"""Can two RCCL ranks share the one GPU of the builder's box?  (torchrun --nproc-per-node 2 tools/rccl_two_ranks_one_gpu.py)
If they can, the N > 1 path of bench.py can be rehearsed with the real backend instead of gloo."""
import os
import sys
import faulthandler
import torch
import torch.distributed as dist
faulthandler.dump_traceback_later(60, exit=True)
rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
torch.cuda.set_device(0)
try:
    dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', 0))
    t = torch.full((1 << 20,), float(rank + 1), device='cuda')
    dist.all_reduce(t)
    torch.cuda.synchronize()
    print('rank', rank, 'all_reduce ok:', float(t[0]), flush=True)
    dist.destroy_process_group()
except Exception as e:      # noqa: BLE001
    print('rank', rank, 'FAILED:', type(e).__name__, str(e)[:300], flush=True)
    sys.exit(3)
