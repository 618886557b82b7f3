#!/usr/bin/env python3
"""Where does the hand-written persistent NT GEMM stand against the vendor library on the model's shapes?
torch.nn.functional.linear in bf16 on ROCm = hipBLASLt / rocBLAS (measurement only: nothing in the product path calls it).
Plain y = x W^T, M = 56 736, dense rows for the library (its natural layout), line-aligned rows for ours; warm chip, alternating."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import istvt_pkg  # noqa: E402
istvt_pkg.load()
from istvt_amd import ops  # noqa: E402

M = int(os.environ.get('GB_M', 56736))
dt = torch.bfloat16
dev = torch.device('cuda', 0)


def timeit(fn, reps=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


for K, N in [(728, 2912), (2912, 728), (728, 1536), (1536, 728), (512, 728), (728, 1024), (728, 512)]:
    xd = (torch.randn(M, K, device=dev) * 0.5).to(dt)
    wd = (torch.randn(N, K, device=dev) * 0.05).to(dt)
    xp = ops.empty_rows(M, K, dt, dev, True); xp.copy_(xd)
    wp = ops.empty_rows(N, K, dt, dev, True); wp.copy_(wd)
    fl = 2.0 * M * N * K
    rows = []
    for rep in range(2):
        t_lib = timeit(lambda: torch.nn.functional.linear(xd, wd))
        t_own = timeit(lambda: ops.linear_fwd(xp, wp, pad=True))
        rows.append((t_lib, t_own))
    t_lib, t_own = min(r[0] for r in rows), min(r[1] for r in rows)
    print('M=%d K=%4d N=%4d   vendor library %7.1f us %7.1f TF/s   |   gemm256q %7.1f us %7.1f TF/s   (ours / library time: %.3f)'
          % (M, K, N, t_lib * 1e6, fl / t_lib / 1e12, t_own * 1e6, fl / t_own / 1e12, t_own / t_lib), flush=True)
