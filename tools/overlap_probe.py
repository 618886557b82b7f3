#!/usr/bin/env python3
"""Would two half-batch pipelines on two streams beat one full-batch pipeline?  The GEMMs are power / MFMA bound and leave HBM
mostly idle, the LayerNorm-class kernels are HBM bound and leave the matrix pipe idle: complementary resources, but one batch is one
dependency chain.  This probe runs a representative slice of a transformer layer -- qkv GEMM (N = 1536), LayerNorm x 2, FF1 GEMM +
GELU (N = 2912), FF2 GEMM + residual (K = 2912), LayerNorm x 2 -- (a) at M = 56 736 on one stream, (b) as two M = 28 368 copies on two
streams, half a sequence apart, each GEMM limited to 128 CUs (ops.set_cu_reserve) so that the other stream's kernels find free CUs.
Pre-allocated outputs, no autograd: GPU time only."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import istvt_pkg  # noqa: E402
istvt_pkg.load()
from istvt_amd import ops  # noqa: E402

dt = torch.bfloat16
dev = torch.device('cuda', 0)
D, H = 728, 2912


def rnd(r, c, s=0.5):
    v = ops.empty_rows(r, c, dt, dev, True)
    v.copy_((torch.randn(r, c, device=dev) * s).to(dt))
    return v


class Slice:
    def __init__(self, M):
        self.M = M
        self.x = rnd(M, D)
        self.wqkv, self.w1, self.w2 = rnd(1536, D, 0.04), rnd(H, D, 0.04), rnd(D, H, 0.02)
        self.b1, self.b2 = torch.randn(H, device=dev), torch.randn(D, device=dev)
        self.g, self.b = torch.ones(D, device=dev), torch.zeros(D, device=dev)

    def run(self):
        x = self.x
        ops.linear_fwd(x, self.wqkv, pad=True)
        y, _, _ = ops.layernorm_fwd(x, self.g, self.b, 1e-5, pad=True)
        y, _, _ = ops.layernorm_fwd(y, self.g, self.b, 1e-5, pad=True)
        u, gl = ops.linear_fwd(y, self.w1, self.b1, gelu=True, pad=True)
        z = ops.linear_fwd(gl, self.w2, self.b2, x, pad=True)
        y, _, _ = ops.layernorm_fwd(z, self.g, self.b, 1e-5, pad=True)
        y, _, _ = ops.layernorm_fwd(y, self.g, self.b, 1e-5, pad=True)


R = int(os.environ.get('OP_REPS', 20))
full = Slice(56736)
a, b = Slice(28368), Slice(28368)
q = [Slice(18912) for _ in range(3)]
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def timed(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(R):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / R


def halves():
    a.run(); b.run()


def thirds():
    for s_ in q:
        s_.run()


# warm the chip up, then alternate the variants on ONE stream (what splitting M costs or gains by itself)
for _ in range(40):
    full.run()
torch.cuda.synchronize()
for rep in range(4):
    t1, t2, t3 = timed(full.run), timed(halves), timed(thirds)
    print('one stream: M = 56736 %.3f ms | 2 x 28368 back to back %.3f ms (%+.1f %%) | 3 x 18912 %.3f ms (%+.1f %%)'
          % (t1, t2, (t2 / t1 - 1) * 100, t3, (t3 / t1 - 1) * 100), flush=True)
t_full = t1

for reserve in (0, 128):
    sa, sb = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    ops.set_cu_reserve(reserve, 0)
    for _ in range(2):
        with torch.cuda.stream(sa):
            a.run()
        with torch.cuda.stream(sb):
            b.run()
    torch.cuda.synchronize()
    e0.record()
    for i in range(R):
        with torch.cuda.stream(sa):
            if i == 0:
                sa.wait_event(e0)
            a.run()
        with torch.cuda.stream(sb):
            if i == 0:
                sb.wait_event(e0)
                y, _, _ = ops.layernorm_fwd(b.x, b.g, b.b, 1e-5, pad=True)       # a phase offset: b starts in a LayerNorm
            b.run()
    torch.cuda.current_stream().wait_stream(sa)
    torch.cuda.current_stream().wait_stream(sb)
    e1.record(); torch.cuda.synchronize()
    ops.set_cu_reserve(0, 0)
    t2 = e0.elapsed_time(e1) / R
    print('two streams, 2 x M = 28368, GEMM grids leave %3d CUs free: %.3f ms per pair of half slices (%+.1f %% vs one stream)'
          % (reserve, t2, (t2 / t_full - 1) * 100), flush=True)
