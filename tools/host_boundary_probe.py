#!/usr/bin/env python3
"""What does the reference loop's host boundary (train_CNN.py:506,512,534-536) cost at C2, piece by piece?
Variants of bench.py's with_host_boundary leg: resident input (the headline), + per-step H2D copy on a copy stream,
+ loss.item() per step, + the accuracy count (a second sync), all three (= with_host_boundary)."""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import istvt_pkg  # noqa: E402
istvt_pkg.load()
from istvt_amd import parallel, stem as stem_mod  # noqa: E402
from istvt_amd.network.vivit.vivit import XceptionVidTr  # noqa: E402

B, T, S, K = 32, 8, 224, int(os.environ.get('HB_STEPS', 20))
dev = torch.device('cuda', 0)
torch.manual_seed(0)
model = XceptionVidTr(num_frames=T, grid=stem_mod.out_side(S), depth=12, compute_dtype=torch.bfloat16).to(dev).train()
live = [p for _, p in parallel.live_named_parameters(model)]
bucket = parallel.GradBucket(live, fuse_accumulate=True, flatten_params=True)
opt = parallel.FusedSGD(bucket, lr=1e-3, momentum=0.9, zero_grad=True)
crit = torch.nn.BCEWithLogitsLoss()
g = torch.Generator().manual_seed(1)
shape = (B, T, 3, S, S)
pin = [torch.randn(shape, generator=g).pin_memory() for _ in range(2)]
pin_l = [(torch.rand((B,), generator=g) > 0.5).float().pin_memory() for _ in range(2)]
dbuf = [torch.empty(shape, device=dev) for _ in range(2)]
dlab = [torch.empty((B,), device=dev) for _ in range(2)]
dbuf[0].copy_(pin[0]); dbuf[1].copy_(pin[1]); dlab[0].copy_(pin_l[0]); dlab[1].copy_(pin_l[1])
copy_stream = torch.cuda.Stream(device=dev)
ready = [torch.cuda.Event() for _ in range(2)]


d2h = torch.cuda.Stream(device=dev)


class HostScalar:
    """value of a device scalar on the host WITHOUT draining the launch queue: a D2H copy on its own stream behind an event
    recorded right after the producer; .item() waits for that copy only"""

    def __init__(self, t):
        self.buf = torch.empty((1,), dtype=t.dtype).pin_memory()
        ev = torch.cuda.Event()
        ev.record()
        self.t = t.reshape(1)
        with torch.cuda.stream(d2h):
            d2h.wait_event(ev)
            self.buf.copy_(self.t, non_blocking=True)
            self.done = torch.cuda.Event()
            self.done.record(d2h)
        self.t.record_stream(d2h)

    def item(self):
        self.done.synchronize()
        return self.buf[0].item()


def run(n, h2d, item, acc):
    def upload(i):
        with torch.cuda.stream(copy_stream):
            dbuf[i % 2].copy_(pin[i % 2], non_blocking=True)
            dlab[i % 2].copy_(pin_l[i % 2], non_blocking=True)
            ready[i % 2].record(copy_stream)
    if h2d:
        upload(0)
    for i in range(n):
        if h2d:
            torch.cuda.current_stream(dev).wait_event(ready[i % 2])
            if i + 1 < n:
                # upload(i + 1) overwrites dbuf[(i - 1) % 2]: step i - 1's backward (conv1 weight gradient) may still be
                # reading it when no .item() drained the stream (the no-drain variants raced) -- as bench.py does
                copy_stream.wait_stream(torch.cuda.current_stream(dev))
                upload(i + 1)
        opt.zero_grad()
        logits = model(dbuf[i % 2])
        loss = crit(logits.view(-1), dlab[i % 2])
        if item == 2:
            hl = HostScalar(loss.detach())
        if acc == 2:
            ha = HostScalar(torch.sum((logits.detach().view(-1) > 0).float() == dlab[i % 2]))
        loss.backward()
        opt.step()
        if item == 1:
            loss.item()
        elif item == 2:
            hl.item()
        if acc == 1:
            int(torch.sum((logits.view(-1) > 0).float() == dlab[i % 2]).item())
        elif acc == 2:
            int(ha.item())


for name, cfg in (('resident input, no sync (the headline step)', (0, 0, 0)), ('+ H2D copy per step (copy stream)', (1, 0, 0)),
                  ('+ loss.item() per step', (0, 1, 0)), ('+ accuracy count per step (sync)', (0, 0, 1)),
                  ('loss.item() + accuracy', (0, 1, 1)), ('all three (= with_host_boundary)', (1, 1, 1)),
                  ('H2D + loss + accuracy read back behind their own events (no drain)', (1, 2, 2)),
                  ('resident input, no sync (again)', (0, 0, 0))):
    run(3, *cfg)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(K, *cfg)
    torch.cuda.synchronize()
    print('%-48s %.3f ms per step' % (name, (time.perf_counter() - t0) / K * 1e3), flush=True)
