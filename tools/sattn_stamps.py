#!/usr/bin/env python3
"""Per-phase timeline of the persistent spatial-attention forward (attn_spatial_pers.h, -DISTVT_SATTN_STAMP build):
    tools/build_variant.sh tmp_ab/lib_sa_stamp.so attn_spatial.hip -DISTVT_SATTN_STAMP
    ISTVT_LIB=tmp_ab/lib_sa_stamp.so python tools/sattn_stamps.py
median over workgroups of each wavefront's cycles per problem and phase (a stamp costs ~40 cycles, included)."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import istvt_pkg  # noqa: E402

istvt_pkg.load()
from istvt_amd import _lib, ops  # noqa: E402

BF, heads, dh, P = 288, 8, 64, int(os.environ.get('SA_P', 197))
dt = torch.bfloat16
qkv = ops.empty_rows(BF * P, 1536, dt, torch.device('cuda'))
qkv.copy_(torch.randn(BF * P, 1536, device='cuda'))
buf = torch.zeros(256 * 16 * 8, dtype=torch.int64, device='cuda')
raw = ctypes.CDLL(_lib.LIB_PATH)
raw.istvt_diag_sattn_stamps.argtypes = [ctypes.c_void_p]
assert raw.istvt_diag_sattn_stamps(buf.data_ptr()) == 0
for _ in range(20):
    ops.attn_spatial_fwd(qkv, BF, P, heads, dh)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    ops.attn_spatial_fwd(qkv, BF, P, heads, dh)
e1.record(); torch.cuda.synchronize()
d = buf.cpu().view(256, 16, 8).double()
per = d[:, :, :6] / d[:, :, 6:7].clamp(min=1)
names = ['issue DMA+Q (next)', 'vmcnt wait', 'barrier A', 'compute', 'stores', 'barrier B']
print('P=%d: %.1f us per launch (stamped build), %d problems per workgroup' % (P, e0.elapsed_time(e1) / 10 * 1e3, int(d[0, 0, 6])))
print('%-22s' % 'cycles per problem' + ''.join('%9s' % ('w%d' % w) for w in (0, 4, 8, 12, 13, 15)))
for j, n in enumerate(names):
    print('%-22s' % n + ''.join('%9.0f' % float(per[:, w, j].median()) for w in (0, 4, 8, 12, 13, 15)))
print('%-22s' % 'sum' + ''.join('%9.0f' % float(per[:, w, :].sum(1).median()) for w in (0, 4, 8, 12, 13, 15)))
