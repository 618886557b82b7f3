#!/usr/bin/env python3
"""Per-phase timeline of dwconv3x3_wgrad_kernel (stem.hip, -DISTVT_DW_STAMP build):
    tools/build_variant.sh tmp_ab/lib_dw_stamp.so stem.hip -DISTVT_DW_STAMP
    ISTVT_LIB=tmp_ab/lib_dw_stamp.so python tools/dw_stamps.py"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import istvt_pkg  # noqa: E402

istvt_pkg.load()
from istvt_amd import _lib, stem as S  # noqa: E402

dt = torch.bfloat16
Fr = 256
buf = torch.zeros(2048 * 4 * 8, dtype=torch.int64, device='cuda')
raw = ctypes.CDLL(_lib.LIB_PATH)
stamped = hasattr(raw, 'istvt_diag_dw_stamps')          # a normal build has no stamps: launch times only
if stamped:
    raw.istvt_diag_dw_stamps.argtypes = [ctypes.c_void_p]
    assert raw.istvt_diag_dw_stamps(buf.data_ptr()) == 0
names = ['top barrier', 'prefetched tile: wait + transform + LDS store', 'barrier', 'issue next tile loads', 'FMAs (+ LDS reads, end-of-tile wait)']
for H, C in ((109, 64), (109, 128), (55, 256), (28, 728)):
    M = Fr * H * H
    x = torch.randn(M, C, device='cuda').to(dt)
    bn = S.BNState(C, 'cuda'); bn.pack.normal_()
    buf.zero_()
    for _ in range(3):
        S.dwconv_wgrad(x, x, Fr, H, H, C, bn, True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        S.dwconv_wgrad(x, x, Fr, H, H, C, bn, True)
    e1.record(); torch.cuda.synchronize()
    if not stamped:
        print('H=%3d C=%3d: %.1f us per launch' % (H, C, e0.elapsed_time(e1) / 10 * 1e3))
        continue
    d = buf.cpu().view(2048, 4, 8).double()
    d = d[d[:, 0, 5] > 0]
    per = d[:, :, :5] / d[:, :, 5:6]
    med = per.reshape(-1, 5).median(0).values
    print('H=%3d C=%3d: %.1f us per launch (stamped), %d workgroups, %.0f tiles per workgroup; cycles per tile and wavefront (median):'
          % (H, C, e0.elapsed_time(e1) / 10 * 1e3, d.shape[0], float(d[:, 0, 5].median())))
    print('   ' + '   '.join('%s %.0f' % (n, float(v)) for n, v in zip(names, med)) + '   sum %.0f' % float(med.sum()))
