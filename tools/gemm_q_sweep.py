#!/usr/bin/env python3
"""Same-box sweep of the persistent NT GEMM (gemm256q) at the model's shapes: one child process per variant
(the library reads its tuning variables once per process), every child times every shape.

    python tools/gemm_q_sweep.py                       # parent: runs the variant list below
    GQ_CHILD=1 python tools/gemm_q_sweep.py            # child: times the shapes under the current environment

Needs a -DISTVT_TUNE (and for QDBG variants -DISTVT_GEMM_DIAG) build of gemm.hip: tools/build_variant.sh.
GQ_VARIANTS = ';'-separated list of 'name:VAR=val,VAR=val'; GQ_SHAPES = ';'-separated 'K,N[,epi]' (epi: p plain,
r bias+residual, g GELU forward, b GELU backward)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

DEFAULT_SHAPES = '728,728;728,512;728,1024;728,1536;2912,728;512,728;1536,728;728,2912,g;2912,728,r;728,2912,b'
DEFAULT_VARIANTS = 'base:;slab:ISTVT_GEMM_WALK=1'


def child():
    import torch
    import istvt_pkg
    istvt_pkg.load()
    from istvt_amd import _lib, ops
    if os.environ.get('GB_LIB'):
        _lib.LIB_PATH = os.path.abspath(os.environ['GB_LIB'])
    M = int(os.environ.get('GB_M', 56736))
    reps = int(os.environ.get('GB_REPS', 10))
    dt = torch.bfloat16

    def rnd(r, c):
        v = ops.empty_rows(r, c, dt, 'cuda', True)
        v.copy_((torch.randn(r, c, device='cuda') * 0.5).to(dt))
        return v

    def timeit(fn):
        fn(); fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e-3

    out = []
    for spec in os.environ.get('GQ_SHAPES', DEFAULT_SHAPES).split(';'):
        f = spec.split(',')
        K, N, epi = int(f[0]), int(f[1]), (f[2] if len(f) > 2 else 'p')
        x, w = rnd(M, K), rnd(N, K)
        b = torch.randn(N, device='cuda')
        if epi == 'p':
            fn = lambda: ops.linear_fwd(x, w, pad=True)
        elif epi == 'r':
            res = rnd(M, N)
            fn = lambda: ops.linear_fwd(x, w, b, res, pad=True)
        elif epi == 'g':
            fn = lambda: ops.linear_fwd(x, w, b, gelu=True, pad=True)
        else:
            # GELU backward: dx_hidden = (dy W2) * gelu'(u): A = dy [M, K], B = W2^T operand [N, K], u [M, N]
            u = rnd(M, N)
            y = ops.empty_rows(M, N, dt, 'cuda', True)
            fn = lambda: ops.gemm_raw(x, x.stride(0), True, w, w.stride(0), True, y, y.stride(0), M, N, K, C2=u, epi=2)
        t = timeit(fn)
        out.append('%s %6.1fus %6.1fTF' % (spec, t * 1e6, 2.0 * M * N * K / t / 1e12))
        del x, w
    print('RESULT ' + os.environ.get('GQ_NAME', '?') + ' | ' + ' | '.join(out), flush=True)


def parent():
    for v in os.environ.get('GQ_VARIANTS', DEFAULT_VARIANTS).split(';'):
        name, _, kv = v.partition(':')
        env = dict(os.environ, GQ_CHILD='1', GQ_NAME=name)
        for item in kv.split(','):
            if item:
                k, _, val = item.partition('=')
                env[k] = val
        r = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, stdout=subprocess.PIPE,
                           stderr=subprocess.STDOUT, text=True, timeout=600)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith('RESULT')]
        print(lines[0] if lines else 'FAILED %s rc=%d\n%s' % (name, r.returncode, r.stdout[-1500:]), flush=True)


if __name__ == '__main__':
    child() if os.environ.get('GQ_CHILD') == '1' else parent()
