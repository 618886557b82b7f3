"""Build libistvt_hip.so (the C-ABI HIP library of this package) in-tree with hipcc for gfx950.

    python 2023-tifs-istvt_amd/build.py [--force]

No torch dependency: plain `hipcc --offload-arch=gfx950 -fPIC -shared`.  Object files are
cached under csrc/build/ and rebuilt when a source or header is newer.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
OBJ = os.path.join(CSRC, 'build')
LIB = os.path.join(HERE, 'libistvt_hip.so')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = ['--offload-arch=gfx950', '-O3', '-fPIC', '-std=c++17', '-Wno-unused-result', '-Wno-inline-asm', '-ffp-contract=fast'] + os.environ.get('ISTVT_EXTRA_HIPCC_FLAGS', '').split()


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith('.hip'))


def _newest_header():
    return max([os.path.getmtime(os.path.join(CSRC, f)) for f in os.listdir(CSRC) if f.endswith('.h')] + [0.0])


def _compile(src):
    obj = os.path.join(OBJ, src[:-4] + '.o')
    path = os.path.join(CSRC, src)
    cmd = [HIPCC] + FLAGS + ['-Rpass-analysis=kernel-resource-usage', '-c', path, '-o', obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError('hipcc failed for %s:\n%s' % (src, r.stderr[-6000:]))
    _check_no_spills(src, r.stderr)
    return obj


# Kernels whose loads are inline-asm (the compiler does not know their results arrive late): a register spill
# there would store a not-yet-landed value, silently.  The build fails instead.
NO_SPILL_KERNELS = ('gemm256q_kernel', 'gemm256t_kernel', 'gemm256t_group_kernel')


def _check_no_spills(src, remarks):
    name = None
    for line in remarks.splitlines():
        if 'Function Name:' in line:
            name = line.split('Function Name:')[1].split()[0]
        elif 'ScratchSize [bytes/lane]:' in line and name and any(k in name for k in NO_SPILL_KERNELS):
            n = int(line.split('ScratchSize [bytes/lane]:')[1].split()[0])
            if n != 0:
                raise RuntimeError('%s: kernel %s spills %d bytes/lane to scratch; it must not (see build.py)' % (src, name, n))


def build(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(OBJ, exist_ok=True)
    srcs = _sources()
    hdr = _newest_header()
    todo = []
    for s in srcs:
        obj = os.path.join(OBJ, s[:-4] + '.o')
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(os.path.join(CSRC, s)), hdr):
            todo.append(s)
    if verbose:
        print('compiling', todo)
    if todo:
        with ThreadPoolExecutor(max_workers=min(6, len(todo))) as ex:
            list(ex.map(_compile, todo))
    objs = [os.path.join(OBJ, s[:-4] + '.o') for s in srcs]
    if todo or not os.path.exists(LIB):
        cmd = [HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('link failed:\n' + r.stderr[-6000:])
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
