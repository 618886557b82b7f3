"""istvt_amd — MI355X-native (gfx950) implementation of the ISTVT video-clip hot path.

The directory name (``2023-tifs-istvt_amd``) is not a valid Python identifier; load the
package through ``istvt_pkg.load()`` at the repo root, which registers it as ``istvt_amd``.

Layout
    csrc/          hand-written HIP kernels + the C ABI (libistvt_hip.so, see include/istvt_hip.h)
    _lib.py        ctypes binding of the C ABI (fails loudly when the library is missing)
    ops.py         tensor-level launch wrappers (shape checks, stream, dtype codes)
    functional.py  torch.autograd.Function glue (autograd plumbing only)
    network/       nn.Modules mirroring the reference's constructor/forward signatures
    parallel.py    data-parallel gradient bucket (one RCCL all-reduce per step)
"""
__all__ = ['ops', 'functional', 'network', 'parallel']
