"""Tensor-level launch wrappers around the C ABI (include/istvt_hip.h).

Every function checks shapes/dtypes in Python *before* launching (so bad geometry raises a
Python exception like the reference does, SURVEY.md 8(b) "Error convention"), passes raw device
pointers + the current torch stream, and never synchronises.  Inputs must live on a ROCm
device: there is deliberately no CPU path.
"""
from __future__ import annotations

import os
import sys
import weakref
from typing import Optional, Tuple

import torch

from . import _lib

Tensor = torch.Tensor
_DT = {torch.float32: 0, torch.bfloat16: 1}


def dtype_code(t: Tensor) -> int:
    try:
        return _DT[t.dtype]
    except KeyError:
        raise TypeError('istvt_amd supports float32 and bfloat16 activations, got %s' % t.dtype) from None


def _req(t: Tensor, name: str = 'tensor') -> Tensor:
    if not t.is_cuda:
        raise RuntimeError('istvt_amd: %s must be on a ROCm device (no CPU fallback exists for the ISTVT hot path)' % name)
    return t


def _c(t: Tensor) -> Tensor:
    return t if t.is_contiguous() else t.contiguous()


def _ptr(t: Optional[Tensor]):
    return None if t is None else t.data_ptr()


def _stream() -> int:
    # the raw handle of torch's current stream; torch.cuda.current_stream() builds a Stream object through three
    # Python layers (10 us x ~1500 launches per step)
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())


# ------------------------------------------------------------------------------------------
# Row-strided activations.  The GEMMs stage their operands with LDS-DMA, which is priced per cache
# line touched: a [M][728] bf16 tensor has 1456-byte rows, so every 128-byte piece of a row straddles
# two lines.  Transformer activations (and the bf16 operand copies of the weights) are therefore
# allocated with rows padded to a multiple of 64 elements and handed around as [M, D] VIEWS of the
# [M, ld] buffer; every kernel takes the row stride.  Pad columns are never read and never written.
ROW_ALIGN = 64


def _parse_pad_mod(spec: str) -> Tuple[int, int]:
    try:
        m, r = (int(v) for v in spec.split(','))
    except ValueError:
        raise ValueError("ISTVT_PAD_MOD must be 'm,r' (lines per row = r mod m), got %r" % (spec,)) from None
    if m < 1 or not 0 <= r < m:
        raise ValueError('ISTVT_PAD_MOD=%r: need m >= 1 and 0 <= r < m' % (spec,))
    return m, r


_PAD_MOD = _parse_pad_mod(os.environ.get('ISTVT_PAD_MOD', '2,1'))   # (m, r): 64-element units per row = r mod m


def pad_ld(n: int) -> int:
    q = (n + ROW_ALIGN - 1) // ROW_ALIGN
    m, r = _PAD_MOD
    return (q + ((r - q) % m)) * ROW_ALIGN


def empty_rows(M: int, D: int, dtype, device, pad: bool = True) -> Tensor:
    ld = pad_ld(D) if pad else D
    buf = torch.empty((M, ld), dtype=dtype, device=device)
    return buf if ld == D else buf[:, :D]


def zeros_rows(M: int, D: int, dtype, device, pad: bool = True) -> Tensor:
    ld = pad_ld(D) if pad else D
    buf = torch.zeros((M, ld), dtype=dtype, device=device)
    return buf if ld == D else buf[:, :D]


def rows(t: Tensor) -> Tuple[Tensor, int]:
    """(2-D view [M, D] with unit column stride, row stride in elements); copies only if the layout is not that."""
    t2 = t if t.dim() == 2 else t.reshape(-1, t.shape[-1])
    M, D = t2.shape
    if M == 1:
        return (t2 if t2.stride(1) == 1 else t2.contiguous()), D
    if t2.stride(1) != 1 or t2.stride(0) < D or t2.stride(0) % 8 != 0 or t2.data_ptr() % 16 != 0:
        t2 = t2.contiguous()
    return t2, t2.stride(0)


# ------------------------------------------------------------------------------------------
# fp32 master weight -> compute-dtype operand, cached per parameter version
_wcache: dict = {}
_wepoch = [0]          # bumped by writers that modify parameters behind autograd's back (parallel.FusedSGD / FusedAdamW)


def invalidate_weight_cache():
    """A kernel wrote parameters through raw pointers (no ``_version`` bump): drop every cached operand copy."""
    _wepoch[0] += 1


# bf16 GEMM operands that come out of the fused cast + transpose pass (weight_as(pad=True) / weight_cat_as):
# cache key -> (weakrefs of the fp32 parameters stacked in the operand, operand [sum R, C], transpose [C, sum R]).
# refresh_stale_operands() re-casts every one whose parameters have changed in ONE grouped launch.
_operands: dict = {}


def _versions(ws) -> tuple:
    return tuple(w._version for w in ws) + (_wepoch[0],)


def _holders(t: Tensor) -> tuple:
    """(Python references, C++ references to the TensorImpl -- autograd SavedVariables --, tensors sharing the storage --
    views and slices) of `t`, each including what THIS call adds.  Only differences between two calls made the same way
    mean anything: _operand_holders() is the one place that calls it."""
    try:
        return (sys.getrefcount(t), t._use_count(), torch._C._storage_Use_Count(t.untyped_storage()._cdata))
    except AttributeError:
        # private torch APIs (Tensor._use_count, torch._C._storage_Use_Count) gone in this build: report "held by somebody",
        # so that no operand is ever rewritten in place (it is re-made instead: correct, one launch per weight slower)
        global _holders_warned
        if not _holders_warned:
            _holders_warned = True
            import warnings
            warnings.warn('istvt_amd.ops: torch lacks the reference-count introspection used to refresh bf16 operand copies in '
                          'place; falling back to re-making them (slower, still correct)')
        return (1 << 30, 1 << 30, 1 << 30)


_holders_warned = False


def _operand_holders(key) -> tuple:
    _refs, out, wt = _operands[key]
    return _holders(out) + _holders(wt)


def _idle_holder_counts(view: bool) -> tuple:
    """what _operand_holders() returns for an operand pair that NOTHING but the two caches holds: measured, not assumed --
    a throw-away pair is put through the same dict / tuple structure and the same call (so another Python version's
    reference accounting, or a refactor of the cache entries, moves the baseline and the check together)."""
    if view not in _idle_counts:
        saved = (dict(_operands), dict(_wcache))
        try:
            key = ('calibration', view)
            mk = (lambda: torch.empty((2, 16))[:, :8]) if view else (lambda: torch.empty((2, 8)))
            # Under the mode real operands are made in (a normal forward): the first stale refresh now runs inside the
            # optimizer step, possibly under torch.inference_mode() / no_grad, where a slice carries no ._base and its storage
            # has one holder fewer -- a baseline taken there made every later refresh look "held" and silently fell back to 84
            # lazy casts per step.
            with torch.inference_mode(False), torch.enable_grad():
                out, wt = mk(), mk()
            if (out._base is not None) != bool(view) or (wt._base is not None) != bool(view):
                raise RuntimeError('istvt_amd.ops: calibration operand is %sa view (expected view=%r)'
                                   % ('' if out._base is not None else 'not ', view))
            _operands[key] = ((), out, wt)
            _wcache[key] = (None, None, out)
            _wcache[(id(out), 'T')] = (None, 0, wt)
            del out, wt
            _idle_counts[view] = _operand_holders(key)
        finally:
            _operands.clear(); _operands.update(saved[0])
            _wcache.clear(); _wcache.update(saved[1])
    return _idle_counts[view]


_idle_counts: dict = {}


def refresh_stale_operands() -> int:
    """Re-cast, in one grouped launch (istvt_cast_transpose_group), the bf16 operand copies (W and W^T) of every weight
    that changed since they were made -- i.e. of all of them after an optimizer step.  The models call this at the start
    of a forward pass; without it the same work happens lazily, one launch per weight (84 per step at depth 12, each far
    shorter than the ~5 us a launch occupies the queue for).  Returns the number of weights re-cast."""
    import ctypes as C
    _refresh_derived()
    static = static_addresses()
    if static:
        _refresh_plain_copies()
    todo = []
    for key, (refs, out, wt) in list(_operands.items()):
        ws = [r() for r in refs]
        hit = _wcache.get(key)
        if any(w is None for w in ws) or hit is None or hit[2] is not out:
            _operands.pop(key, None)
            continue
        ver = _versions(ws)
        if hit[1] != ver:
            # Rewritten IN PLACE only when nothing but the two caches (and this loop) holds the operand or its transpose.  A
            # live autograd graph that saved them (ctx attributes or save_for_backward of RepChainFn / StemFn / LinearFn:
            # forward A -> optimizer step -> forward B -> backward A) must still find forward A's weights: those copies are
            # left alone -- dropped from the caches, so the next use makes fresh ones -- and die with the graph.
            # Holders are counted three ways (_holders: Python references, C++ references such as SavedVariable, tensors
            # sharing the storage such as views / slices) and compared with the counts of an operand pair that only the
            # caches hold, measured once through the same structures (_idle_holder_counts).
            del out, wt, hit
            idle = (_idle_holder_counts(_operands[key][1]._base is not None)[:3]
                    + _idle_holder_counts(_operands[key][2]._base is not None)[3:])
            if not static and any(a > b for a, b in zip(_operand_holders(key), idle)):
                _refs, out, _wt = _operands[key]
                _operands.pop(key, None)
                _wcache.pop(key, None)
                _wcache.pop((id(out), 'T'), None)
                continue
            _refs, out, wt = _operands[key]
            todo.append((key, ws, out, wt, ver, _wcache[key]))
    if not todo:
        return 0
    srcs, ins, ldi, outs, ldo, outts, ldt, Rs, Cs = [], [], [], [], [], [], [], [], []
    for key, ws, out, wt, ver, hit in todo:
        es, r0 = out.element_size(), 0
        for w in ws:
            w2 = _c(w.detach().reshape(w.shape[0], -1))
            srcs.append(w2)                       # kept alive until the launch is enqueued
            n, K = w2.shape
            ins.append(w2.data_ptr()); ldi.append(K)
            outs.append(out.data_ptr() + r0 * out.stride(0) * es); ldo.append(out.stride(0))
            outts.append(wt.data_ptr() + r0 * es); ldt.append(wt.stride(0))
            Rs.append(n); Cs.append(K)
            r0 += n
    n = len(ins)
    PA, LA, IA = C.c_void_p * n, C.c_long * n, C.c_int * n
    _lib.check(_lib.lib().istvt_cast_transpose_group(n, PA(*ins), LA(*ldi), PA(*outs), LA(*ldo), PA(*outts), LA(*ldt),
                                                     IA(*Rs), IA(*Cs), _stream()), 'istvt_cast_transpose_group')
    for key, ws, out, wt, ver, hit in todo:
        _wcache[key] = (hit[0], ver, out)
    return n

# ------------------------------------------------------------------------------------------
# Derived weight layouts that are built with a few torch ops (the stem's tap-major depthwise weights, conv1 / conv2 in
# GEMM form): cached per parameter version like the operand copies, and -- like them -- rebuilt by
# refresh_stale_operands(), i.e. right behind the optimizer step when a fused optimizer drives the loop, instead of at
# the head of the next forward pass (where a loop that syncs every step has the GPU waiting for the host).
_derived: dict = {}


def derived(key, w: Tensor, builder):
    """builder(w) -> Tensor, cached until `w` changes (its version counter, or a raw-pointer writer's epoch)"""
    hit = _derived.get(key)
    ver = _versions((w,))
    if hit is not None and hit[0]() is w and hit[1] == ver:
        return hit[2]
    out = builder(w)
    _derived[key] = (weakref.ref(w, lambda _r, k=key, c=_derived: c.pop(k, None)), ver, out, builder)
    return out


def _refresh_derived() -> int:
    n = 0
    for key, (ref, ver, _out, builder) in list(_derived.items()):
        w = ref()
        if w is None:
            _derived.pop(key, None)
        elif ver != _versions((w,)):
            if static_addresses():
                # captured HIP graphs hold the ADDRESS of the layout (parallel.StepGraphs): rebuilt in place
                _out.copy_(builder(w))
                _derived[key] = (ref, _versions((w,)), _out, builder)
            else:
                # a FRESH tensor (never in place): whatever a live autograd graph still holds of the old layout stays intact
                _derived[key] = (ref, _versions((w,)), builder(w), builder)
            n += 1
    return n


# Static-address mode (parallel.StepGraphs: the forward / backward launch sequences captured as HIP graphs hold raw
# pointers): every cached copy derived from a parameter -- bf16 operand pairs, plain casts, stacked operands, derived
# layouts -- is then refreshed IN PLACE, always, by refresh_stale_operands(); nothing is ever re-made at a new address.
# The price is the guarantee the eager mode gives a live autograd graph (its saved operand copies stay intact when the
# parameter changes before backward): with graphs on, parameters must not be modified between a forward and its backward
# -- which the graphs' own static activations forbid anyway.
_static = [False]
_static_holders = weakref.WeakSet()     # parallel.StepGraphs objects: the mode is on while any of them holds a captured graph


def set_static_addresses(on: bool) -> bool:
    """explicit switch (beside the automatic one: on while a parallel.StepGraphs holds captured graphs)"""
    prev = _static[0]
    _static[0] = bool(on)
    return prev


def static_addresses() -> bool:
    return _static[0] or any(h.entries for h in _static_holders)


def _refresh_plain_copies() -> int:
    """static-address mode: the cached copies that are NOT (operand, transpose) pairs of the grouped refresh -- plain casts
    (weight_as without padding or of a narrow weight), padded casts without a fused transpose, stacked operands without one,
    weight_t_as transposes, and the lazily made transposes of such copies (_transposed_operand) -- re-made into the buffers
    they already live in.  A handful per model (the head's Linear, narrow 1x1 convolutions)."""
    n = 0
    for key, hit in list(_wcache.items()):
        kind = key[1] if isinstance(key, tuple) and len(key) == 2 else None
        if kind == 'T' or key in _operands:
            continue                                    # transposes follow their source; pairs: the grouped refresh
        refs, ver, out = hit
        ws = [r() for r in (refs if isinstance(refs, tuple) else (refs,))]
        if any(w is None for w in ws) or _versions(ws) == tuple(ver):
            continue
        if kind == 'cat':
            r0 = 0
            for w in ws:
                out[r0:r0 + w.shape[0]].copy_(w.detach())
                r0 += w.shape[0]
        else:
            w2 = ws[0].detach()
            if w2.dim() != 2:
                w2 = w2.reshape(w2.shape[0], -1)
            out.copy_(w2.t() if kind == 't' else w2)    # casts on the way; `out` may be a row-padded view
        tr = _wcache.get((id(out), 'T'))
        if tr is not None and tr[0]() is out:
            tr[2].copy_(out.t())
        _wcache[key] = (refs, _versions(ws), out)
        n += 1
    return n


G256_MIN = 64          # smallest output edge routed to the 256x256 DMA GEMM (mirrors ISTVT_G256_MIN in gemm.hip)


def cast(t: Tensor, dtype: torch.dtype) -> Tensor:
    _req(t)
    if t.dtype == dtype:
        return t
    t = _c(t)
    out = torch.empty(t.shape, dtype=dtype, device=t.device)
    if t.numel():
        _lib.check(_lib.lib().istvt_cast(t.data_ptr(), dtype_code(t), out.data_ptr(), _DT[dtype], t.numel(), _stream()),
                   'istvt_cast')
    return out


def weight_as(w: Tensor, dtype: torch.dtype, pad: bool = False) -> Tensor:
    """2-D view of a (fp32) parameter in the compute dtype; bf16 copies are cached until the parameter is modified
    in place (optimizer step bumps ``_version``).  pad=True: the copy has line-aligned rows (a [N, K] view of a
    [N, pad_ld(K)] buffer), the layout the DMA-staged GEMMs want for their B operand."""
    w2 = w.detach()
    if w2.dim() != 2:
        w2 = w2.reshape(w2.shape[0], -1)
    if w2.dtype == dtype:
        return _c(w2)
    pad = pad and w2.shape[1] % 8 == 0 and pad_ld(w2.shape[1]) != w2.shape[1]
    key = (id(w), 'p') if pad else id(w)    # id-keyed: Tensor.__eq__ is elementwise, so tensors cannot be dict keys
    hit = _wcache.get(key)
    if hit is not None and hit[0]() is w and hit[1] == _versions((w,)) and hit[2].dtype == dtype:
        return hit[2]
    if pad:
        w2 = _c(w2)
        R, C = w2.shape
        out = empty_rows(R, C, dtype, w2.device)
        if w2.dtype == torch.float32 and dtype == torch.bfloat16 and R % 8 == 0 and R >= G256_MIN and C >= G256_MIN:
            # the operand of the input-gradient GEMM (W^T, k-contiguous) comes out of the same pass over the fp32 weight
            wt = empty_rows(C, R, dtype, w2.device)
            _lib.check(_lib.lib().istvt_cast_transpose(w2.data_ptr(), C, out.data_ptr(), out.stride(0), wt.data_ptr(),
                                                       wt.stride(0), R, C, _stream()), 'istvt_cast_transpose')
            tkey = (id(out), 'T')
            _wcache[tkey] = (weakref.ref(out, lambda _r, k=tkey, c=_wcache: c.pop(k, None)), 0, wt)
            _operands[key] = ((weakref.ref(w),), out, wt)          # refreshed in place by refresh_stale_operands()
        else:
            _lib.check(_lib.lib().istvt_cast2d(w2.data_ptr(), dtype_code(w2), C, out.data_ptr(), _DT[dtype],
                                               out.stride(0), R, C, _stream()), 'istvt_cast2d')
    else:
        out = cast(w2, dtype)
    # (the parameter's death drops the operand copies at once, not at the next refresh_stale_operands())
    _wcache[key] = (weakref.ref(w, lambda _r, k=key, c=_wcache, o=_operands: (c.pop(k, None), o.pop(k, None))), _versions((w,)), out)
    return out


def weight_cat_as(ws, dtype: torch.dtype) -> Tensor:
    """[w_0; w_1; ...] stacked along the output dimension as ONE line-aligned GEMM operand in the compute dtype: the
    parameters stay separate (state dict, optimizer), the operand copy is cached until one of them is modified.
    TemporalResidualAttention's [to_qk | to_v] (module.py:182-183): one 728 -> 1536 GEMM instead of two.  For bf16 the
    operand of the input-gradient GEMM ([K, sum N_i], k-contiguous W^T) comes out of the same passes over the fp32
    weights."""
    ws = tuple(ws)
    key = (tuple(id(w) for w in ws), 'cat')
    ver = _versions(ws)
    hit = _wcache.get(key)
    if hit is not None and all(r() is w for r, w in zip(hit[0], ws)) and hit[1] == ver and hit[2].dtype == dtype:
        return hit[2]
    K = ws[0].shape[1]
    if any(w.dim() != 2 or w.shape[1] != K for w in ws):
        raise RuntimeError('weight_cat_as: the weights must be 2-D with one input width')
    R = sum(w.shape[0] for w in ws)
    dev = ws[0].device
    out = empty_rows(R, K, dtype, dev, K % 8 == 0)
    fused_t = (dtype == torch.bfloat16 and all(w.dtype == torch.float32 and w.shape[0] % 8 == 0 and w.shape[0] >= G256_MIN
                                                for w in ws) and K >= G256_MIN and K % 8 == 0)
    wt = empty_rows(K, R, dtype, dev) if fused_t else None
    r0 = 0
    for w in ws:
        w2 = _c(w.detach())
        n = w2.shape[0]
        if fused_t:
            es = out.element_size()
            _lib.check(_lib.lib().istvt_cast_transpose(w2.data_ptr(), K, out.data_ptr() + r0 * out.stride(0) * es, out.stride(0),
                                                       wt.data_ptr() + r0 * es, wt.stride(0), n, K, _stream()),
                       'istvt_cast_transpose')
        else:
            out[r0:r0 + n].copy_(w2)
        r0 += n
    if wt is not None:
        tkey = (id(out), 'T')
        _wcache[tkey] = (weakref.ref(out, lambda _r, k=tkey, c=_wcache: c.pop(k, None)), 0, wt)
        _operands[key] = (tuple(weakref.ref(w) for w in ws), out, wt)
    drop = lambda _r, k=key, c=_wcache, o=_operands: (c.pop(k, None), o.pop(k, None))
    _wcache[key] = (tuple(weakref.ref(w, drop) for w in ws), ver, out)
    return out


# ------------------------------------------------------------------------------------------
# bench.py sets this to a list to time every GEMM launch with events on the launch stream:
# entries are (start_event, end_event, algorithmic_flops, (a_kc, b_kc), (M, N, K)).
gemm_profile = None
# likewise for the memory-bound / attention kernels: entries are (name, start_event, end_event, algorithmic_bytes,
# algorithmic_flops); bench.py turns them into achieved GB/s and TFLOP/s per kernel class.
kernel_profile = None


class prof:
    """with prof('ln_fwd', nbytes): <launch>  -- records events on the launch stream when bench.py asks for it"""

    def __init__(self, name, nbytes=0, flops=0):
        self.rec = (name, nbytes, flops) if kernel_profile is not None else None

    def __enter__(self):
        if self.rec is not None:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e0.record()
        return self

    def __exit__(self, *exc):
        if self.rec is not None and kernel_profile is not None:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            kernel_profile.append((self.rec[0], self.e0, e1, self.rec[1], self.rec[2]))
        return False


_cus: dict = {}


def _cu_count(device) -> int:
    idx = device.index if device.index is not None else torch.cuda.current_device()
    if idx not in _cus:
        n = torch.cuda.get_device_properties(idx).multi_processor_count
        _cus[idx] = (n if n >= 8 else 256) & ~7
    return _cus[idx]


def gemm_kernel_name(A, lda, a_kc, B, ldb, b_kc, C, ldc, M, N, K, epi=0, residual=None, bias=None, out_mode=0,
                     splitk=1, alpha=1.0, stats=0) -> str:
    """which kernel istvt_gemm launches for these operands (mirrors the dispatch rule in csrc/gemm.hip; the names
    are the ones rocprofv3 prints, so bench.py's per-kernel timings can be checked against profiles/)."""
    big = (A.dtype == torch.bfloat16 and bool(a_kc) == bool(b_kc) and M >= G256_MIN and N >= G256_MIN and N % 8 == 0
           and lda % 8 == 0 and ldb % 8 == 0 and ldc % 8 == 0 and (K % 8 == 0 if a_kc else M % 8 == 0))
    if big:
        q_ok = (a_kc and M * lda * 2 < 0x7fffffff and N * ldb * 2 < 0x7fffffff and K >= 32 and out_mode == 0 and splitk == 1
                and (bias is None or alpha == 1.0) and not (epi != 0 and residual is not None))
        if q_ok:
            side = 'true' if (epi == 0 and residual is not None) else 'false'
            # row tile height as gemm.hip picks it: 256 unless ISTVT_GEMM_TM=224 (force) / -1 (rounds x height)
            cus = _cu_count(A.device)
            t256 = -(-M // 256) * -(-N // 256)
            t224 = -(-M // 224) * -(-N // 256)
            tm_env = int(os.environ.get('ISTVT_GEMM_TM', '0'))
            use224 = tm_env == 224 or (tm_env == -1 and -(-t224 // cus) * 224 < -(-t256 // cus) * 256)
            if stats:                   # 1: BatchNorm statistics, 2: column sums in the epilogue (always the 256-row tile)
                use224 = False
            # KHALF: K % 64 in 1..32 -> the last K tile of every output tile runs half its MFMAs (256-row tile only)
            khalf = (not use224) and K > 64 and 0 < (K & 63) <= 32
            return 'gemm256q_kernel<%d, %s, 0, %d, %d, %s>' % (epi, side, 224 if use224 else 256, stats,
                                                               'true' if khalf else 'false')
        t_ok = (not a_kc and out_mode == 3 and bias is None and residual is None and epi == 0 and K * lda * 2 < 0x7fffffff
                and K * ldb * 2 < 0x7fffffff)
        if t_ok:
            return 'gemm256t_kernel'
    t = '__bf16' if A.dtype == torch.bfloat16 else 'float'
    return 'gemm_kernel<%s, %s, %s>' % (t, str(bool(a_kc)).lower(), str(bool(b_kc)).lower())


def gemm_raw(A: Tensor, lda: int, a_kc: bool, B: Tensor, ldb: int, b_kc: bool, C: Tensor, ldc: int, M: int, N: int,
             K: int, *, bias: Optional[Tensor] = None, residual: Optional[Tensor] = None, ldr: int = 0,
             C2: Optional[Tensor] = None, epi: int = 0, out_mode: int = 0, splitk: int = 1, alpha: float = 1.0,
             stats: Optional[Tensor] = None, csum: Optional[Tensor] = None, blocked: bool = True, a_sel_col: int = 0,
             gelu_d: bool = False):
    """stats: double [R][2][N] accumulator (stem.new_stats): the kernel adds the column sums / sums of squares of the
    stored outputs (fused train-mode BatchNorm statistics); only legal where stats_fusable() says so.
    csum=True (with stats, epi 2 only): only the column sums are accumulated (a bias gradient; the caller folds them
    with istvt_stats_reduce_add).
    blocked (float32 only): blocked summation over the reduction dimension (istvt_gemm flags bit 0).  False = one
    sequential fp32 chain, which the Xception stem's forward / input-gradient convolutions keep (it reproduces the
    reference CPU run's ReLU / arg-max decisions).
    a_sel_col > 0 (istvt_gemm flags bit 1): A is two planes of M rows, the second directly behind the first; output
    columns at or past a_sel_col (a multiple of 256) take their rows from the second plane."""
    _req(A); _req(B); _req(C)
    if A.dtype != B.dtype:
        raise TypeError('gemm operands must share a dtype (%s vs %s)' % (A.dtype, B.dtype))
    if bias is not None and bias.dtype != torch.float32:
        raise TypeError('bias must be float32')
    prof = gemm_profile
    if prof is not None:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    rc = _lib.lib().istvt_gemm(A.data_ptr(), lda, int(a_kc), B.data_ptr(), ldb, int(b_kc), C.data_ptr(), ldc, M, N, K,
                               _ptr(bias), _ptr(residual), ldr, _ptr(C2), epi, out_mode, splitk, alpha,
                               stats[0, 0].data_ptr() if stats is not None else None,
                               stats[0, 1].data_ptr() if (stats is not None and csum is None) else None,
                               int(bool(blocked)) | ((2 | ((a_sel_col // 64) << 16)) if a_sel_col else 0) | (16 if gelu_d else 0)
                               | ((_cu_reserve.get(A.device.index, 0) >> 3) << 8 if _cu_reserve else 0),
                               dtype_code(A), _stream())
    if prof is not None:
        ev1.record()
        prof.append((ev0, ev1, 2.0 * M * N * K, (bool(a_kc), bool(b_kc)), (M, N, K),
                     gemm_kernel_name(A, lda, a_kc, B, ldb, b_kc, C, ldc, M, N, K, epi, residual, bias, out_mode,
                                      splitk, alpha, 0 if stats is None else (2 if csum else 1))))
    _lib.check(rc, 'istvt_gemm')


# CUs the persistent NT GEMM launches leave free, per device index: an ARGUMENT of every istvt_gemm call (flags bits
# 8..15), not state of the library.  Python-side dict, read and written under the GIL by the main thread (all_reduce) and
# the autograd thread (the early all-reduce hook).
_cu_reserve: dict = {}


def set_cu_reserve(n: int, device=None) -> int:
    """CUs (a multiple of 8, at most 192) that the persistent NT GEMM launches issued on `device` from now on leave free;
    returns the previous value so the caller can restore it.  parallel.GradBucket raises it while its asynchronous
    all-reduce is in flight.  The value travels with each launch (istvt_gemm flags)."""
    n = int(n)
    if n < 0 or n > 192:
        raise RuntimeError('set_cu_reserve: %d is outside 0..192' % n)
    idx = torch._C._cuda_getDevice() if device is None else (device.index if isinstance(device, torch.device) else int(device))
    old = _cu_reserve.get(idx, 0)
    _cu_reserve[idx] = (n + 7) & ~7
    return old


def get_cu_reserve(device=None) -> int:
    idx = torch._C._cuda_getDevice() if device is None else (device.index if isinstance(device, torch.device) else int(device))
    return _cu_reserve.get(idx, 0)


def stats_fusable(x: Tensor, w: Tensor) -> bool:
    """whether linear_fwd(x, w, stats=...) may be used: the problem runs on the persistent bf16 NT kernel (the only one
    whose epilogue accumulates column statistics)"""
    M, K = x.shape
    N = w.shape[0]
    x2, lda = rows(x)
    w2, ldb = rows(w)
    # (with ISTVT_GEMM_TM=224 forced the C side still runs a statistics launch on the 256-row kernel)
    return (x.dtype == torch.bfloat16
            and gemm_kernel_name(x2, lda, True, w2, ldb, True, None, N, M, N, K).startswith('gemm256q_kernel<0, false'))


def linear_fwd(x: Tensor, w: Tensor, bias: Optional[Tensor] = None, residual: Optional[Tensor] = None,
               gelu: bool = False, pad: bool = False, stats: Optional[Tensor] = None, blocked: bool = True,
               a_sel_col: int = 0, gelu_d: bool = False):
    """y = x @ w.T (+bias) (+residual); with gelu=True returns (u, gelu(u)) -- (gelu'(u), gelu(u)) with gelu_d=True: the
    derivative is all FeedForward's backward needs of u (istvt_gemm flags bit 4; pass the same flag to linear_dgrad).  x [M,K], w [N,K] (x's dtype); both may
    be row-strided views.  pad=True: the outputs are [M, N] views with line-aligned rows.  stats: see gemm_raw."""
    M, K = x.shape
    N = w.shape[0]
    if w.shape[1] != K:
        raise RuntimeError('linear: weight %s does not match input width %d' % (tuple(w.shape), K))
    x, lda = rows(x)
    w, ldb = rows(w)
    y = empty_rows(M, N, x.dtype, x.device, pad)
    ldc = y.stride(0) if M > 1 else N
    if gelu:
        g = empty_rows(M, N, x.dtype, x.device, pad)
        gemm_raw(x, lda, True, w, ldb, True, y, ldc, M, N, K, bias=bias, C2=g, epi=1, blocked=blocked, gelu_d=gelu_d)
        return y, g
    ldr = 0
    if residual is not None:
        residual, ldr = rows(residual)
    gemm_raw(x, lda, True, w, ldb, True, y, ldc, M, N, K, bias=bias, residual=residual, ldr=ldr, stats=stats, blocked=blocked,
             a_sel_col=a_sel_col)
    return y


def weight_t_as(w: Tensor, dtype: torch.dtype) -> Tensor:
    """W^T ([K,N] contiguous) of a [N,K] parameter in the compute dtype, cached per parameter
    version.  With it the input gradient dx = dy W is the same k-contiguous ("NT") GEMM as the
    forward, i.e. it runs on the DMA-staged 256x256 kernel; transposing 88 M weights once per
    optimizer step is noise next to transposing activations."""
    key = (id(w), 't')
    hit = _wcache.get(key)
    if hit is not None and hit[0]() is w and hit[1] == (w._version, _wepoch[0]) and hit[2].dtype == dtype:
        return hit[2]
    wt = weight_as(w, dtype).t()
    out = empty_rows(wt.shape[0], wt.shape[1], dtype, w.device)
    out.copy_(wt)
    _wcache[key] = (weakref.ref(w, lambda _r, k=key, c=_wcache: c.pop(k, None)), (w._version, _wepoch[0]), out)
    return out


def _transposed_operand(w: Tensor) -> Tensor:
    """w^T of an already-cast GEMM operand (the per-version cached bf16 weight copy), cached for
    as long as that operand tensor lives."""
    key = (id(w), 'T')
    hit = _wcache.get(key)
    if hit is not None and hit[0]() is w:
        return hit[2]
    wt = empty_rows(w.shape[1], w.shape[0], w.dtype, w.device)        # line-aligned rows for the DMA-staged GEMM
    wt.copy_(w.t())
    _wcache[key] = (weakref.ref(w, lambda _r, k=key, c=_wcache: c.pop(k, None)), 0, wt)
    return wt


def linear_dgrad(dy: Tensor, w: Tensor, gelu_u: Optional[Tensor] = None, wt: Optional[Tensor] = None,
                 pad: bool = False, csum: Optional[Tensor] = None, blocked: bool = True, gelu_d: bool = False) -> Tensor:
    """dx = dy @ w  (dy [M,N], w [N,K]); with gelu_u: dx *= gelu'(gelu_u) (dx shaped like gelu_u); gelu_d=True: gelu_u IS the
    derivative linear_fwd(gelu=True, gelu_d=True) saved, dx *= gelu_u.
    wt = w^T [K,N] (optional): use the k-contiguous kernel instead of the transposed-operand one.
    csum (with gelu_u): float32 [K] += column sums of dx, taken in the GEMM's epilogue where the kernel supports it
    (else by a colsum pass here): the bias gradient of the Linear whose pre-activation gelu_u is."""
    M, N = dy.shape
    K = w.shape[1]
    dy, lda = rows(dy)
    dx = empty_rows(M, K, dy.dtype, dy.device, pad)
    ldc = dx.stride(0) if M > 1 else K
    epi = 2 if gelu_u is not None else 0
    c2 = None
    if gelu_u is not None:
        c2, ld2 = rows(gelu_u)
        if ld2 != ldc:                    # the kernel reads C2 with C's row stride
            c2 = empty_rows(M, K, dy.dtype, dy.device, pad)
            c2.copy_(gelu_u)
    if wt is None and dy.dtype == torch.bfloat16 and M >= G256_MIN and K >= G256_MIN and N % 8 == 0:
        wt = _transposed_operand(w)
    if wt is not None:
        wt, ldb = rows(wt)
        fuse = (csum is not None and epi == 2 and
                gemm_kernel_name(dy, lda, True, wt, ldb, True, dx, ldc, M, K, N, epi=2).startswith('gemm256q_kernel<2, false'))
        if fuse:
            # replicated double accumulator [R][2][K] (row 0 used): per-tile flushes from 256 workgroups into the 2912
            # addresses of the gradient itself serialise on same-address atomics (+60 us per launch, measured)
            from . import stem as _stem
            acc = _stem.new_stats(K, dy.device)
            gemm_raw(dy, lda, True, wt, ldb, True, dx, ldc, M, K, N, C2=c2, epi=epi, stats=acc, csum=True, gelu_d=gelu_d)
            _lib.check(_lib.lib().istvt_stats_reduce_add(acc.data_ptr(), K, csum.data_ptr(), _stream()), 'istvt_stats_reduce_add')
        else:
            gemm_raw(dy, lda, True, wt, ldb, True, dx, ldc, M, K, N, C2=c2, epi=epi, blocked=blocked, gelu_d=gelu_d)
            if csum is not None:
                colsum(dx, out=csum)
    else:
        w, ldb = rows(w)
        gemm_raw(dy, lda, True, w, ldb, False, dx, ldc, M, K, N, C2=c2, epi=epi, blocked=blocked, gelu_d=gelu_d)
        if csum is not None:
            colsum(dx, out=csum)
    return dx


_WG_TARGET = int(os.environ.get('ISTVT_WGRAD_WGS', '256'))      # workgroups a weight-gradient launch aims for


def _pick_splitk(out_rows: int, out_cols: int, red: int, big_tiles: bool = False) -> int:
    """split of the reduction dim for weight gradients so the grid fills the 256 CUs."""
    if big_tiles:          # 256x256 kernel, one workgroup per CU
        tiles = ((out_rows + 255) // 256) * ((out_cols + 255) // 256)
        s = max(1, _WG_TARGET // tiles)
    else:
        tiles = ((out_rows + 127) // 128) * ((out_cols + 127) // 128)
        s = max(1, 1024 // tiles)
    return min(s, max(1, red // 512))


def linear_wgrad(dy: Tensor, x: Tensor, out: Optional[Tensor] = None) -> Tensor:
    """out[N,K] (+)= dy.T @ x   (dy [M,N], x [M,K]); fp32 accumulate; the reduction dimension is split over workgroups
    that write fp32 partial slabs, summed in split order by one reduce pass (no atomics: bit-reproducible)."""
    M, N = dy.shape
    K = x.shape[1]
    (dy, ldy), (x, ldx) = rows(dy), rows(x)
    if out is None:
        out = torch.zeros((N, K), dtype=torch.float32, device=dy.device)
    big = dy.dtype == torch.bfloat16 and N >= G256_MIN and K >= G256_MIN and N % 8 == 0 and K % 8 == 0
    splits = _pick_splitk(N, K, M, big)
    if big:
        # partial slabs + a reduce pass: every split of a tile finishes at the same moment, so
        # atomics into the same 256x256 tile contend (measured 3-8x slower than this)
        splits = min(splits, (M + 63) // 64)
        kper = -(-(-(-M // splits)) // 64) * 64          # what the C side derives: ceil(ceil(M/s)/64)*64
        splits = -(-M // kper)
        ws = torch.empty((splits, N, K), dtype=torch.float32, device=dy.device)
        gemm_raw(dy, ldy, False, x, ldx, False, ws, K, N, K, M, out_mode=3, splitk=splits)
        _lib.check(_lib.lib().istvt_splitk_reduce(ws.data_ptr(), splits, N * K, out.data_ptr(), _stream()),
                   'istvt_splitk_reduce')
    elif splits > 1:
        bk = 64 if dy.dtype == torch.bfloat16 else 32
        kper = -(-(-(-M // splits)) // bk) * bk            # what the C side derives: ceil(ceil(M/s)/bk)*bk
        splits = -(-M // kper)
        ws = torch.empty((splits, N, K), dtype=torch.float32, device=dy.device)
        gemm_raw(dy, ldy, False, x, ldx, False, ws, K, N, K, M, out_mode=3, splitk=splits)
        if (N * K) % 4 == 0:
            _lib.check(_lib.lib().istvt_splitk_reduce(ws.data_ptr(), splits, N * K, out.data_ptr(), _stream()),
                       'istvt_splitk_reduce')
        else:
            _lib.check(_lib.lib().istvt_rows_reduce(ws.data_ptr(), splits, N * K, out.data_ptr(), _stream()), 'istvt_rows_reduce')
    else:
        gemm_raw(dy, ldy, False, x, ldx, False, out, K, N, K, M, out_mode=2, splitk=1)       # one writer per element
    return out


WGRAD_GROUP_MAX = 8


def wgrad_groupable(dy: Tensor, x: Tensor) -> bool:
    """whether linear_wgrad_group takes this weight gradient (the 256x256 bf16 TN kernel's conditions)"""
    if dy.dtype != torch.bfloat16 or x.dtype != torch.bfloat16 or dy.dim() != 2 or x.dim() != 2:
        return False
    M, N = dy.shape
    K = x.shape[1]
    if N < G256_MIN or K < G256_MIN or N % 8 or K % 8 or x.shape[0] != M:
        return False
    for t in (dy, x):
        if t.stride(1) != 1 or (t.stride(0) * 2) % 16 or t.data_ptr() % 16 or M * t.stride(0) * 2 >= 0x7fffffff:
            return False
    return True


def linear_wgrad_group(items) -> None:
    """items: [(dy [M, N_i], x [M, K_i], out float [N_i, K_i])] with one M, each accepted by wgrad_groupable():
    out_i += dy_i.T @ x_i for all of them in one GEMM launch + one reduce launch (istvt_wgrad_group)."""
    import ctypes as C
    n = len(items)
    if not 1 <= n <= WGRAD_GROUP_MAX:
        raise RuntimeError('linear_wgrad_group: %d problems (1..%d)' % (n, WGRAD_GROUP_MAX))
    M = items[0][0].shape[0]
    Ns, Ks = [], []
    for dy, x, out in items:
        _req(dy); _req(x); _req(out)
        if dy.shape[0] != M or not wgrad_groupable(dy, x):
            raise RuntimeError('linear_wgrad_group: problem not groupable: dy %s x %s' % (tuple(dy.shape), tuple(x.shape)))
        if out.dtype != torch.float32 or not out.is_contiguous() or out.numel() != dy.shape[1] * x.shape[1]:
            raise RuntimeError('linear_wgrad_group: out must be a contiguous float [N, K] buffer')
        Ns.append(dy.shape[1]); Ks.append(x.shape[1])
    PA, LA, IA = C.c_void_p * n, C.c_long * n, C.c_int * n
    a_n, a_k = IA(*Ns), IA(*Ks)
    lib = _lib.lib()
    splits = lib.istvt_wgrad_group_splits(n, a_n, a_k, M)
    if splits < 1:
        _lib.check(splits, 'istvt_wgrad_group_splits')
    elems = sum(a * b for a, b in zip(Ns, Ks))
    ws = torch.empty((splits * elems,), dtype=torch.float32, device=items[0][0].device)
    prof = gemm_profile
    if prof is not None:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    rc = lib.istvt_wgrad_group(n, PA(*[it[0].data_ptr() for it in items]), LA(*[it[0].stride(0) for it in items]),
                               PA(*[it[1].data_ptr() for it in items]), LA(*[it[1].stride(0) for it in items]),
                               PA(*[it[2].data_ptr() for it in items]), a_n, a_k, M, splits, ws.data_ptr(),
                               ws.numel(), _stream())
    if prof is not None:
        ev1.record()
        prof.append((ev0, ev1, 2.0 * M * elems, (False, False), (max(Ns), max(Ks), M), 'gemm256t_group_kernel(GemmGroupArgs)'))
    _lib.check(rc, 'istvt_wgrad_group')


def colsum(x: Tensor, out: Optional[Tensor] = None) -> Tensor:
    M, N = x.shape
    x, ld = rows(_req(x))
    if out is None:
        out = torch.zeros((N,), dtype=torch.float32, device=x.device)
    if N % 8 == 0:
        lib = _lib.lib()
        ws = torch.empty((lib.istvt_colsum_ws_elems(M, N),), dtype=torch.float32, device=x.device)
        with prof('colsum', M * N * x.element_size()):
            _lib.check(lib.istvt_colsum(x.data_ptr(), out.data_ptr(), M, N, ld, ws.data_ptr(), ws.numel(), dtype_code(x),
                                        _stream()), 'istvt_colsum')
    else:
        x = _c(x)
        # narrow outputs (e.g. the 1-logit head): a [N][1] GEMM against ones keeps it on the HIP path
        ones = torch.ones((M, 8), dtype=x.dtype, device=x.device)
        tmp = torch.zeros((N, 8), dtype=torch.float32, device=x.device)
        gemm_raw(x, N, False, ones, 8, False, tmp, 8, N, 8, M, out_mode=2, splitk=1)
        out += tmp[:, 0]
    return out


# ------------------------------------------------------------------------------------------
def layernorm_fwd(x: Tensor, gamma: Tensor, beta: Tensor, eps: float, pad: bool = False):
    """x [..., D] (row-strided views allowed) -> y [M, D] (pad=True: line-aligned rows), mean [M], rstd [M]."""
    x2, ldx = rows(_req(x))
    M, D = x2.shape
    y = empty_rows(M, D, x.dtype, x.device, pad)
    mean = torch.empty((M,), dtype=torch.float32, device=x.device)
    rstd = torch.empty_like(mean)
    with prof('ln_fwd', 2 * M * D * x.element_size()):
        _lib.check(_lib.lib().istvt_layernorm_fwd(x2.data_ptr(), ldx, gamma.data_ptr(), beta.data_ptr(), y.data_ptr(),
                                                  y.stride(0) if M > 1 else D, mean.data_ptr(), rstd.data_ptr(), M, D, eps,
                                                  dtype_code(x), _stream()), 'istvt_layernorm_fwd')
    return (y if x.dim() == 2 else y.view(*x.shape)), mean, rstd


def layernorm_fwd_diff(x: Tensor, gamma: Tensor, beta: Tensor, eps: float, B: int, F: int, P: int):
    """LayerNorm of x [B*F*P, D] (rows (b, f, p)) plus the frame difference of module.py:193 taken in fp32 before the
    rounding to x's dtype.  Returns (y, diff, mean, rstd): y and diff are the two [M, D] planes (line-aligned rows) of
    ONE buffer, diff first -- the layout gemm_raw(a_sel_col=...) takes."""
    x2, ldx = rows(_req(x))
    M, D = x2.shape
    if M != B * F * P:
        raise RuntimeError('layernorm_fwd_diff: %d rows are not B=%d x F=%d x P=%d' % (M, B, F, P))
    ld = pad_ld(D)
    planes = torch.empty((2, M, ld), dtype=x.dtype, device=x.device)
    mean = torch.empty((M,), dtype=torch.float32, device=x.device)
    rstd = torch.empty_like(mean)
    with prof('ln_fwd', 3 * M * D * x.element_size()):
        _lib.check(_lib.lib().istvt_layernorm_fwd_diff(x2.data_ptr(), ldx, gamma.data_ptr(), beta.data_ptr(),
                                                       planes[1].data_ptr(), ld, planes[0].data_ptr(), ld, mean.data_ptr(),
                                                       rstd.data_ptr(), B, F, P, D, eps, dtype_code(x), _stream()),
                   'istvt_layernorm_fwd_diff')
    return planes[1][:, :D], planes[0][:, :D], mean, rstd


def layernorm_bwd(dy: Tensor, x: Tensor, mean: Tensor, rstd: Tensor, gamma: Tensor, dgamma: Tensor, dbeta: Tensor,
                  dres: Optional[Tensor] = None, pad: bool = False, dcol: Optional[Tensor] = None, defer=None) -> Tensor:
    """dx of LayerNorm (+ dres, the gradient arriving through the residual fork); dgamma / dbeta (float32 [D]) accumulate;
    dcol (float32 [D], optional) accumulates the column sums of dx (the producing Linear's bias gradient).  Row-strided
    views allowed.  Reproducible: the column sums are reduced in a fixed order (per-workgroup partial rows + one
    reduce launch), no atomics.
    defer (callable, optional): only the row kernel is launched here; the fold of the partial rows into dgamma / dbeta /
    dcol is handed to defer(ws, M, D, dgamma, dbeta, dcol), which runs layernorm_bwd_reduce(...) with those arguments
    wherever it likes (functional: on the weight-gradient stream) and keeps `ws` alive until that launch has run."""
    x2, ldx = rows(_req(x))
    M, D = x2.shape
    dy, ld_dy = rows(_req(dy))
    ld_res = 0
    if dres is not None:
        dres, ld_res = rows(dres)
    dx = empty_rows(M, D, x.dtype, x.device, pad)
    lib = _lib.lib()
    ws = torch.empty((lib.istvt_layernorm_bwd_ws_elems(M, D),), dtype=torch.float32, device=x.device)
    ntens = 3 + (dres is not None)          # dy, x, dx (+ dres)
    with prof('ln_bwd', ntens * M * D * x.element_size()):
        if defer is None:
            _lib.check(lib.istvt_layernorm_bwd(dy.data_ptr(), ld_dy, x2.data_ptr(), ldx, mean.data_ptr(), rstd.data_ptr(),
                                               gamma.data_ptr(), _ptr(dres), ld_res, dx.data_ptr(), dx.stride(0) if M > 1 else D,
                                               dgamma.data_ptr(), dbeta.data_ptr(), _ptr(dcol), ws.data_ptr(), ws.numel(), M, D,
                                               dtype_code(x), _stream()), 'istvt_layernorm_bwd')
        else:
            _lib.check(lib.istvt_layernorm_bwd_partial(dy.data_ptr(), ld_dy, x2.data_ptr(), ldx, mean.data_ptr(),
                                                       rstd.data_ptr(), gamma.data_ptr(), _ptr(dres), ld_res, dx.data_ptr(),
                                                       dx.stride(0) if M > 1 else D, int(dcol is not None), ws.data_ptr(),
                                                       ws.numel(), M, D, dtype_code(x), _stream()),
                       'istvt_layernorm_bwd_partial')
    if defer is not None:
        defer(ws, M, D, dgamma, dbeta, dcol)
    return dx if x.dim() == 2 else dx.view(*x.shape)


def layernorm_bwd_reduce(ws: Tensor, M: int, D: int, dgamma: Tensor, dbeta: Tensor, dcol: Optional[Tensor] = None) -> None:
    """the second half of layernorm_bwd(defer=...): folds the partial rows in `ws` into dgamma / dbeta (/ dcol), on the
    current stream, in the same fixed order as the one-call form"""
    _lib.check(_lib.lib().istvt_layernorm_bwd_reduce(ws.data_ptr(), ws.numel(), M, D, dgamma.data_ptr(), dbeta.data_ptr(),
                                                     _ptr(dcol), _stream()), 'istvt_layernorm_bwd_reduce')


def _same_rows(like: Tensor, ld: int, M: int, D: int) -> Tensor:
    """an uninitialised [M, D] tensor with row stride ld (the layout of a saved operand: its gradient shares it)"""
    buf = torch.empty((M, ld), dtype=like.dtype, device=like.device)
    return buf if ld == D else buf[:, :D]


def attn_spatial_fwd(qkv: Tensor, BF: int, P: int, heads: int, dh: int, fp8: bool = False):
    """qkv [BF*P, 3*heads*dh] (row-strided views allowed) -> out [BF*P, heads*dh] with line-aligned rows, lse"""
    inner = heads * dh
    qkv, ldq = rows(_req(qkv))
    if tuple(qkv.shape) != (BF * P, 3 * inner):
        raise RuntimeError('attn_spatial: qkv %s is not (%d*%d, 3*%d)' % (tuple(qkv.shape), BF, P, inner))
    out = empty_rows(BF * P, inner, qkv.dtype, qkv.device)
    lse = torch.empty((BF * P, heads, 2), dtype=torch.float32, device=qkv.device)   # (row max [log2], 1/rowsum)
    if fp8 and qkv.dtype != torch.bfloat16:
        raise TypeError('the fp8 attention path needs bfloat16 activations')
    fn = _lib.lib().istvt_attn_spatial_fwd_fp8 if fp8 else _lib.lib().istvt_attn_spatial_fwd
    with prof('attn_spatial_fwd', 4 * BF * P * inner * qkv.element_size(), 4.0 * BF * heads * P * P * dh):
        _lib.check(fn(qkv.data_ptr(), ldq, out.data_ptr(), out.stride(0), lse.data_ptr(), BF, P, heads, dh, dh ** -0.5,
                      dtype_code(qkv), _stream()), 'istvt_attn_spatial_fwd')
    return out, lse


def attn_spatial_bwd(qkv: Tensor, out: Tensor, dout: Tensor, lse: Tensor, BF: int, P: int, heads: int, dh: int,
                     fp8: bool = False) -> Tensor:
    inner = heads * dh
    qkv, ldq = rows(_req(qkv))
    out, ldo = rows(_req(out))
    dout, ldd = rows(_req(dout))
    if ldd != ldo:                       # the kernels take one stride for out and dout
        d2 = _same_rows(out, ldo, BF * P, inner)
        d2.copy_(dout)
        dout = d2
    dqkv = _same_rows(qkv, ldq, BF * P, 3 * inner)
    delta = torch.empty((BF * P, heads), dtype=torch.float32, device=qkv.device)
    fn = _lib.lib().istvt_attn_spatial_bwd_fp8 if fp8 else _lib.lib().istvt_attn_spatial_bwd
    with prof('attn_spatial_bwd', 8 * BF * P * inner * qkv.element_size(), 10.0 * BF * heads * P * P * dh):
        _lib.check(fn(qkv.data_ptr(), ldq, out.data_ptr(), dout.data_ptr(), ldo, lse.data_ptr(), delta.data_ptr(),
                      dqkv.data_ptr(), BF, P, heads, dh, dh ** -0.5, dtype_code(qkv), _stream()), 'istvt_attn_spatial_bwd')
    return dqkv


def attn_temporal_fwd(qk: Tensor, v: Tensor, B: int, F: int, P: int, heads: int, dh: int, diff: bool = False):
    """qk [B*F*P, 2*heads*dh], v [B*F*P, heads*dh] (row-strided views allowed: e.g. the two column ranges of one packed
    q|k|v projection) -> out [B*F*P, heads*dh] with line-aligned rows.  diff = 1: frame difference on q, k in the kernel
    (TemporalResidualAttention, module.py:193); diff = 2 (bfloat16): q, k arrive differenced (layernorm_fwd_diff), the
    backward still returns gradients w.r.t. the un-differenced projections."""
    inner = heads * dh
    (qk, ldqk), (v, ldv) = rows(_req(qk)), rows(_req(v))
    if F > 17:
        raise RuntimeError('attn_temporal: at most 17 frames (T <= 16) are supported, got F=%d' % F)
    if tuple(qk.shape) != (B * F * P, 2 * inner) or tuple(v.shape) != (B * F * P, inner):
        raise RuntimeError('attn_temporal: shapes %s / %s do not match B=%d F=%d P=%d' % (tuple(qk.shape), tuple(v.shape), B, F, P))
    out = empty_rows(B * F * P, inner, v.dtype, v.device)
    ldo = out.stride(0)
    with prof('attn_temporal_fwd', 4 * B * F * P * inner * qk.element_size(), 4.0 * B * P * heads * F * F * dh):
        _lib.check(_lib.lib().istvt_attn_temporal_fwd(qk.data_ptr(), ldqk, v.data_ptr(), ldv, out.data_ptr(), ldo, B, F, P,
                                                      heads, dh, dh ** -0.5, int(diff), dtype_code(qk), _stream()),
                   'istvt_attn_temporal_fwd')
    return out


def attn_temporal_bwd(qk: Tensor, v: Tensor, dout: Tensor, B: int, F: int, P: int, heads: int, dh: int, diff: bool = False,
                      packed: bool = False):
    """-> (dqk, dv) laid out like qk and v.  packed=True (qk and v are the column ranges [0, 2*inner) and [2*inner,
    3*inner) of one projection buffer): the two gradients are the same column ranges of ONE [B*F*P, 3*inner] buffer,
    returned as (dqkv, None) -- the operand of a single input-gradient GEMM."""
    inner = heads * dh
    (qk, ldqk), (v, ldv) = rows(_req(qk)), rows(_req(v))
    dout, ldo = rows(_req(dout))
    M = B * F * P
    if packed:
        if ldqk != ldv or v.data_ptr() != qk.data_ptr() + 2 * inner * qk.element_size():
            raise RuntimeError('attn_temporal_bwd(packed=True): v is not the third column range of the qk buffer')
        dqkv = _same_rows(qk, ldqk, M, 3 * inner)
        dqk, dv = dqkv[:, :2 * inner], dqkv[:, 2 * inner:]
    else:
        dqkv = None
        dqk = _same_rows(qk, ldqk, M, 2 * inner)
        dv = _same_rows(v, ldv, M, inner)
    with prof('attn_temporal_bwd', 7 * M * inner * qk.element_size(), 10.0 * B * P * heads * F * F * dh):
        _lib.check(_lib.lib().istvt_attn_temporal_bwd(qk.data_ptr(), ldqk, v.data_ptr(), ldv, dout.data_ptr(), ldo,
                                                      dqk.data_ptr(), dv.data_ptr(), B, F, P, heads, dh, dh ** -0.5,
                                                      int(diff), dtype_code(qk), _stream()), 'istvt_attn_temporal_bwd')
    return (dqkv, None) if packed else (dqk, dv)


# ------------------------------------------------------------------------------------------
def tokens_fwd(feats: Tensor, space: Tensor, temporal: Tensor, pos: Tensor, pad: bool = False) -> Tensor:
    """feats [B,T,hw,D] -> x [B,(T+1)*(hw+1),D]; pos is the full (1,T,P_decl,D) parameter."""
    feats = _c(_req(feats))
    B, T, hw, D = feats.shape
    F, P = T + 1, hw + 1
    if pos.shape[1] != T:
        raise RuntimeError('The size of tensor a (%d) must match the size of tensor b (%d) at non-singleton dimension 1'
                           % (T, pos.shape[1]))          # same failure the reference hits at vivit.py:138
    if pos.shape[2] < P:
        raise RuntimeError('pos_embedding has %d tokens per frame, input needs %d' % (pos.shape[2], P))
    x = empty_rows(B * F * P, D, feats.dtype, feats.device, pad)
    with prof('tokens_fwd', (B * T * hw + B * F * P) * D * feats.element_size()):
        _lib.check(_lib.lib().istvt_tokens_fwd(feats.data_ptr(), space.data_ptr(), temporal.data_ptr(), pos.data_ptr(),
                                               x.data_ptr(), x.stride(0), B, F, P, D, pos.shape[2], dtype_code(feats),
                                               _stream()), 'istvt_tokens_fwd')
    return x.view(B, F * P, D)


def tokens_bwd(dx: Tensor, B: int, T: int, hw: int, D: int, dspace: Tensor, dtemporal: Tensor, dpos: Tensor,
               need_dfeats: bool) -> Optional[Tensor]:
    dx, lddx = rows(_req(dx))
    F, P = T + 1, hw + 1
    dfeats = torch.empty((B, T, hw, D), dtype=dx.dtype, device=dx.device) if need_dfeats else None
    ws = torch.empty(((P + F - 1) * D,), dtype=torch.float32, device=dx.device)       # partial rows of the two token gradients
    with prof('tokens_bwd', (B * F * P + (B * T * hw if need_dfeats else 0)) * D * dx.element_size()):
        _lib.check(_lib.lib().istvt_tokens_bwd(dx.data_ptr(), lddx, _ptr(dfeats), dspace.data_ptr(), dtemporal.data_ptr(),
                                               dpos.data_ptr(), ws.data_ptr(), B, F, P, D, dpos.shape[2], dtype_code(dx),
                                               _stream()), 'istvt_tokens_bwd')
    return dfeats


def frame_diff(x: Tensor, B: int, F: int, P: int, adjoint: bool = False) -> Tensor:
    x = _c(_req(x))
    D = x.shape[-1]
    if x.numel() != B * F * P * D:
        raise RuntimeError('frame_diff: %s is not (B=%d, F=%d, P=%d, D)' % (tuple(x.shape), B, F, P))
    out = torch.empty_like(x)
    _lib.check(_lib.lib().istvt_frame_diff(x.data_ptr(), out.data_ptr(), B, F, P, D, int(adjoint), dtype_code(x),
                                           _stream()), 'istvt_frame_diff')
    return out
