"""ctypes binding of libistvt_hip.so (C ABI declared in include/istvt_hip.h).

There is no fallback: if the library is missing or a symbol is absent, importing the ops
raises.  Build with ``python 2023-tifs-istvt_amd/build.py`` (or ``__graft_entry__.build()``).
"""
import ctypes
import os
from ctypes import c_float, c_int, c_long, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# ISTVT_LIB: another build of the same library (same-box A/B of a kernel change); it must export the same entry points
LIB_PATH = os.environ.get('ISTVT_LIB') or os.path.join(_HERE, 'libistvt_hip.so')

P, I, L, F = c_void_p, c_int, c_long, c_float

# name -> argtypes (all return int).  Mirrors include/istvt_hip.h one to one.
SIGNATURES = {
    'istvt_gemm': [P, L, I, P, L, I, P, L, I, I, I, P, P, L, P, I, I, I, F, P, P, I, I, P],
    'istvt_layernorm_fwd': [P, L, P, P, P, L, P, P, L, I, F, I, P],
    'istvt_layernorm_fwd_diff': [P, L, P, P, P, L, P, L, P, P, I, I, I, I, F, I, P],
    'istvt_layernorm_bwd': [P, L, P, L, P, P, P, P, L, P, L, P, P, P, P, L, L, I, I, P],
    'istvt_layernorm_bwd_ws_elems': [L, I],
    'istvt_layernorm_bwd_partial': [P, L, P, L, P, P, P, P, L, P, L, I, P, L, L, I, I, P],
    'istvt_layernorm_bwd_reduce': [P, L, L, I, P, P, P, P],
    'istvt_attn_spatial_fwd': [P, L, P, L, P, I, I, I, I, F, I, P],
    'istvt_attn_spatial_bwd': [P, L, P, P, L, P, P, P, I, I, I, I, F, I, P],
    'istvt_attn_spatial_fwd_fp8': [P, L, P, L, P, I, I, I, I, F, I, P],
    'istvt_attn_spatial_bwd_fp8': [P, L, P, P, L, P, P, P, I, I, I, I, F, I, P],
    'istvt_attn_temporal_fwd': [P, L, P, L, P, L, I, I, I, I, I, F, I, I, P],
    'istvt_attn_temporal_bwd': [P, L, P, L, P, L, P, P, I, I, I, I, I, F, I, I, P],
    'istvt_tokens_fwd': [P, P, P, P, P, L, I, I, I, I, I, I, P],
    'istvt_tokens_bwd': [P, L, P, P, P, P, P, I, I, I, I, I, I, P],
    'istvt_frame_diff': [P, P, I, I, I, I, I, I, P],
    'istvt_stats_replicas': [],
    'istvt_stats_reduce': [P, I, P],
    'istvt_stats_reduce_add': [P, I, P, P],
    'istvt_bn_stats': [P, P, P, L, I, I, P],
    'istvt_bn_finalize': [P, P, ctypes.c_double, P, P, P, P, F, F, P, I, I, I, P],
    'istvt_bn_apply': [P, P, P, L, I, I, I, P],
    'istvt_bn_bwd_stats': [P, P, P, P, P, L, I, I, P],
    'istvt_bn_bwd_apply': [P, P, P, P, P, P, P, P, P, L, I, I, I, P],
    'istvt_bn_add_fwd': [P, P, P, P, P, L, I, I, P],
    'istvt_im2col_conv1': [P, P, I, I, I, P],
    'istvt_conv1_fwd': [P, P, P, I, I, I, P],
    'istvt_conv1_wgrad': [P, P, P, P, I, I, I, P],
    'istvt_conv1_wgrad_slabs': [],
    'istvt_conv2_fwd': [P, P, P, P, I, I, I, P],
    'istvt_conv2_dgrad': [P, P, P, P, P, I, I, I, P],
    'istvt_conv2_wgrad': [P, P, P, P, P, I, I, I, P],
    'istvt_conv2_wgrad_slabs': [],
    'istvt_col2im_conv1': [P, P, I, I, I, P],
    'istvt_im2col3x3': [P, P, I, P, I, I, I, I, I, P],
    'istvt_col2im3x3': [P, P, P, P, I, I, I, I, I, P],
    'istvt_dwconv3x3': [P, P, P, I, I, I, I, P, I, I, P, P, I, I, P, I, I, P, P, I, P],
    'istvt_dwconv3x3_wgrad': [P, P, I, P, P, P, L, I, I, I, I, I, P],
    'istvt_dwconv3x3_wgrad_ws_elems': [I, I, I, I],
    'istvt_pool_add_fwd': [P, P, P, P, P, P, I, I, I, I, I, P],
    'istvt_pool_bwd': [P, P, P, I, I, I, I, P, P, P, P, I, P],
    'istvt_subsample2': [P, P, I, I, I, I, I, P],
    'istvt_splitk_reduce': [P, I, L, P, P],
    'istvt_wgrad_group': [I, P, P, P, P, P, P, P, I, I, P, L, P],
    'istvt_wgrad_group_splits': [I, P, P, I],
    'istvt_colsum': [P, P, L, I, L, P, L, I, P],
    'istvt_colsum_ws_elems': [L, I],
    'istvt_rows_reduce': [P, I, L, P, P],
    'istvt_cast': [P, I, P, I, L, P],
    'istvt_cast2d': [P, I, L, P, I, L, L, I, P],
    'istvt_cast_transpose': [P, L, P, L, P, L, I, I, P],
    'istvt_cast_transpose_group': [I, P, P, P, P, P, P, P, P, P],
    'istvt_relu_avgpool_fwd': [P, P, I, I, I, I, I, P],
    'istvt_relu_avgpool_bwd': [P, P, P, I, I, I, I, I, P],
    'istvt_prepend_fwd': [P, P, P, P, L, L, I, I, I, I, I, P],
    'istvt_prepend_bwd': [P, L, P, P, P, P, L, I, I, I, I, I, P],
    'istvt_prepend_bwd_ws_rows': [L, I, I],
    'istvt_seq_mean_fwd': [P, L, P, L, I, I, I, P],
    'istvt_seq_mean_bwd': [P, P, L, L, I, I, I, P],
    'istvt_dropout_fwd': [P, L, P, L, P, L, I, F, ctypes.c_ulonglong, I, P],
    'istvt_dropout_bwd': [P, L, P, P, L, L, I, F, I, P],
    'istvt_add': [P, L, P, L, P, L, L, I, I, P],
    'istvt_sgd_momentum': [P, P, P, L, F, F, F, F, I, I, I, F, P],
    'istvt_adamw': [P, P, P, P, L, F, F, F, F, F, L, I, F, P],
}

_lib = None


class IstvtLibraryError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise IstvtLibraryError(
                'libistvt_hip.so not found at %s: build it with `python 2023-tifs-istvt_amd/build.py`; '
                'there is no CPU or PyTorch fallback for the ISTVT hot path' % LIB_PATH)
        l = ctypes.CDLL(LIB_PATH)
        for name, argtypes in SIGNATURES.items():
            try:
                fn = getattr(l, name)
            except AttributeError as e:
                raise IstvtLibraryError('libistvt_hip.so lacks symbol %s (stale build?)' % name) from e
            fn.argtypes = argtypes
            fn.restype = c_int
        _lib = l
    return _lib


def check(rc: int, what: str):
    if rc == 0:
        return
    if rc == -2:
        msg = 'unsupported dtype'
    elif rc == -3:
        msg = 'invalid shape/argument'
    elif rc <= -1000:
        msg = 'hipError_t %d at launch' % (-rc - 1000)
    else:
        msg = 'error %d' % rc
    raise RuntimeError('%s failed: %s' % (what, msg))
