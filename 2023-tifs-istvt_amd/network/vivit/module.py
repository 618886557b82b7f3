"""Transformer sub-modules of ISTVT with the reference's constructor/forward signatures and
parameter names (reference: network/vivit/module.py), executing on hand-written HIP kernels.

The four modules on the hot path: PreNorm (module.py:15-21), FeedForward (:23-34), SpatialOnlyAttention
(:66-93), TemporalResidualAttention (:174-208); and the ablation siblings: Attention (:36-64) and
TemporalOnlyAttention (:145-172), which are the same two attention kernels under other index maps (one "frame" of
all n tokens; one packed to_qkv and no frame difference).  The tokens-per-frame count the reference hard-codes as
``19 * 19 + 1`` is the ``hw`` keyword (default 362).

All forwards take ``(b, n, dim)`` float32 or bfloat16 tensors on a ROCm device.  Extra keyword
arguments (``hw=``, ``residual=``) travel through PreNorm's ``**kwargs`` exactly like the
reference forwards them; ``residual`` lets the caller fuse ``fn(x) + residual`` into the output
projection's epilogue.
"""
import torch
from torch import nn

from istvt_amd import functional as Fn
from istvt_amd import ops


def _frames(n, hw, what):
    if hw <= 0 or n % hw != 0:
        # the reference fails here inside einops.rearrange with an EinopsError
        raise RuntimeError("%s: cannot split %d tokens into frames of hw=%d tokens "
                           "(Shape mismatch, can't divide axis of length %d in chunks of %d)" % (what, n, hw, n, hw))
    return n // hw


def _dropout(seq, y):
    """the nn.Dropout that follows an output projection (module.py:78,187), on the HIP dropout kernel"""
    return Fn.dropout(y, seq[1].p, seq[1].training)


class PreNorm(nn.Module):
    def __init__(self, dim, fn):
        super().__init__()
        self.norm = nn.LayerNorm(dim)
        self.fn = fn

    def forward(self, x, **kwargs):
        # Two extensions for STTransformer (both default off, the reference semantics are unchanged):
        #   fork=True        -> returns (fn(norm(x)), x'): x' is x routed through the LayerNorm's autograd node, so
        #                       a caller that adds x as a residual LATER gets that gradient summed inside the
        #                       LayerNorm backward kernel instead of a separate add pass;
        #   residual='input' -> fn(norm(x), residual=x) with the same routing.
        #   sink=BiasSink(b) -> x is the output of the nn.Linear with bias b (called with defer_bias=True): its bias
        #                       gradient, the column sum of this LayerNorm's input gradient, is accumulated by the
        #                       LayerNorm backward kernel instead of a separate pass.
        fork = kwargs.pop('fork', False)
        sink = kwargs.pop('sink', None)
        own_res = isinstance(kwargs.get('residual'), str) and kwargs['residual'] == 'input'
        geom = self.fn.frame_diff_geometry(x, kwargs.get('hw')) if hasattr(self.fn, 'frame_diff_geometry') else None
        if geom is not None:
            # TemporalResidualAttention in bfloat16: the LayerNorm kernel also writes the frame difference (module.py:193),
            # taken in fp32 before the rounding, and the attention projects q | k from it
            outs = Fn.layer_norm_diff(x, self.norm.weight, self.norm.bias, self.norm.eps, geom, fork=fork or own_res, sink=sink)
            if own_res:
                kwargs['residual'] = outs[2]
            out = self.fn(outs[0], x_diff=outs[1], **kwargs)
            return (out, outs[2]) if fork else out
        if fork or own_res:
            y, xr = Fn.layer_norm(x, self.norm.weight, self.norm.bias, self.norm.eps, fork=True, sink=sink)
            if own_res:
                kwargs['residual'] = xr
            out = self.fn(y, **kwargs)
            return (out, xr) if fork else out
        return self.fn(Fn.layer_norm(x, self.norm.weight, self.norm.bias, self.norm.eps, sink=sink), **kwargs)


class FeedForward(nn.Module):
    def __init__(self, dim, hidden_dim, dropout=0.):
        super().__init__()
        self.net = nn.Sequential(
            nn.Linear(dim, hidden_dim),
            nn.GELU(),
            nn.Dropout(dropout),
            nn.Linear(hidden_dim, dim),
            nn.Dropout(dropout)
        )

    def forward(self, x, residual=None, defer_bias=False):
        fc1, fc2 = self.net[0], self.net[3]
        lead = x.shape[:-1]
        x2 = x.reshape(-1, x.shape[-1])
        r2 = residual.reshape(-1, fc2.out_features) if residual is not None else None
        if self.training and (self.net[2].p > 0 or self.net[4].p > 0):
            # dropout > 0 (reference default is 0): the unfused chain Linear+GELU -> Dropout -> Linear -> Dropout (+ residual)
            y = Fn.FeedForwardDropFn.apply(x2, fc1.weight, fc1.bias, fc2.weight, fc2.bias, self.net[2].p, self.net[4].p,
                                           Fn._new_seed(), Fn._new_seed())
            if r2 is not None:
                y = Fn.add(y, r2)
            return y.view(*lead, fc2.out_features)
        y = Fn.FeedForwardFn.apply(x2, fc1.weight, fc1.bias, fc2.weight, fc2.bias, r2, defer_bias)
        return y.view(*lead, fc2.out_features)


class SpatialOnlyAttention(nn.Module):
    def __init__(self, dim, heads=8, dim_head=64, dropout=0., hw=19 * 19 + 1):
        super().__init__()
        inner_dim = dim_head * heads
        self.heads = heads
        self.dim_head = dim_head
        self.scale = dim_head ** -0.5
        self.hw = hw
        self.attn_fp8 = False           # fp8 (e4m3) operands in the attention MFMAs (bfloat16 activations only)
        self.to_qkv = nn.Linear(dim, inner_dim * 3, bias=False)
        self.to_out = nn.Sequential(
            nn.Linear(inner_dim, dim),
            nn.Dropout(dropout)
        )

    def forward(self, x, hw=None, residual=None, defer_bias=False):
        b, n, _ = x.shape
        hw = self.hw if hw is None else hw
        frames = _frames(n, hw, 'SpatialOnlyAttention')
        x2 = x.reshape(b * n, -1)
        qkv = Fn.LinearFn.apply(x2, self.to_qkv.weight, None, None)
        out = Fn.SpatialAttnFn.apply(qkv, b * frames, hw, self.heads, self.dim_head, self.attn_fp8)
        proj = self.to_out[0]
        plain = self.to_out[1].p == 0.0 or not self.training
        r2 = residual.reshape(b * n, -1) if (residual is not None and plain) else None
        y = Fn.LinearFn.apply(out, proj.weight, proj.bias, r2, defer_bias and plain).view(b, n, -1)
        if not plain:
            y = _dropout(self.to_out, y)
            if residual is not None:
                y = Fn.add(y, residual)
        return y


class TemporalResidualAttention(nn.Module):
    def __init__(self, dim, heads=8, dim_head=64, dropout=0., hw=19 * 19 + 1):
        super().__init__()
        inner_dim = dim_head * heads
        self.heads = heads
        self.dim_head = dim_head
        self.scale = dim_head ** -0.5
        self.hw = hw
        self.to_qk = nn.Linear(dim, inner_dim * 2, bias=False)
        self.to_v = nn.Linear(dim, inner_dim, bias=False)
        self.to_out = nn.Sequential(
            nn.Linear(inner_dim, dim),
            nn.Dropout(dropout)
        )

    def frame_diff_geometry(self, x, hw=None):
        """(B, F, P) when the enclosing PreNorm should produce the frame difference beside the LayerNorm output
        (bfloat16 on the MFMA temporal kernels with q | k ending on a GEMM column tile), else None"""
        hw = self.hw if hw is None else hw
        inner = self.heads * self.dim_head
        if x.dtype != torch.bfloat16 or x.dim() != 3 or hw <= 0 or x.shape[1] % hw or (2 * inner) % 256:
            return None
        frames = x.shape[1] // hw
        if frames > 32 or self.dim_head not in (32, 64) or x.shape[0] * x.shape[1] < 256:
            return None
        # the kernel-side preconditions of this path, mirrored (csrc/gemm.hip q_ok + a_sel): the two-plane A operand of
        # the persistent NT GEMM is addressed with 32-bit byte offsets (both planes below 2 GiB), K is a multiple of 8 and
        # at least 32; otherwise the caller falls back to the one-plane path (diff = 1), which every kernel takes
        rows, d = x.shape[0] * x.shape[1], x.shape[2]
        if d % 8 or d < 32 or 2 * rows * ops.pad_ld(d) * 2 >= 0x7fffffff or 3 * inner * ops.pad_ld(d) * 2 >= 0x7fffffff:
            return None
        return (x.shape[0], frames, hw)

    def forward(self, x, hw=None, residual=None, defer_bias=False, x_diff=None):
        # module.py:192-206.  The reference differences the LayerNorm output over frames (:193), projects the difference
        # with to_qk and the un-differenced rows with to_v.  to_qk has no bias (:182), so to_qk(x[f] - x[f-1]) =
        # to_qk(x[f]) - to_qk(x[f-1]): ONE GEMM on the stacked [to_qk | to_v] operand either way.
        #   float32: the operand is x and the temporal attention kernels difference q and k in registers (all frames of a
        #            position sit in one wavefront) -- exact, nothing is rounded in between;
        #   bfloat16 (x_diff given, from PreNorm): the operand is x_diff for the q | k column tiles and x for the v tiles
        #            (the GEMM picks the plane per column tile): q', k' are rounded at the magnitude of the DIFFERENCE, as
        #            in the reference's order.  (Rounds 2-3 differenced bf16 q, k in the kernels: for correlated
        #            consecutive frames that loses |q| / |q'| in relative precision.)
        b, n, d = x.shape
        hw = self.hw if hw is None else hw
        frames = _frames(n, hw, 'TemporalResidualAttention')
        if x_diff is not None:
            qkv = Fn.LinearCatSelFn.apply(x.reshape(b * n, d), x_diff.reshape(b * n, d), 2 * self.heads * self.dim_head,
                                          self.to_qk.weight, self.to_v.weight)
            out = Fn.TemporalAttnFn.apply(qkv, b, frames, hw, self.heads, self.dim_head, 2)
        else:
            qkv = Fn.LinearCatFn.apply(x.reshape(b * n, d), self.to_qk.weight, self.to_v.weight)
            out = Fn.TemporalAttnFn.apply(qkv, b, frames, hw, self.heads, self.dim_head, 1)
        proj = self.to_out[0]
        plain = self.to_out[1].p == 0.0 or not self.training
        r2 = residual.reshape(b * n, -1) if (residual is not None and plain) else None
        y = Fn.LinearFn.apply(out, proj.weight, proj.bias, r2, defer_bias and plain).view(b, n, -1)
        if not plain:
            y = _dropout(self.to_out, y)
            if residual is not None:
                y = Fn.add(y, residual)
        return y


class Attention(nn.Module):
    """Plain multi-head self-attention over all n tokens of each sequence (reference module.py:36-64): the spatial
    attention kernel with one "frame" per sequence."""

    def __init__(self, dim, heads=8, dim_head=64, dropout=0.):
        super().__init__()
        inner_dim = dim_head * heads
        project_out = not (heads == 1 and dim_head == dim)
        self.heads = heads
        self.dim_head = dim_head
        self.scale = dim_head ** -0.5
        self.to_qkv = nn.Linear(dim, inner_dim * 3, bias=False)
        self.to_out = nn.Sequential(
            nn.Linear(inner_dim, dim),
            nn.Dropout(dropout)
        ) if project_out else nn.Identity()

    def forward(self, x, residual=None):
        b, n, _ = x.shape
        qkv = Fn.LinearFn.apply(x.reshape(b * n, -1), self.to_qkv.weight, None, None)
        out = Fn.SpatialAttnFn.apply(qkv, b, n, self.heads, self.dim_head, False)
        if isinstance(self.to_out, nn.Identity):
            y = out.view(b, n, -1)
            return y if residual is None else Fn.add(y, residual)
        proj = self.to_out[0]
        plain = self.to_out[1].p == 0.0 or not self.training
        r2 = residual.reshape(b * n, -1) if (residual is not None and plain) else None
        y = Fn.LinearFn.apply(out, proj.weight, proj.bias, r2).view(b, n, -1)
        if not plain:
            y = _dropout(self.to_out, y)
            if residual is not None:
                y = Fn.add(y, residual)
        return y


class TemporalOnlyAttention(nn.Module):
    """Attention over the frame axis per (batch, head, position) from ONE packed to_qkv, no frame difference
    (reference module.py:145-172; its hard-coded ``hw = 19 * 19 + 1`` is the ``hw`` keyword).  The temporal
    attention kernel takes q|k and v as column ranges of the one packed projection."""

    def __init__(self, dim, heads=8, dim_head=64, dropout=0., hw=19 * 19 + 1):
        super().__init__()
        inner_dim = dim_head * heads
        self.heads = heads
        self.dim_head = dim_head
        self.scale = dim_head ** -0.5
        self.hw = hw
        self.to_qkv = nn.Linear(dim, inner_dim * 3, bias=False)
        self.to_out = nn.Sequential(
            nn.Linear(inner_dim, dim),
            nn.Dropout(dropout)
        )

    def forward(self, x, hw=None, residual=None):
        b, n, d = x.shape
        hw = self.hw if hw is None else hw
        frames = _frames(n, hw, 'TemporalOnlyAttention')
        inner = self.heads * self.dim_head
        x2 = x.reshape(b * n, d)
        qkv = Fn.LinearFn.apply(x2, self.to_qkv.weight, None, None)
        out = Fn.TemporalAttnFn.apply(qkv, b, frames, hw, self.heads, self.dim_head, False)
        proj = self.to_out[0]
        plain = self.to_out[1].p == 0.0 or not self.training
        r2 = residual.reshape(b * n, -1) if (residual is not None and plain) else None
        y = Fn.LinearFn.apply(out, proj.weight, proj.bias, r2).view(b, n, -1)
        if not plain:
            y = _dropout(self.to_out, y)
            if residual is not None:
                y = Fn.add(y, residual)
        return y


for _cls in (PreNorm, FeedForward, SpatialOnlyAttention, TemporalResidualAttention, Attention, TemporalOnlyAttention):
    _cls._replicate_for_data_parallel = Fn.no_data_parallel      # nn.DataParallel: see functional.no_data_parallel
