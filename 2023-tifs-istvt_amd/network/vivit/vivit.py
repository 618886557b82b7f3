"""ISTVT model classes with the reference's signatures (reference: network/vivit/vivit.py):
STTransformer (:85-101), DSTTr (:103-148), XceptionVidTr (:193-208).

Geometry the reference hard-codes is exposed as keyword arguments whose defaults reproduce it:
``XceptionVidTr()`` == 6 frames, 19x19 grid (300^2 input), depth 12.  ``compute_dtype``
selects float32 (parity mode, default) or bfloat16 (throughput mode) activation storage;
parameters stay float32 so reference checkpoints load unchanged (state-dict names identical).
"""
import os

import torch
from torch import nn

from istvt_amd import functional as Fn
from istvt_amd import ops
from .module import (Attention, PreNorm, FeedForward, SpatialOnlyAttention, TemporalOnlyAttention,  # noqa: F401
                     TemporalResidualAttention)


class Transformer(nn.Module):
    """Reference vivit.py:10-25 (the ablation baselines' encoder): x = attn(LN(x)) + x ; x = ff(LN(x)) + x ; LN."""

    def __init__(self, dim, depth, heads, dim_head, mlp_dim, dropout=0.):
        super().__init__()
        self.layers = nn.ModuleList([])
        self.norm = nn.LayerNorm(dim)
        for _ in range(depth):
            self.layers.append(nn.ModuleList([
                PreNorm(dim, Attention(dim, heads=heads, dim_head=dim_head, dropout=dropout)),
                PreNorm(dim, FeedForward(dim, mlp_dim, dropout=dropout))
            ]))

    def forward(self, x):
        ops.refresh_stale_operands()
        # both residual adds ride in the epilogue of the block's last GEMM; the residual's gradient is summed inside
        # the LayerNorm backward kernel (PreNorm residual='input')
        for attn, ff in self.layers:
            x = attn(x, residual='input')
            x = ff(x, residual='input')
        return Fn.layer_norm(x, self.norm.weight, self.norm.bias, self.norm.eps)


class STTransformer(nn.Module):
    def __init__(self, dim, depth, heads, dim_head, mlp_dim, dropout=0.):
        super().__init__()
        # Opt-in (off by default, so the default model executes every reference operation): DSTTr reads ONE row per clip
        # of the last layer's output -- (frame 0, token 0), vivit.py:144-146.  Spatial attention mixes tokens only inside
        # a frame and the feed-forward mixes nothing, so in the LAST layer only frame 0 (197 of 1773 rows per clip at C2)
        # has to go through the spatial block and only the 32 class rows through its output residual and the feed-forward;
        # every other row's output is dead and its gradient exactly zero.  With the flag on those rows are skipped:
        # identical logits and gradients (tests/test_model_gpu.py), ~4 % fewer FLOPs per step.
        self.dead_row_elimination = False
        self.layers = nn.ModuleList([])
        self.norm = nn.LayerNorm(dim)
        for _ in range(depth):
            self.layers.append(nn.ModuleList([
                PreNorm(dim, TemporalResidualAttention(dim, heads=heads, dim_head=dim_head, dropout=dropout)),
                PreNorm(dim, SpatialOnlyAttention(dim, heads=heads, dim_head=dim_head, dropout=dropout)),
                PreNorm(dim, FeedForward(dim, mlp_dim, dropout=dropout))
            ]))

    def forward(self, x, hw=None, cls_of=None):
        ops.refresh_stale_operands()     # every bf16 weight operand the optimizer step invalidated: ONE grouped cast launch
        # x = attn_s(attn_t(x)) + x ; x = ff(x) + x   (ONE residual around temporal-then-spatial,
        # vivit.py:99); both adds run in the epilogue of the block's last GEMM, and the residual's gradient is
        # summed inside the backward kernel of the LayerNorm that shares its input (PreNorm fork / 'input').
        # Bias gradients of the three output projections: each projection's output is normalised by the NEXT block's
        # LayerNorm, whose backward kernel sums its input gradient over rows on the way (Fn.BiasSink) -- when the
        # gradient buffers are the fused bucket and no dropout sits between the projection and the LayerNorm.
        def sink_for(lin, drop):
            plain = drop.p == 0.0 or not self.training
            return Fn.BiasSink(lin.bias) if (plain and Fn.BiasSink.usable(lin.bias)) else None

        prev = None                              # sink for the previous layer's FeedForward output bias
        last = len(self.layers) - 1
        for li, (attn_t, attn_s, ff) in enumerate(self.layers):
            s_t = sink_for(attn_t.fn.to_out[0], attn_t.fn.to_out[1])
            s_s = sink_for(attn_s.fn.to_out[0], attn_s.fn.to_out[1])
            # the last FeedForward feeds the class-row LayerNorm only: it keeps its own bias-gradient pass
            s_f = sink_for(ff.fn.net[3], ff.fn.net[4]) if li != last else None
            if li == last and cls_of is not None and self.dead_row_elimination:
                b_, f_, p_ = cls_of
                y_t, x_res = attn_t(x, hw=hw, fork=True, sink=prev, defer_bias=False)
                y0 = Fn.TakeFrameFn.apply(y_t, b_, f_, p_)                       # (b, p, d): frame 0 of every clip
                s0 = attn_s(y0, hw=p_)                                           # LayerNorm + spatial attention + out proj
                x1 = Fn.add(Fn.TakeFirstFn.apply(s0), Fn.TakeClsFn.apply(x_res, b_, f_, p_))      # (b, d): the class rows
                x2 = ff(x1.view(b_, 1, -1), residual='input').view(b_, -1)
                return Fn.layer_norm(x2, self.norm.weight, self.norm.bias, self.norm.eps)
            y_t, x_res = attn_t(x, hw=hw, fork=True, sink=prev, defer_bias=s_t is not None)
            x = attn_s(y_t, hw=hw, residual=x_res, sink=s_t, defer_bias=s_s is not None)
            x = ff(x, residual='input', sink=s_s, defer_bias=s_f is not None)
            prev = s_f
        # The reference normalises every row and DSTTr then reads one per clip (vivit.py:100,144-146): the default does the
        # same.  LayerNorm is row-wise, so normalising only the rows that are read gives the same bits; like every other
        # skipped-dead-work shortcut it belongs to the opt-in dead_row_elimination, not to the default model.
        if cls_of is not None and self.dead_row_elimination:
            return Fn.layer_norm(Fn.TakeClsFn.apply(x, *cls_of), self.norm.weight, self.norm.bias, self.norm.eps)
        x = Fn.layer_norm(x, self.norm.weight, self.norm.bias, self.norm.eps)
        return Fn.TakeClsFn.apply(x, *cls_of) if cls_of is not None else x


class DSTTr(nn.Module):
    def __init__(self, image_size, patch_size, num_classes, num_frames, dim=728, depth=12, heads=8, pool='cls',
                 in_channels=728, dim_head=64, dropout=0., emb_dropout=0., scale_dim=4, compute_dtype=torch.float32):
        super().__init__()
        assert pool in {'cls', 'mean'}, 'pool type must be either cls (cls token) or mean (mean pooling)'
        assert image_size % patch_size == 0, 'Image dimensions must be divisible by the patch size.'
        num_patches = (image_size // patch_size) ** 2
        self.pos_embedding = nn.Parameter(torch.randn(1, num_frames, num_patches + 1, dim))
        self.space_token = nn.Parameter(torch.randn(1, 1, dim))
        self.temporal_token = nn.Parameter(torch.randn(1, 1, dim))
        self.transformer = STTransformer(dim, depth, heads, dim_head, dim * scale_dim, dropout)
        self.dropout = nn.Dropout(emb_dropout)      # constructed, never applied (as in the reference)
        self.pool = pool
        self.mlp_head = nn.Sequential(
            nn.LayerNorm(dim),
            nn.Linear(dim, num_classes)
        )
        self.compute_dtype = compute_dtype

    def forward_features(self, feats):
        """feats: (b, t, h*w, c) channels-last features -> (b, num_classes) float32 logits."""
        b, t, hw, c = feats.shape
        feats = ops.cast(feats, self.compute_dtype)
        x = Fn.TokensFn.apply(feats, self.space_token, self.temporal_token, self.pos_embedding)
        p = hw + 1
        cls = self.transformer(x, hw=p, cls_of=(b, t + 1, p))     # (b, c): temporal-token frame, space-token slot
        ln, fc = self.mlp_head[0], self.mlp_head[1]
        y = Fn.layer_norm(cls, ln.weight, ln.bias, ln.eps)
        y = Fn.LinearFn.apply(y, fc.weight, fc.bias, None)
        return y.float()

    def forward(self, x):
        """x: (b, t, c, h, w) as in the reference (vivit.py:132)."""
        b, t, c, h, w = x.shape
        return self.forward_features(x.flatten(3).transpose(2, 3).contiguous())


class XceptionVidTr(nn.Module):
    """Reference: vivit.py:193-208.  ``XceptionVidTr()`` reproduces the reference exactly
    (6 frames, 19x19 grid from 300^2 crops, depth 12); the keyword arguments generalise the
    geometry it hard-codes."""

    def __init__(self, *, num_frames=6, grid=19, depth=12, dim=728, heads=8, dim_head=64, scale_dim=4,
                 num_classes=1, compute_dtype=torch.float32, attn_fp8=False):
        super(XceptionVidTr, self).__init__()
        from ..models import model_selection
        self.xcep = model_selection(modelname='xception', num_out_classes=2, dropout=0.5, batch_size=1)
        self.vit = DSTTr(grid, 1, num_classes, num_frames, dim=dim, depth=depth, heads=heads, dim_head=dim_head,
                         scale_dim=scale_dim, compute_dtype=compute_dtype)
        self.compute_dtype = compute_dtype
        self.set_attn_fp8(attn_fp8)
        self._step_graphs = None
        if os.environ.get('ISTVT_STEP_GRAPHS', '0') == '1':
            self.enable_step_graphs(True)

    def set_attn_fp8(self, on=True):
        """fp8 (OCP e4m3) operands in the spatial-attention MFMAs (BASELINE.json configs[4]); needs bfloat16 compute."""
        if on and self.compute_dtype != torch.bfloat16:
            raise ValueError('attn_fp8 needs compute_dtype=torch.bfloat16')
        for m in self.modules():
            if isinstance(m, SpatialOnlyAttention):
                m.attn_fp8 = bool(on)
        return self

    def set_compute_dtype(self, dtype):
        self.compute_dtype = dtype
        self.vit.compute_dtype = dtype
        return self

    def set_dead_row_elimination(self, on=True):
        """skip the rows of the last layer that cannot reach the logits (STTransformer.dead_row_elimination)"""
        self.vit.transformer.dead_row_elimination = bool(on)
        return self

    def enable_step_graphs(self, on=True, warmup=2):
        """Replay the training forward and backward as two captured HIP graphs instead of ~1 400 launches issued from Python
        (istvt_amd.parallel.StepGraphs: preconditions, fall-backs and the address contract).  The caller's loop does not
        change: ``model(x)`` returns logits inside its autograd graph, ``loss.backward()`` replays the backward graph.
        Also switched on by ISTVT_STEP_GRAPHS=1 at construction."""
        if on:
            from istvt_amd import parallel
            self._step_graphs = parallel.StepGraphs(self, self._forward_eager, warmup=warmup)
        else:
            self._step_graphs = None
        return self

    def forward(self, x):
        g = self._step_graphs
        if g is not None:
            return g(x)
        return self._forward_eager(x)

    def _forward_eager(self, x):
        b, t = x.shape[:2]
        feats = self.xcep.model.low_level_features_nhwc(x.flatten(0, 1), self.compute_dtype)   # (b*t, h, w, c)
        n, h, w, c = feats.shape
        return self.vit.forward_features(feats.view(b, t, h * w, c))


def _head(mlp_head, x):
    ln, fc = mlp_head[0], mlp_head[1]
    y = Fn.layer_norm(x, ln.weight, ln.bias, ln.eps)
    return Fn.LinearFn.apply(y, fc.weight, fc.bias, None).float()


class ViViT(nn.Module):
    """Reference vivit.py:29-81 (factorised-encoder ablation): a space transformer per frame over
    [space_token | patches] + pos, whose cls rows -> [temporal_token | frames] -> temporal transformer -> cls / mean
    -> mlp_head.  (patch_size must be 1: the reference's patch Linear is commented out, vivit.py:40-43.)"""

    def __init__(self, image_size, patch_size, num_classes, num_frames, dim=728, depth=12, heads=8, pool='cls',
                 in_channels=728, dim_head=64, dropout=0., emb_dropout=0., scale_dim=4, compute_dtype=torch.float32):
        super().__init__()
        assert pool in {'cls', 'mean'}, 'pool type must be either cls (cls token) or mean (mean pooling)'
        assert image_size % patch_size == 0, 'Image dimensions must be divisible by the patch size.'
        num_patches = (image_size // patch_size) ** 2
        self.patch_size = patch_size
        self.to_patch_embedding = nn.Sequential()          # the reference's Rearrange has no parameters
        self.pos_embedding = nn.Parameter(torch.randn(1, num_frames, num_patches + 1, dim))
        self.space_token = nn.Parameter(torch.randn(1, 1, dim))
        self.space_transformer = Transformer(dim, depth, heads, dim_head, dim * scale_dim, dropout)
        self.temporal_token = nn.Parameter(torch.randn(1, 1, dim))
        self.temporal_transformer = Transformer(dim, depth, heads, dim_head, dim * scale_dim, dropout)
        self.dropout = nn.Dropout(emb_dropout)
        self.pool = pool
        self.mlp_head = nn.Sequential(
            nn.LayerNorm(dim),
            nn.Linear(dim, num_classes)
        )
        self.compute_dtype = compute_dtype

    def forward(self, x):
        """x: (b, t, c, h, w)"""
        if self.patch_size != 1:
            raise NotImplementedError('ViViT patch_size > 1 (the reference never projects the patches, vivit.py:42)')
        b, t, c, h, w = x.shape
        if t != self.pos_embedding.shape[1]:
            raise RuntimeError('The size of tensor a (%d) must match the size of tensor b (%d) at non-singleton dimension 1'
                               % (t, self.pos_embedding.shape[1]))
        feats = ops.cast(x.flatten(3).transpose(2, 3).contiguous(), self.compute_dtype).view(b * t, h * w, c)
        xs = Fn.PrependFn.apply(feats, self.space_token, self.pos_embedding, t)           # (b t) (n+1) d
        xs = Fn.dropout(xs, self.dropout.p, self.dropout.training)
        xs = self.space_transformer(xs)
        cls = Fn.TakeFirstFn.apply(xs).view(b, t, c)
        xt = Fn.PrependFn.apply(cls, self.temporal_token, None, 1)                         # b (t+1) d
        xt = self.temporal_transformer(xt)
        y = Fn.SeqMeanFn.apply(xt) if self.pool == 'mean' else Fn.TakeFirstFn.apply(xt)
        return _head(self.mlp_head, y)


class VanillaTr(nn.Module):
    """Reference vivit.py:150-191 (joint space-time ablation): Linear patch embedding of every (frame, position), one
    cls token, one transformer over all t*h*w + 1 tokens."""

    def __init__(self, image_size, patch_size, num_classes, num_frames, dim=728, depth=12, heads=8, pool='cls',
                 in_channels=728, dim_head=64, dropout=0., emb_dropout=0., scale_dim=4, compute_dtype=torch.float32):
        super().__init__()
        assert pool in {'cls', 'mean'}, 'pool type must be either cls (cls token) or mean (mean pooling)'
        assert image_size % patch_size == 0, 'Image dimensions must be divisible by the patch size.'
        num_patches = (image_size // patch_size) ** 2
        patch_dim = in_channels * patch_size ** 2
        # index 1 is the Linear, as in the reference's Sequential(Rearrange, Linear, Rearrange): same state-dict key
        self.to_patch_embedding = nn.Sequential(nn.Identity(), nn.Linear(patch_dim, dim), nn.Identity())
        self.pos_embedding = nn.Parameter(torch.randn(1, (num_frames * num_patches) + 1, dim))
        self.cls_token = nn.Parameter(torch.randn(1, 1, dim))
        self.transformer = Transformer(dim, depth, heads, dim_head, dim * scale_dim, dropout)
        self.dropout = nn.Dropout(emb_dropout)      # constructed, never applied (as in the reference)
        self.pool = pool
        self.mlp_head = nn.Sequential(
            nn.LayerNorm(dim),
            nn.Linear(dim, num_classes)
        )
        self.compute_dtype = compute_dtype

    def forward(self, x):
        b, t, c, h, w = x.shape
        fc = self.to_patch_embedding[1]
        feats = ops.cast(x.flatten(3).transpose(2, 3).contiguous(), self.compute_dtype).view(b * t * h * w, c)
        emb = Fn.LinearFn.apply(feats, fc.weight, fc.bias, None)
        n = t * h * w
        if n + 1 != self.pos_embedding.shape[1]:
            raise RuntimeError('The size of tensor a (%d) must match the size of tensor b (%d) at non-singleton dimension 1'
                               % (n + 1, self.pos_embedding.shape[1]))
        emb = emb.reshape(b, n, -1)
        xs = Fn.PrependFn.apply(emb, self.cls_token, self.pos_embedding, 1)
        xs = self.transformer(xs)
        return _head(self.mlp_head, Fn.TakeFirstFn.apply(xs))


for _cls in (Transformer, STTransformer, DSTTr, XceptionVidTr, ViViT, VanillaTr):
    _cls._replicate_for_data_parallel = Fn.no_data_parallel      # nn.DataParallel: see functional.no_data_parallel
