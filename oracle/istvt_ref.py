"""ORACLE — test infrastructure only.

CPU restatement (plain PyTorch fp32/fp64 ops, functional style) of the ISTVT hot path of
Vill-Lab/2023-TIFS-ISTVT: Xception entry-flow stem -> decomposed spatial-temporal
transformer.  It is the checker for the HIP path; nothing under ``2023-tifs-istvt_amd/``
imports it.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may use this file.

Parity status: PINNED.  Every function below is checked against golden vectors G1-G6
(``tests/golden/*.npz``) captured by ``tests/golden/make_golden.py`` from the
reference's own modules, see ``tests/test_oracle_golden.py``.

The restatement takes the geometry the reference hard-codes (tokens per frame
``19*19+1`` at network/vivit/module.py:84,192,197,198 and network/vivit/vivit.py:144,
frames ``6`` at vivit.py:201) as arguments, so it also runs at 14x14 / 6x6 grids
(224^2 / 96^2 inputs) which the reference itself cannot run.

All functions are pure: parameters come in as a ``{state_dict name: tensor}`` mapping
with the reference's names; gradients are obtained with
``torch.autograd`` over these functions.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
Params = Dict[str, Tensor]


# ----------------------------------------------------------------------------------
# Xception entry flow  (reference: network/xception.py)
# ----------------------------------------------------------------------------------
def _bn(p: Params, name: str, x: Tensor, training: bool, momentum: float = 0.1, eps: float = 1e-5) -> Tensor:
    """nn.BatchNorm2d (xception.py:58,69,75,119,123): batch statistics in train mode
    (biased variance normalises, unbiased variance goes to running_var), running
    statistics in eval mode."""
    rm = p.get(name + '.running_mean')
    rv = p.get(name + '.running_var')
    y = F.batch_norm(x, rm, rv, p[name + '.weight'], p[name + '.bias'], training, momentum, eps)
    nbt = p.get(name + '.num_batches_tracked')
    if training and nbt is not None:
        nbt += 1
    return y


def _sepconv(p: Params, name: str, x: Tensor) -> Tensor:
    """SeparableConv2d.forward (xception.py:46-49): depthwise 3x3 s1 p1 (groups=C, no
    bias) then pointwise 1x1 (no bias); nothing in between."""
    c = x.shape[1]
    x = F.conv2d(x, p[name + '.conv1.weight'], None, 1, 1, 1, c)
    return F.conv2d(x, p[name + '.pointwise.weight'])


def _block(p: Params, name: str, inp: Tensor, start_with_relu: bool, training: bool) -> Tensor:
    """Block(in,out,reps=2,strides=2,grow_first=True).forward (xception.py:52-101).

    rep = [ReLU?] Sep BN ReLU Sep BN MaxPool(3,2,1); the leading ReLU is the non-inplace
    one (xception.py:85) so the skip branch sees the un-rectified input; rep indices are
    positional: (0,1,3,4) without the leading ReLU, (1,2,4,5) with it."""
    i0 = 1 if start_with_relu else 0
    x = F.relu(inp) if start_with_relu else inp
    x = _sepconv(p, '%s.rep.%d' % (name, i0), x)
    x = _bn(p, '%s.rep.%d' % (name, i0 + 1), x, training)
    x = F.relu(x)
    x = _sepconv(p, '%s.rep.%d' % (name, i0 + 3), x)
    x = _bn(p, '%s.rep.%d' % (name, i0 + 4), x, training)
    x = F.max_pool2d(x, 3, 2, 1)
    skip = F.conv2d(inp, p[name + '.skip.weight'], None, 2)
    skip = _bn(p, name + '.skipbn', skip, training)
    return x + skip


def stem_forward(p: Params, x: Tensor, prefix: str = '', training: bool = True) -> Tensor:
    """Xception.low_level_features (xception.py:193-206): (n,3,S,S) -> (n,728,h,h)."""
    q = {k[len(prefix):]: v for k, v in p.items() if k.startswith(prefix)} if prefix else p
    x = F.conv2d(x, q['conv1.weight'], None, 2, 0)
    x = F.relu(_bn(q, 'bn1', x, training))
    x = F.conv2d(x, q['conv2.weight'])
    x = F.relu(_bn(q, 'bn2', x, training))
    x = _block(q, 'block1', x, False, training)
    x = _block(q, 'block2', x, True, training)
    x = _block(q, 'block3', x, True, training)
    return x


def stem_out_side(side: int) -> int:
    """Spatial size after conv1(s2,p0), conv2(p0) and three MaxPool(3,2,1)."""
    s = (side - 3) // 2 + 1
    s = s - 2
    for _ in range(3):
        s = (s - 1) // 2 + 1
    return s


# ----------------------------------------------------------------------------------
# transformer modules  (reference: network/vivit/module.py)
# ----------------------------------------------------------------------------------
def layer_norm(p: Params, name: str, x: Tensor, eps: float = 1e-5) -> Tensor:
    """nn.LayerNorm(dim) as used by PreNorm (module.py:15-21) and vivit.py:89,128."""
    w = p[name + '.weight']
    return F.layer_norm(x, (w.shape[0],), w, p[name + '.bias'], eps)


def feed_forward(p: Params, name: str, x: Tensor) -> Tensor:
    """FeedForward (module.py:23-34): Linear -> exact-erf GELU -> Linear (dropouts are p=0)."""
    h = F.linear(x, p[name + '.net.0.weight'], p[name + '.net.0.bias'])
    h = F.gelu(h)
    return F.linear(h, p[name + '.net.3.weight'], p[name + '.net.3.bias'])


def _heads_split(t: Tensor, b: int, frames: int, hw: int, heads: int) -> Tensor:
    # 'b (t hw) (h d) -> b h t hw d'
    d = t.shape[-1] // heads
    return t.view(b, frames, hw, heads, d).permute(0, 3, 1, 2, 4)


def spatial_attention(p: Params, name: str, x: Tensor, hw: int, heads: int) -> Tensor:
    """SpatialOnlyAttention.forward (module.py:81-93): softmax(q k^T / sqrt(d)) v per
    (batch, head, frame) over the hw tokens of that frame; to_qkv has no bias."""
    b, n, _ = x.shape
    frames = n // hw
    qkv = F.linear(x, p[name + '.to_qkv.weight']).chunk(3, dim=-1)
    q, k, v = (_heads_split(t, b, frames, hw, heads) for t in qkv)          # b h t hw d
    scale = q.shape[-1] ** -0.5
    dots = torch.matmul(q, k.transpose(-1, -2)) * scale
    attn = dots.softmax(dim=-1)
    out = torch.matmul(attn, v)                                               # b h t hw d
    out = out.permute(0, 2, 3, 1, 4).reshape(b, n, -1)                        # b (t hw) (h d)
    return F.linear(out, p[name + '.to_out.0.weight'], p[name + '.to_out.0.bias'])


def temporal_residual_attention(p: Params, name: str, x: Tensor, hw: int, heads: int) -> Tensor:
    """TemporalResidualAttention.forward (module.py:190-208): q,k come from the frame
    *difference* (frames 0,1 raw, frame f>=2 minus frame f-1; module.py:193), v from the
    un-differenced input; attention runs over the frame axis per (batch, head, position)."""
    b, n, dim = x.shape
    frames = n // hw
    xr = x.view(b, frames, hw, dim)
    res = torch.cat((xr[:, 0:2], xr[:, 2:] - xr[:, 1:-1]), dim=1).reshape(b, n, dim)
    q, k = F.linear(res, p[name + '.to_qk.weight']).chunk(2, dim=-1)
    v = F.linear(x, p[name + '.to_v.weight'])
    # 'b (t hw) (h d) -> b h hw t d'
    q, k, v = (_heads_split(t, b, frames, hw, heads).transpose(2, 3) for t in (q, k, v))
    scale = q.shape[-1] ** -0.5
    dots = torch.matmul(q, k.transpose(-1, -2)) * scale                       # b h hw t t
    attn = dots.softmax(dim=-1)
    out = torch.matmul(attn, v)                                               # b h hw t d
    out = out.permute(0, 3, 2, 1, 4).reshape(b, n, -1)                        # b (t hw) (h d)
    return F.linear(out, p[name + '.to_out.0.weight'], p[name + '.to_out.0.bias'])


def st_transformer(p: Params, name: str, x: Tensor, depth: int, hw: int, heads: int) -> Tensor:
    """STTransformer.forward (vivit.py:97-101): per layer
    ``x = S(LN_s(T(LN_t(x)))) + x`` (ONE residual around temporal-then-spatial) and
    ``x = FF(LN_f(x)) + x``; final LayerNorm."""
    for i in range(depth):
        lp = '%s.layers.%d' % (name, i)
        t = temporal_residual_attention(p, lp + '.0.fn', layer_norm(p, lp + '.0.norm', x), hw, heads)
        s = spatial_attention(p, lp + '.1.fn', layer_norm(p, lp + '.1.norm', t), hw, heads)
        x = s + x
        x = feed_forward(p, lp + '.2.fn', layer_norm(p, lp + '.2.norm', x)) + x
    return layer_norm(p, name + '.norm', x)


def dsttr_forward(p: Params, feats: Tensor, prefix: str = '', depth: int = 12, heads: int = 8) -> Tensor:
    """DSTTr.forward (vivit.py:132-148): (b,t,c,h,w) features -> (b,num_classes) logits."""
    q = {k[len(prefix):]: v for k, v in p.items() if k.startswith(prefix)} if prefix else p
    b, t, c, h, w = feats.shape
    x = feats.flatten(3).transpose(2, 3)                                      # b t (h w) c
    n = x.shape[2]
    space = q['space_token'].view(1, 1, 1, c).expand(b, t, 1, c)
    x = torch.cat((space, x), dim=2)
    x = x + q['pos_embedding'][:, :, :n + 1]            # needs t == num_frames (vivit.py:138)
    temporal = q['temporal_token'].view(1, 1, 1, c).expand(b, 1, n + 1, c)   # after pos-emb (vivit.py:139-140)
    x = torch.cat((temporal, x), dim=1)
    hw = n + 1
    x = x.reshape(b, (t + 1) * hw, c)
    x = st_transformer(q, 'transformer', x, depth, hw, heads)
    x = x.view(b, t + 1, hw, c)[:, 0, 0]
    x = layer_norm(q, 'mlp_head.0', x)
    return F.linear(x, q['mlp_head.1.weight'], q['mlp_head.1.bias'])


def xception_vidtr_forward(p: Params, x: Tensor, depth: int = 12, heads: int = 8, training: bool = True) -> Tensor:
    """XceptionVidTr.forward (vivit.py:202-208): fold T into batch, stem, unfold, DSTTr."""
    b, t = x.shape[:2]
    f = stem_forward(p, x.flatten(0, 1), 'xcep.model.', training)
    f = f.view(b, t, *f.shape[1:])
    return dsttr_forward(p, f, 'vit.', depth, heads)


def bce_with_logits(logits: Tensor, labels: Tensor) -> Tensor:
    """criterion at train_CNN.py:148,526: BCEWithLogitsLoss(mean)(outputs.view(-1), labels.float())."""
    return F.binary_cross_entropy_with_logits(logits.view(-1), labels.float())


# ----------------------------------------------------------------------------------
# parameter containers
# ----------------------------------------------------------------------------------
def stem_param_shapes() -> Dict[str, tuple]:
    """Shapes of the stem entries that ``low_level_features`` touches (conv1..block3)."""
    s: Dict[str, tuple] = {}

    def bn(name, c):
        s[name + '.weight'] = (c,)
        s[name + '.bias'] = (c,)
        s[name + '.running_mean'] = (c,)
        s[name + '.running_var'] = (c,)
        s[name + '.num_batches_tracked'] = ()

    s['conv1.weight'] = (32, 3, 3, 3)
    bn('bn1', 32)
    s['conv2.weight'] = (64, 32, 3, 3)
    bn('bn2', 64)
    for name, cin, cout, relu in (('block1', 64, 128, False), ('block2', 128, 256, True), ('block3', 256, 728, True)):
        i0 = 1 if relu else 0
        s[name + '.skip.weight'] = (cout, cin, 1, 1)
        bn(name + '.skipbn', cout)
        s['%s.rep.%d.conv1.weight' % (name, i0)] = (cin, 1, 3, 3)
        s['%s.rep.%d.pointwise.weight' % (name, i0)] = (cout, cin, 1, 1)
        bn('%s.rep.%d' % (name, i0 + 1), cout)
        s['%s.rep.%d.conv1.weight' % (name, i0 + 3)] = (cout, 1, 3, 3)
        s['%s.rep.%d.pointwise.weight' % (name, i0 + 3)] = (cout, cout, 1, 1)
        bn('%s.rep.%d' % (name, i0 + 4), cout)
    return s


def dsttr_param_shapes(num_frames: int, grid: int, dim: int = 728, depth: int = 12, heads: int = 8,
                       dim_head: int = 64, scale_dim: int = 4, num_classes: int = 1) -> Dict[str, tuple]:
    inner = heads * dim_head
    mlp = dim * scale_dim
    s: Dict[str, tuple] = {
        'pos_embedding': (1, num_frames, grid * grid + 1, dim),
        'space_token': (1, 1, dim),
        'temporal_token': (1, 1, dim),
    }
    for i in range(depth):
        lp = 'transformer.layers.%d' % i
        for j in range(3):
            s['%s.%d.norm.weight' % (lp, j)] = (dim,)
            s['%s.%d.norm.bias' % (lp, j)] = (dim,)
        s[lp + '.0.fn.to_qk.weight'] = (2 * inner, dim)
        s[lp + '.0.fn.to_v.weight'] = (inner, dim)
        s[lp + '.0.fn.to_out.0.weight'] = (dim, inner)
        s[lp + '.0.fn.to_out.0.bias'] = (dim,)
        s[lp + '.1.fn.to_qkv.weight'] = (3 * inner, dim)
        s[lp + '.1.fn.to_out.0.weight'] = (dim, inner)
        s[lp + '.1.fn.to_out.0.bias'] = (dim,)
        s[lp + '.2.fn.net.0.weight'] = (mlp, dim)
        s[lp + '.2.fn.net.0.bias'] = (mlp,)
        s[lp + '.2.fn.net.3.weight'] = (dim, mlp)
        s[lp + '.2.fn.net.3.bias'] = (dim,)
    s['transformer.norm.weight'] = (dim,)
    s['transformer.norm.bias'] = (dim,)
    s['mlp_head.0.weight'] = (dim,)
    s['mlp_head.0.bias'] = (dim,)
    s['mlp_head.1.weight'] = (num_classes, dim)
    s['mlp_head.1.bias'] = (num_classes,)
    return s


def random_params(shapes: Dict[str, tuple], seed: int = 0, dtype=torch.float32) -> Params:
    """Default-nn-style random init under a seed (kaiming-uniform-like bounds), for the
    CPU baseline and for HIP-vs-oracle parity at geometries the goldens cannot cover."""
    g = torch.Generator().manual_seed(seed)
    out: Params = {}
    for name, shape in shapes.items():
        leaf = name.split('.')[-1]
        if leaf == 'num_batches_tracked':
            out[name] = torch.zeros((), dtype=torch.long)
        elif leaf == 'running_mean':
            out[name] = torch.zeros(shape, dtype=dtype)
        elif leaf == 'running_var':
            out[name] = torch.ones(shape, dtype=dtype)
        elif leaf in ('pos_embedding', 'space_token', 'temporal_token'):
            out[name] = torch.randn(shape, generator=g, dtype=dtype)
        elif len(shape) == 1 and leaf == 'weight':
            out[name] = torch.ones(shape, dtype=dtype) + 0.1 * torch.randn(shape, generator=g, dtype=dtype)
        elif leaf == 'bias':
            out[name] = 0.1 * torch.randn(shape, generator=g, dtype=dtype)
        else:
            fan_in = int(math.prod(shape[1:]))
            bound = 1.0 / math.sqrt(fan_in)
            out[name] = (torch.rand(shape, generator=g, dtype=dtype) * 2 - 1) * bound
    return out


def with_grad(p: Params) -> Params:
    return {k: (v.clone().requires_grad_(True) if v.is_floating_point() and 'running' not in k else v.clone())
            for k, v in p.items()}


# ----------------------------------------------------------------------------------
# Round 2 -- SURVEY 8(f) rows 3 and 4: the whole Xception (middle + exit flow) and the
# ablation attention variants.  Pinned by goldens G7 / G8 (tests/test_oracle_golden.py).
# ----------------------------------------------------------------------------------
def block_units(cin: int, cout: int, reps: int, grow_first: bool):
    """(in, out) channels of the separable convolutions of Block(cin, cout, reps, ...) in execution order
    (xception.py:64-80)."""
    units = []
    filters = cin
    if grow_first:
        units.append((cin, cout))
        filters = cout
    units += [(filters, filters)] * (reps - 1)
    if not grow_first:
        units.append((cin, cout))
    return units


def block_forward(p: Params, name: str, inp: Tensor, cin: int, cout: int, reps: int, strides: int = 1,
                  start_with_relu: bool = True, grow_first: bool = True, training: bool = True) -> Tensor:
    """Block.forward for any constructor arguments (xception.py:52-101).  rep is [ReLU, Sep, BN] per unit with the
    leading ReLU dropped (start_with_relu=False) or made non-inplace (the skip branch sees the raw input either way),
    MaxPool2d(3, strides, 1) appended when strides != 1; skip = BN(Conv1x1(stride)) when the shape changes, identity
    otherwise.  `name` may be '' for a bare Block's own state dict."""
    pre = name + '.' if name else ''
    units = block_units(cin, cout, reps, grow_first)
    idx = 0 if not start_with_relu else 1           # position of the first SeparableConv2d inside rep
    x = inp
    for i, _ in enumerate(units):
        if i > 0 or start_with_relu:
            x = F.relu(x)
        x = _sepconv(p, '%srep.%d' % (pre, idx), x)
        x = _bn(p, '%srep.%d' % (pre, idx + 1), x, training)
        idx += 3
    if strides != 1:
        x = F.max_pool2d(x, 3, strides, 1)
    if (pre + 'skip.weight') in p:
        skip = F.conv2d(inp, p[pre + 'skip.weight'], None, strides)
        skip = _bn(p, pre + 'skipbn', skip, training)
    else:
        skip = inp
    return x + skip


XCEPTION_BLOCKS = ([('block1', 64, 128, 2, 2, False, True), ('block2', 128, 256, 2, 2, True, True),
                    ('block3', 256, 728, 2, 2, True, True)]
                   + [('block%d' % i, 728, 728, 3, 1, True, True) for i in range(4, 12)]
                   + [('block12', 728, 1024, 2, 2, True, False)])


def xception_features(p: Params, x: Tensor, training: bool = True) -> Tensor:
    """Xception.features (xception.py:161-191): entry, middle and exit flow up to bn4 (no final ReLU)."""
    x = F.relu(_bn(p, 'bn1', F.conv2d(x, p['conv1.weight'], None, 2, 0), training))
    x = F.relu(_bn(p, 'bn2', F.conv2d(x, p['conv2.weight']), training))
    for name, cin, cout, reps, strides, relu0, grow in XCEPTION_BLOCKS:
        x = block_forward(p, name, x, cin, cout, reps, strides, relu0, grow, training)
    x = F.relu(_bn(p, 'bn3', _sepconv(p, 'conv3', x), training))
    return _bn(p, 'bn4', _sepconv(p, 'conv4', x), training)


def xception_logits(p: Params, feats: Tensor, head: str = 'last_linear') -> Tensor:
    """Xception.logits (xception.py:208-215): ReLU, global average pool, last_linear."""
    x = F.adaptive_avg_pool2d(F.relu(feats), (1, 1)).flatten(1)
    return F.linear(x, p[head + '.weight'], p[head + '.bias'])


# ---- DualNet's Xception halves (network/xception_for_dualnet.py:215-284; consumed at dual_net.py:210-232) ----
def _xception_entry(p: Params, x: Tensor, training: bool) -> Tensor:
    x = F.relu(_bn(p, 'bn1', F.conv2d(x, p['conv1.weight'], None, 2, 0), training))
    return F.relu(_bn(p, 'bn2', F.conv2d(x, p['conv2.weight']), training))


def _xception_blocks(p: Params, x: Tensor, first: int, last: int, training: bool) -> Tensor:
    for name, cin, cout, reps, strides, relu0, grow in XCEPTION_BLOCKS[first - 1:last]:
        x = block_forward(p, name, x, cin, cout, reps, strides, relu0, grow, training)
    return x


def _xception_exit(p: Params, x: Tensor, training: bool) -> Tensor:
    x = F.relu(_bn(p, 'bn3', _sepconv(p, 'conv3', x), training))
    return _bn(p, 'bn4', _sepconv(p, 'conv4', x), training)


def xception_fea(p: Params, which: str, x: Tensor, training: bool = True) -> Tensor:
    """The partial forwards of xception_for_dualnet.Xception: 'fea_0_7' (conv1 .. block7, :215-231), 'fea_8_12' (block8
    .. bn4, :233-246), 'fea_0_4' (:248-262), 'fea_5_8' (:264-270), 'fea_9_12' (:272-284).  Same layers and state-dict
    names as network/xception.py's Xception; the exit flow ends at bn4 with no ReLU."""
    if which == 'fea_0_7':
        return _xception_blocks(p, _xception_entry(p, x, training), 1, 7, training)
    if which == 'fea_0_4':
        return _xception_blocks(p, _xception_entry(p, x, training), 1, 4, training)
    if which == 'fea_5_8':
        return _xception_blocks(p, x, 5, 8, training)
    if which == 'fea_8_12':
        return _xception_exit(p, _xception_blocks(p, x, 8, 12, training), training)
    if which == 'fea_9_12':
        return _xception_exit(p, _xception_blocks(p, x, 9, 12, training), training)
    raise ValueError(which)


def xception_dualnet_logits(p: Params, feats: Tensor, head: str = 'last_linear'):
    """xception_for_dualnet.Xception.logits (:317-324) in eval mode (dp is then the identity): (pooled features, head
    output)."""
    y = F.adaptive_avg_pool2d(F.relu(feats), (1, 1)).flatten(1)
    return y, F.linear(y, p[head + '.weight'], p[head + '.bias'])


def xception_param_shapes(num_classes: int = 1000) -> Dict[str, tuple]:
    s: Dict[str, tuple] = {}

    def bn(name, c):
        s[name + '.weight'] = (c,); s[name + '.bias'] = (c,)
        s[name + '.running_mean'] = (c,); s[name + '.running_var'] = (c,); s[name + '.num_batches_tracked'] = ()

    s['conv1.weight'] = (32, 3, 3, 3); bn('bn1', 32)
    s['conv2.weight'] = (64, 32, 3, 3); bn('bn2', 64)
    for name, cin, cout, reps, strides, relu0, grow in XCEPTION_BLOCKS:
        s.update({name + '.' + k: v for k, v in block_param_shapes(cin, cout, reps, strides, relu0, grow).items()})
    s['conv3.conv1.weight'] = (1024, 1, 3, 3); s['conv3.pointwise.weight'] = (1536, 1024, 1, 1); bn('bn3', 1536)
    s['conv4.conv1.weight'] = (1536, 1, 3, 3); s['conv4.pointwise.weight'] = (2048, 1536, 1, 1); bn('bn4', 2048)
    s['last_linear.weight'] = (num_classes, 2048); s['last_linear.bias'] = (num_classes,)
    return s


def block_param_shapes(cin: int, cout: int, reps: int, strides: int, start_with_relu: bool, grow_first: bool):
    s: Dict[str, tuple] = {}

    def bn(name, c):
        s[name + '.weight'] = (c,); s[name + '.bias'] = (c,)
        s[name + '.running_mean'] = (c,); s[name + '.running_var'] = (c,); s[name + '.num_batches_tracked'] = ()

    if cout != cin or strides != 1:
        s['skip.weight'] = (cout, cin, 1, 1)
        bn('skipbn', cout)
    idx = 0 if not start_with_relu else 1
    for a, b in block_units(cin, cout, reps, grow_first):
        s['rep.%d.conv1.weight' % idx] = (a, 1, 3, 3)
        s['rep.%d.pointwise.weight' % idx] = (b, a, 1, 1)
        bn('rep.%d' % (idx + 1), b)
        idx += 3
    return s


# ---- ablation attention variants (module.py:36-64, 145-172; vivit.py:10-25, 29-81, 150-191) ----
def attention(p: Params, name: str, x: Tensor, heads: int) -> Tensor:
    """Attention.forward (module.py:52-64): plain multi-head self-attention over all n tokens of a sequence."""
    b, n, _ = x.shape
    q, k, v = (t.view(b, n, heads, -1).transpose(1, 2) for t in F.linear(x, p[name + '.to_qkv.weight']).chunk(3, dim=-1))
    dots = torch.matmul(q, k.transpose(-1, -2)) * q.shape[-1] ** -0.5
    out = torch.matmul(dots.softmax(dim=-1), v).transpose(1, 2).reshape(b, n, -1)
    return F.linear(out, p[name + '.to_out.0.weight'], p[name + '.to_out.0.bias'])


def temporal_only_attention(p: Params, name: str, x: Tensor, hw: int, heads: int) -> Tensor:
    """TemporalOnlyAttention.forward (module.py:161-172): one packed to_qkv, attention over the frame axis per
    (batch, head, position); no frame difference."""
    b, n, _ = x.shape
    frames = n // hw
    q, k, v = (_heads_split(t, b, frames, hw, heads).transpose(2, 3)
               for t in F.linear(x, p[name + '.to_qkv.weight']).chunk(3, dim=-1))          # b h hw t d
    dots = torch.matmul(q, k.transpose(-1, -2)) * q.shape[-1] ** -0.5
    out = torch.matmul(dots.softmax(dim=-1), v).permute(0, 3, 2, 1, 4).reshape(b, n, -1)
    return F.linear(out, p[name + '.to_out.0.weight'], p[name + '.to_out.0.bias'])


def transformer(p: Params, name: str, x: Tensor, depth: int, heads: int) -> Tensor:
    """Transformer.forward (vivit.py:21-25): x = attn(LN(x)) + x ; x = ff(LN(x)) + x ; final LayerNorm."""
    for i in range(depth):
        lp = '%s.layers.%d' % (name, i)
        x = attention(p, lp + '.0.fn', layer_norm(p, lp + '.0.norm', x), heads) + x
        x = feed_forward(p, lp + '.1.fn', layer_norm(p, lp + '.1.norm', x)) + x
    return layer_norm(p, name + '.norm', x)


def vivit_forward(p: Params, feats: Tensor, depth: int, heads: int, pool: str = 'cls') -> Tensor:
    """ViViT.forward (vivit.py:60-81): per-frame space transformer on [space_token | patches] + pos, its cls rows
    -> [temporal_token | frames] -> temporal transformer -> cls (or mean) -> mlp_head."""
    b, t, c, h, w = feats.shape
    x = feats.flatten(3).transpose(2, 3)                                      # b t (h w) c
    n = x.shape[2]
    x = torch.cat((p['space_token'].view(1, 1, 1, c).expand(b, t, 1, c), x), dim=2) + p['pos_embedding'][:, :, :n + 1]
    x = transformer(p, 'space_transformer', x.reshape(b * t, n + 1, c), depth, heads)
    x = x[:, 0].view(b, t, c)
    x = torch.cat((p['temporal_token'].view(1, 1, c).expand(b, 1, c), x), dim=1)
    x = transformer(p, 'temporal_transformer', x, depth, heads)
    x = x.mean(dim=1) if pool == 'mean' else x[:, 0]
    return F.linear(layer_norm(p, 'mlp_head.0', x), p['mlp_head.1.weight'], p['mlp_head.1.bias'])


def vanilla_tr_forward(p: Params, feats: Tensor, depth: int, heads: int) -> Tensor:
    """VanillaTr.forward (vivit.py:180-191): Linear patch embedding of every (frame, position), one cls token, one
    joint transformer over all t*h*w + 1 tokens."""
    b, t, c, h, w = feats.shape
    x = F.linear(feats.flatten(3).transpose(2, 3), p['to_patch_embedding.1.weight'], p['to_patch_embedding.1.bias'])
    x = x.reshape(b, t * h * w, -1)
    x = torch.cat((p['cls_token'].expand(b, 1, -1), x), dim=1) + p['pos_embedding']
    x = transformer(p, 'transformer', x, depth, heads)[:, 0]
    return F.linear(layer_norm(p, 'mlp_head.0', x), p['mlp_head.1.weight'], p['mlp_head.1.bias'])
