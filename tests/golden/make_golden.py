#!/usr/bin/env python3
"""Capture golden vectors G1-G6 from the *reference's own modules* (SURVEY.md section 8(c)).

Runs ONLY in the build container, where /root/reference is mounted.  It imports the
reference's ``network.xception``, ``network.vivit.module`` and ``network.vivit.vivit``
(the last one needs a stand-in for ``network.models_copy``, whose third-party imports are
absent), drives them with the closed-form tensors of ``recipe.py`` and stores only
OUTPUTS (plus the config that produced them) as small ``.npz`` files next to this script.
No reference source or bytecode is written anywhere.

    python tests/golden/make_golden.py [--only G1,G5]
"""
import argparse
import os
import sys
import types

import numpy as np

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
REF = os.environ.get('ISTVT_REFERENCE', '/root/reference')
sys.path.insert(0, REF)

import torch  # noqa: E402
from torch import nn  # noqa: E402

import recipe  # noqa: E402

torch.set_num_threads(os.cpu_count() or 1)


# --------------------------------------------------------------------------------------
# reference imports (+ the one stub)
# --------------------------------------------------------------------------------------
import network.xception as ref_xception  # noqa: E402
import network.vivit.module as ref_module  # noqa: E402


class _XcepWrapper(nn.Module):
    """Stand-in for reference TransferModel('xception') (models_copy.py:34-45, 233-234):
    `.model` is the Xception with `last_linear = Sequential(Dropout, Linear(2048, n))`."""

    def __init__(self, num_out_classes):
        super().__init__()
        self.model = ref_xception.return_pytorch04_xception(pretrained=False)
        num_ftrs = self.model.last_linear.in_features
        self.model.last_linear = nn.Sequential(nn.Dropout(p=0.5), nn.Linear(num_ftrs, num_out_classes))

    def low_level_features(self, x):
        return self.model.low_level_features(x)


def _stub_model_selection(modelname, num_out_classes, dropout=None, batch_size=16):
    assert modelname == 'xception'
    return _XcepWrapper(num_out_classes)


_stub = types.ModuleType('network.models_copy')
_stub.model_selection = _stub_model_selection
sys.modules['network.models_copy'] = _stub
import network.vivit.vivit as ref_vivit  # noqa: E402


# --------------------------------------------------------------------------------------
def load_recipe(module: nn.Module, prefix: str = ''):
    sd = module.state_dict()
    vals = recipe.fill_state_dict(sd, prefix)
    module.load_state_dict({k: torch.from_numpy(v) for k, v in vals.items()})


def t(a):
    return torch.from_numpy(a)


def npy(x):
    return x.detach().cpu().numpy()


def save(name, **arrays):
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **arrays)
    print('wrote', path, os.path.getsize(path) // 1024, 'KiB')


# --------------------------------------------------------------------------------------
def g1_stem():
    out = {}
    for side in (96, 139):
        net = ref_xception.xception(pretrained=False)
        load_recipe(net, 'xcep.model.')
        net.train()
        x = t(recipe.input_value('g1.x%d' % side, (2, 3, side, side))).requires_grad_(True)
        y = net.low_level_features(x)
        coef = t(recipe.input_value('g1.coef%d' % side, tuple(y.shape)))
        (y * coef).sum().backward()
        tag = 's%d.' % side
        out[tag + 'y'] = npy(y)
        out[tag + 'dx_norm'] = npy(x.grad.norm())
        out[tag + 'dx_slice'] = npy(x.grad[0, :, 10:14, 20:28])
        sd = net.state_dict()
        for k in ('bn1', 'bn2', 'block1.skipbn', 'block2.rep.2', 'block3.rep.5', 'block3.skipbn'):
            out[tag + k + '.running_mean'] = npy(sd[k + '.running_mean'])
            out[tag + k + '.running_var'] = npy(sd[k + '.running_var'])
        for k, p in net.named_parameters():
            if p.grad is not None:
                out[tag + 'gnorm.' + k] = npy(p.grad.norm())
        for k in ('conv1.weight', 'conv2.weight', 'block1.rep.0.conv1.weight', 'block2.rep.4.conv1.weight',
                  'block3.skip.weight', 'block3.rep.5.weight', 'block3.rep.5.bias', 'bn1.weight',
                  'block2.rep.1.pointwise.weight'):
            g = dict(net.named_parameters())[k].grad
            out[tag + 'grad.' + k] = npy(g.reshape(-1)[:4096])
    save('G1_stem', **out)


MOD_CFG = dict(dim=64, heads=2, dim_head=32)
ROW_STRIDE = {5: 3, 9: 7}


def _run_module(mod, x_name, shape):
    x = t(recipe.input_value(x_name, shape)).requires_grad_(True)
    y = mod(x)
    coef = t(recipe.input_value(x_name + '.coef', tuple(y.shape)))
    (y * coef).sum().backward()
    res = {'y': npy(y), 'dx': npy(x.grad)}
    for k, p in mod.named_parameters():
        res['grad.' + k] = npy(p.grad)
    return res


def g2_modules():
    out = {}
    dim, heads, dh = MOD_CFG['dim'], MOD_CFG['heads'], MOD_CFG['dim_head']
    for F in (5, 9):
        shape = (1, F * 362, dim)
        mods = {
            'prenorm_ff': ref_module.PreNorm(dim, ref_module.FeedForward(dim, 4 * dim)),
            'ff': ref_module.FeedForward(dim, 4 * dim),
            'spatial': ref_module.SpatialOnlyAttention(dim, heads=heads, dim_head=dh),
            'temporal': ref_module.TemporalResidualAttention(dim, heads=heads, dim_head=dh),
        }
        for name, mod in mods.items():
            load_recipe(mod, 'g2.%s.' % name)
            res = _run_module(mod, 'g2.%s.F%d' % (name, F), shape)
            for k, v in res.items():
                if k in ('y', 'dx'):      # row subsample keeps the fixture small
                    v = v[:, ::ROW_STRIDE[F]]
                out['F%d.%s.%s' % (F, name, k)] = v
    save('G2_modules', **out)


def g3_layer():
    dim, heads, dh = MOD_CFG['dim'], MOD_CFG['heads'], MOD_CFG['dim_head']
    mod = ref_vivit.STTransformer(dim, 2, heads, dh, 2 * dim)
    load_recipe(mod, 'g3.')
    res = _run_module(mod, 'g3.x', (1, 5 * 362, dim))
    res['y'] = res['y'][:, ::3]
    res['dx'] = res['dx'][:, ::3]
    save('G3_sttransformer', **res)


def g4_dsttr():
    out = {}
    dim, heads, dh = MOD_CFG['dim'], MOD_CFG['heads'], MOD_CFG['dim_head']
    for T in (4, 8):
        mod = ref_vivit.DSTTr(19, 1, 1, T, dim=dim, depth=2, heads=heads, dim_head=dh, in_channels=dim, scale_dim=2)
        load_recipe(mod, 'g4.')
        x = t(recipe.input_value('g4.x.T%d' % T, (2, T, dim, 19, 19))).requires_grad_(True)
        y = mod(x)
        coef = t(recipe.input_value('g4.coef', tuple(y.shape)))
        (y * coef).sum().backward()
        tag = 'T%d.' % T
        out[tag + 'logits'] = npy(y)
        out[tag + 'dx_norm'] = npy(x.grad.norm())
        out[tag + 'dx_frame_norms'] = npy(x.grad.flatten(2).norm(dim=2))
        for k, p in mod.named_parameters():
            out[tag + 'gnorm.' + k] = npy(p.grad.norm())
        out[tag + 'grad.pos_embedding'] = npy(mod.pos_embedding.grad[0, :, ::37])
        out[tag + 'grad.space_token'] = npy(mod.space_token.grad)
        out[tag + 'grad.temporal_token'] = npy(mod.temporal_token.grad)
    save('G4_dsttr', **out)


def g5_end_to_end():
    """Reference-native geometry: XceptionVidTr() on (1,6,3,300,300), BCE vs label 1, one SGD step."""
    model = ref_vivit.XceptionVidTr()
    load_recipe(model, '')
    model.train()
    x = t(recipe.input_value('g5.x', (1, 6, 3, 300, 300)))
    labels = torch.ones(1)
    params = [p for p in model.parameters() if p.requires_grad]
    opt = torch.optim.SGD(params, lr=1e-3, momentum=0.9, weight_decay=0)
    opt.zero_grad()
    logits = model(x)
    loss = nn.BCEWithLogitsLoss()(logits.view(-1), labels.float())
    loss.backward()
    out = {'logits': npy(logits), 'loss': npy(loss)}
    names = []
    for k, p in model.named_parameters():
        if p.grad is not None:
            names.append(k)
            out['gnorm.' + k] = npy(p.grad.norm())
    out['live_param_names'] = np.array(names)
    slice_names = ['vit.transformer.layers.0.0.fn.to_qk.weight', 'vit.transformer.layers.11.2.fn.net.3.weight',
                   'vit.transformer.layers.5.1.fn.to_qkv.weight', 'vit.pos_embedding', 'xcep.model.conv1.weight',
                   'xcep.model.block3.rep.4.pointwise.weight']
    named = dict(model.named_parameters())
    for k in slice_names:
        out['grad.' + k] = npy(named[k].grad.reshape(-1)[:64])
    opt.step()
    for k in slice_names:
        out['after_sgd.' + k] = npy(named[k].reshape(-1)[:64])
    sd = model.state_dict()
    out['bn1.running_mean'] = npy(sd['xcep.model.bn1.running_mean'])
    out['bn1.running_var'] = npy(sd['xcep.model.bn1.running_var'])
    save('G5_native', **out)


def g6_fullwidth():
    mod = ref_vivit.DSTTr(19, 1, 1, 8, depth=2)
    load_recipe(mod, 'vit.')
    x = t(recipe.input_value('g6.x', (1, 8, 728, 19, 19))).requires_grad_(True)
    y = mod(x)
    y.sum().backward()
    out = {'logits': npy(y), 'dx_norm': npy(x.grad.norm()), 'dx_frame_norms': npy(x.grad.flatten(2).norm(dim=2))}
    for k, p in mod.named_parameters():
        out['gnorm.' + k] = npy(p.grad.norm())
    out['grad.qk0'] = npy(dict(mod.named_parameters())['transformer.layers.0.0.fn.to_qk.weight'].grad.reshape(-1)[:256])
    save('G6_fullwidth', **out)


def g0_state_dict():
    """names, shapes and parameter count of the reference XceptionVidTr().state_dict()
    (checkpoint interchange, train_CNN.py:183,999-1011)."""
    import json
    model = ref_vivit.XceptionVidTr()
    sd = model.state_dict()
    out = {'entries': [[k, list(v.shape)] for k, v in sd.items()],
           'num_parameters': sum(p.numel() for p in model.parameters())}
    path = os.path.join(HERE, 'G0_state_dict.json')
    with open(path, 'w') as f:
        json.dump(out, f)
    print('wrote', path, len(out['entries']), 'entries')


ALL = {'G0': g0_state_dict, 'G1': g1_stem, 'G2': g2_modules, 'G3': g3_layer, 'G4': g4_dsttr, 'G5': g5_end_to_end, 'G6': g6_fullwidth}


# --------------------------------------------------------------------------------------
# round 2: the fixtures SURVEY 8(c) lists that round 1 skipped, float64 references, Xception blocks / full network,
# the sibling attention variants.  New files only: G1-G6 above stay byte-identical.
# --------------------------------------------------------------------------------------
def g1b_stem224():
    """G1 at 224^2 (xception.py:193-206 at the benchmark geometry: 14x14 output)."""
    out = {}
    side = 224
    net = ref_xception.xception(pretrained=False)
    load_recipe(net, 'xcep.model.')
    net.train()
    x = t(recipe.input_value('g1.x%d' % side, (2, 3, side, side))).requires_grad_(True)
    y = net.low_level_features(x)
    coef = t(recipe.input_value('g1.coef%d' % side, tuple(y.shape)))
    (y * coef).sum().backward()
    tag = 's%d.' % side
    out[tag + 'y'] = npy(y)
    out[tag + 'dx_norm'] = npy(x.grad.norm())
    out[tag + 'dx_slice'] = npy(x.grad[0, :, 10:14, 20:28])
    sd = net.state_dict()
    for k in ('bn1', 'bn2', 'block1.skipbn', 'block2.rep.2', 'block3.rep.5', 'block3.skipbn'):
        out[tag + k + '.running_mean'] = npy(sd[k + '.running_mean'])
        out[tag + k + '.running_var'] = npy(sd[k + '.running_var'])
    for k, p in net.named_parameters():
        if p.grad is not None:
            out[tag + 'gnorm.' + k] = npy(p.grad.norm())
    for k in ('conv1.weight', 'block2.rep.4.conv1.weight', 'block3.skip.weight'):
        out[tag + 'grad.' + k] = npy(dict(net.named_parameters())[k].grad.reshape(-1)[:4096])
    save('G1b_stem224', **out)


def g2b_modules_f17():
    """G2 at F = 17 frames (T = 16: the C4 temporal tile; module.py:174-208 with 17 keys per softmax)."""
    out = {}
    dim, heads, dh = MOD_CFG['dim'], MOD_CFG['heads'], MOD_CFG['dim_head']
    F, st = 17, 13
    shape = (1, F * 362, dim)
    mods = {
        'prenorm_ff': ref_module.PreNorm(dim, ref_module.FeedForward(dim, 4 * dim)),
        'spatial': ref_module.SpatialOnlyAttention(dim, heads=heads, dim_head=dh),
        'temporal': ref_module.TemporalResidualAttention(dim, heads=heads, dim_head=dh),
    }
    for name, mod in mods.items():
        load_recipe(mod, 'g2.%s.' % name)
        res = _run_module(mod, 'g2.%s.F%d' % (name, F), shape)
        for k, v in res.items():
            if k in ('y', 'dx'):
                v = v[:, ::st]
            out['F%d.%s.%s' % (F, name, k)] = v
    save('G2b_modules_F17', **out)


def g4b_dsttr_t16():
    out = {}
    dim, heads, dh = MOD_CFG['dim'], MOD_CFG['heads'], MOD_CFG['dim_head']
    T = 16
    mod = ref_vivit.DSTTr(19, 1, 1, T, dim=dim, depth=2, heads=heads, dim_head=dh, in_channels=dim, scale_dim=2)
    load_recipe(mod, 'g4.')
    x = t(recipe.input_value('g4.x.T%d' % T, (2, T, dim, 19, 19))).requires_grad_(True)
    y = mod(x)
    coef = t(recipe.input_value('g4.coef', tuple(y.shape)))
    (y * coef).sum().backward()
    tag = 'T%d.' % T
    out[tag + 'logits'] = npy(y)
    out[tag + 'dx_norm'] = npy(x.grad.norm())
    out[tag + 'dx_frame_norms'] = npy(x.grad.flatten(2).norm(dim=2))
    for k, p in mod.named_parameters():
        out[tag + 'gnorm.' + k] = npy(p.grad.norm())
    out[tag + 'grad.pos_embedding'] = npy(mod.pos_embedding.grad[0, :, ::37])
    out[tag + 'grad.space_token'] = npy(mod.space_token.grad)
    out[tag + 'grad.temporal_token'] = npy(mod.temporal_token.grad)
    save('G4b_dsttr_T16', **out)


def g5b_native_fp64():
    """G5 again with the reference run in float64 (same float32-valued recipe weights / input, exactly representable):
    the yardstick for "HIP error <= k x the reference's own float32 error" per gradient tensor."""
    model = ref_vivit.XceptionVidTr()
    load_recipe(model, '')
    model = model.double().train()
    x = t(recipe.input_value('g5.x', (1, 6, 3, 300, 300))).double()
    labels = torch.ones(1, dtype=torch.float64)
    logits = model(x)
    loss = nn.BCEWithLogitsLoss()(logits.view(-1), labels)
    loss.backward()
    out = {'logits64': npy(logits), 'loss64': npy(loss)}
    for k, p in model.named_parameters():
        if p.grad is not None:
            out['gnorm64.' + k] = npy(p.grad.norm())
    named = dict(model.named_parameters())
    for k in ['vit.transformer.layers.0.0.fn.to_qk.weight', 'vit.transformer.layers.11.2.fn.net.3.weight',
              'vit.transformer.layers.5.1.fn.to_qkv.weight', 'vit.pos_embedding', 'xcep.model.conv1.weight',
              'xcep.model.block3.rep.4.pointwise.weight']:
        out['grad64.' + k] = npy(named[k].grad.reshape(-1)[:64])
    save('G5b_native_fp64', **out)


def load_rand(module: nn.Module, prefix: str = ''):
    sd = module.state_dict()
    vals = recipe.rand_fill_state_dict(sd, prefix)
    module.load_state_dict({k: torch.from_numpy(v) for k, v in vals.items()})


def _block_case(out, tag, mod, prefix, shape, dtype=torch.float32):
    load_rand(mod, prefix)
    mod = mod.to(dtype).train()
    x = t(recipe.rand_input_value(prefix + 'x', shape)).to(dtype).requires_grad_(True)
    y = mod(x)
    coef = t(recipe.rand_input_value(prefix + 'coef', tuple(y.shape))).to(dtype)
    (y * coef).sum().backward()
    f64 = dtype == torch.float64
    out[tag + 'y'] = npy(y)
    out[tag + 'dx_norm'] = npy(x.grad.norm())
    if not f64:
        out[tag + 'dx'] = npy(x.grad)
    for k, p in mod.named_parameters():
        out[tag + 'gnorm.' + k] = npy(p.grad.norm())
        if not f64:
            out[tag + 'grad.' + k] = npy(p.grad.reshape(-1)[:4096])      # whole tensor when it is that small
    for k, v in mod.state_dict().items():
        if 'running' in k and not f64:
            out[tag + 'buf.' + k] = npy(v)


def g7_xception():
    """Block / SeparableConv2d by themselves (xception.py:46-49, 91-101), then the whole network: features(), logits(),
    forward() (xception.py:161-215) at 299^2, train mode, batch 2.  Pseudo-random recipe (recipe.rand_*)."""
    out = {}
    B = ref_xception.Block
    cases = [('b1', lambda: B(64, 128, 2, 2, start_with_relu=False, grow_first=True), (2, 64, 21, 21)),
             ('b2', lambda: B(128, 256, 2, 2, start_with_relu=True, grow_first=True), (2, 128, 15, 15)),
             ('b4', lambda: B(728, 728, 3, 1, start_with_relu=True, grow_first=True), (2, 728, 10, 10)),
             ('b12', lambda: B(728, 1024, 2, 2, start_with_relu=True, grow_first=False), (2, 728, 10, 10)),
             ('sep', lambda: ref_xception.SeparableConv2d(64, 128, 3, 1, 1), (2, 64, 13, 17))]
    for name, ctor, shape in cases:
        _block_case(out, name + '.', ctor(), 'g7.%s.' % name, shape)
        _block_case(out, name + '.f64.', ctor(), 'g7.%s.' % name, shape, torch.float64)
    # the whole network
    for dtype, tag in ((torch.float32, 'net.'), (torch.float64, 'net.f64.')):
        net = ref_xception.xception(pretrained=False)
        load_rand(net, 'g7.net.')
        net = net.to(dtype).train()
        x = t(recipe.rand_input_value('g7.net.x', (2, 3, 299, 299))).to(dtype).requires_grad_(True)
        feats = net.features(x)
        out[tag + 'features_sub'] = npy(feats[:, ::16]).copy()   # every 16th channel of (2, 2048, 10, 10), pre-ReLU (a copy: logits() rectifies feats in place)
        logits = net.logits(feats)                                # NB: relu is in place on feats (xception.py:209)
        coef = t(recipe.rand_input_value('g7.net.coef', tuple(logits.shape))).to(dtype)
        (logits * coef).sum().backward()
        out[tag + 'logits'] = npy(logits)
        out[tag + 'dx_norm'] = npy(x.grad.norm())
        for k, p in net.named_parameters():
            out[tag + 'gnorm.' + k] = npy(p.grad.norm())
        sd = net.state_dict()
        for k in ('bn1', 'block5.rep.2', 'block12.skipbn', 'bn3', 'bn4'):
            out[tag + 'buf.' + k + '.running_mean'] = npy(sd[k + '.running_mean'])
            out[tag + 'buf.' + k + '.running_var'] = npy(sd[k + '.running_var'])
    # eval-mode forward (running statistics as updated by the one training step above would differ per dtype: use fresh)
    net = ref_xception.xception(pretrained=False)
    load_rand(net, 'g7.net.')
    net.eval()
    with torch.no_grad():
        out['net.eval.logits'] = npy(net(t(recipe.rand_input_value('g7.net.x', (2, 3, 299, 299)))))
    save('G7_xception', **out)


def g8_siblings():
    """The ablation attention variants (module.py:36-64 Attention, :145-172 TemporalOnlyAttention; vivit.py:10-25
    Transformer, :29-81 ViViT, :150-191 VanillaTr), small widths, pseudo-random recipe, float32 + float64."""
    out = {}
    dim, heads, dh = 64, 2, 32
    cases = [('attention', lambda: ref_module.Attention(dim, heads=heads, dim_head=dh), (2, 50, dim)),
             ('temporal_only', lambda: ref_module.TemporalOnlyAttention(dim, heads=heads, dim_head=dh), (1, 5 * 362, dim)),
             ('transformer', lambda: ref_vivit.Transformer(dim, 2, heads, dh, 2 * dim), (2, 50, dim)),
             ('vivit', lambda: ref_vivit.ViViT(19, 1, 3, 4, dim=dim, depth=1, heads=heads, dim_head=dh, in_channels=dim,
                                               scale_dim=2), (2, 4, dim, 19, 19)),
             ('vivit_mean', lambda: ref_vivit.ViViT(19, 1, 3, 4, dim=dim, depth=1, heads=heads, dim_head=dh, pool='mean',
                                                    in_channels=dim, scale_dim=2), (2, 4, dim, 19, 19)),
             ('vanilla', lambda: ref_vivit.VanillaTr(7, 1, 3, 4, dim=dim, depth=1, heads=heads, dim_head=dh,
                                                     in_channels=dim, scale_dim=2), (2, 4, dim, 7, 7))]
    for name, ctor, shape in cases:
        for dtype, tag in ((torch.float32, name + '.'), (torch.float64, name + '.f64.')):
            mod = ctor()
            load_rand(mod, 'g8.%s.' % name)
            mod = mod.to(dtype).train()
            x = t(recipe.rand_input_value('g8.%s.x' % name, shape)).to(dtype).requires_grad_(True)
            y = mod(x)
            coef = t(recipe.rand_input_value('g8.%s.coef' % name, tuple(y.shape))).to(dtype)
            (y * coef).sum().backward()
            big = y.numel() > 20000
            f64 = dtype == torch.float64
            out[tag + 'y'] = npy(y)[:, ::7] if big else npy(y)
            if not f64:
                out[tag + 'dx'] = npy(x.grad)[:, ::7] if (big or x.grad.numel() > 200000) else npy(x.grad)
            out[tag + 'dx_norm'] = npy(x.grad.norm())
            for k, p in mod.named_parameters():
                out[tag + 'gnorm.' + k] = npy(p.grad.norm())
                if not f64:
                    out[tag + 'grad.' + k] = npy(p.grad.reshape(-1)[:4096])
    save('G8_siblings', **out)


def g9_dualnet_halves():
    """DualNet's Xception halves (xception_for_dualnet.py:215-284, the split points dual_net.py:210-232 calls): fea_0_7,
    fea_8_12, fea_0_4, fea_5_8, fea_9_12 of the reference's own class at 171^2, batch 2, train mode, float32 + float64;
    eval-mode forward() -> (pooled features, logits) with the classifier renamed as get_xception does (:349-351)."""
    import network.xception_for_dualnet as ref_xd
    out = {}
    shape = (2, 3, 171, 171)
    for dtype, tag in ((torch.float32, ''), (torch.float64, 'f64.')):
        f64 = dtype == torch.float64
        net = ref_xd.Xception(num_classes=1)
        load_rand(net, 'g9.net.')
        net = net.to(dtype).train()
        x = t(recipe.rand_input_value('g9.net.x', shape)).to(dtype).requires_grad_(True)
        a = net.fea_0_7(x)                                  # (2, 728, 11, 11)
        b = net.fea_8_12(a)                                 # (2, 2048, 6, 6)
        coef = t(recipe.rand_input_value('g9.net.coef', tuple(b.shape))).to(dtype)
        (b * coef).sum().backward()
        out[tag + 'fea_0_7_sub'] = npy(a[:, ::8])
        out[tag + 'fea_8_12_sub'] = npy(b[:, ::16])
        out[tag + 'dx_norm'] = npy(x.grad.norm())
        for k, p_ in net.named_parameters():
            if p_.grad is not None:
                out[tag + 'gnorm.' + k] = npy(p_.grad.norm())
        if not f64:
            sd = net.state_dict()
            for k in ('bn2', 'block7.rep.8', 'block8.rep.2', 'block12.skipbn', 'bn4'):
                out['buf.' + k + '.running_mean'] = npy(sd[k + '.running_mean'])
                out['buf.' + k + '.running_var'] = npy(sd[k + '.running_var'])
        # the three-way split on a fresh copy (fresh running statistics): same layers, same result
        net3 = ref_xd.Xception(num_classes=1)
        load_rand(net3, 'g9.net.')
        net3 = net3.to(dtype).train()
        with torch.no_grad():
            c4 = net3.fea_0_4(t(recipe.rand_input_value('g9.net.x', shape)).to(dtype))
            c8 = net3.fea_5_8(c4)
            c12 = net3.fea_9_12(c8)
        out[tag + 'fea_0_4_sub'] = npy(c4[:, ::8])
        out[tag + 'fea_5_8_sub'] = npy(c8[:, ::8])
        out[tag + 'fea_9_12_sub'] = npy(c12[:, ::16])
        # a middle piece by itself with its own input and gradient (what dual_net.py:221-222 does with the fused features)
        net5 = ref_xd.Xception(num_classes=1)
        load_rand(net5, 'g9.net.')
        net5 = net5.to(dtype).train()
        xm = t(recipe.rand_input_value('g9.mid.x', (2, 728, 11, 11))).to(dtype).requires_grad_(True)
        ym = net5.fea_8_12(xm)
        cm = t(recipe.rand_input_value('g9.mid.coef', tuple(ym.shape))).to(dtype)
        (ym * cm).sum().backward()
        out[tag + 'mid.fea_8_12_sub'] = npy(ym[:, ::16])
        out[tag + 'mid.dx_norm'] = npy(xm.grad.norm())
        if not f64:
            out['mid.dx'] = npy(xm.grad[:, ::8])
        for k in ('block8.rep.1.conv1.weight', 'block11.rep.7.pointwise.weight', 'block12.skip.weight', 'conv4.pointwise.weight', 'bn4.weight'):
            out[tag + 'mid.gnorm.' + k] = npy(dict(net5.named_parameters())[k].grad.norm())
    # eval-mode forward: (y, x) = (pooled features, last_linear(dp(y))), fc renamed as get_xception does
    net = ref_xd.Xception(num_classes=1)
    load_rand(net, 'g9.net.')
    net.last_linear = net.fc
    del net.fc
    net.eval()
    with torch.no_grad():
        y, lg = net(t(recipe.rand_input_value('g9.net.x', shape)))
    out['eval.pooled_sub'] = npy(y[:, ::16])
    out['eval.logits'] = npy(lg)
    save('G9_dualnet_halves', **out)


def load_cond(module: nn.Module, prefix: str = ''):
    sd = module.state_dict()
    vals = recipe.cond_fill_state_dict(sd, prefix)
    module.load_state_dict({k: torch.from_numpy(v) for k, v in vals.items()})


class _ScoreProbe:
    """max |score| and mean softmax entropy (relative to log F) of every TemporalResidualAttention, recomputed from the
    output of its to_qk Linear (forward hook): says in the fixture itself that no temporal softmax is saturated"""

    def __init__(self, model, tokens_per_frame):
        self.rows = []
        self.p = tokens_per_frame
        for name, m in model.named_modules():
            if isinstance(m, ref_module.TemporalResidualAttention):
                m.to_qk.register_forward_hook(self._hook(name, m.heads, m.scale))

    def _hook(self, name, heads, scale):
        def fn(_mod, _inp, out):
            with torch.no_grad():
                q, k = out.double().chunk(2, dim=-1)
                b, n, hd = q.shape
                f = n // self.p
                q = q.view(b, f, self.p, heads, hd // heads).permute(0, 3, 2, 1, 4)
                k = k.view(b, f, self.p, heads, hd // heads).permute(0, 3, 2, 1, 4)
                s = torch.einsum('...id,...jd->...ij', q, k) * scale
                pr = s.softmax(-1)
                ent = -(pr * pr.clamp_min(1e-300).log()).sum(-1).mean() / np.log(f)
                self.rows.append((name, float(s.abs().max()), float(ent)))
        return fn

    def summary(self):
        return (np.array([r[0] for r in self.rows]), np.array([r[1] for r in self.rows], dtype=np.float64),
                np.array([r[2] for r in self.rows], dtype=np.float64))


def _capture_grads(model, out, suffix, store_sub):
    for k, p in model.named_parameters():
        if p.grad is None:
            continue
        out['gnorm%s.%s' % (suffix, k)] = npy(p.grad.double().norm())
        if store_sub:
            g = p.grad.reshape(-1)
            idx = torch.from_numpy(recipe.grad_subsample_index(g.numel()))
            out['gsub64.' + k] = npy(g[idx]).astype(np.float32)


def g5c_native_conditioned():
    """G5 for bfloat16 gradient parity: the reference XceptionVidTr() on a clip of CORRELATED frames
    (base + 0.3 x noise per frame: what a face video is, and the hard case for the frame-differenced q / k operands) with
    the well-conditioned recipe (recipe.cond_param_value): logit, loss, the norm of every live gradient in float32 AND
    float64, and 4096 evenly spaced entries of every float64 gradient tensor (cosine / direction checks)."""
    out = {}
    x32 = t(recipe.correlated_frames('g5c.x', (1, 6, 3, 300, 300)))
    for dt, suffix in ((torch.float64, '64'), (torch.float32, '')):
        model = ref_vivit.XceptionVidTr()
        load_cond(model, '')
        model = model.to(dt).train()
        probe = _ScoreProbe(model, 362) if dt == torch.float64 else None
        logits = model(x32.to(dt))
        loss = nn.BCEWithLogitsLoss()(logits.view(-1), torch.ones(1, dtype=dt))
        loss.backward()
        out['logits' + suffix] = npy(logits)
        out['loss' + suffix] = npy(loss)
        _capture_grads(model, out, suffix, store_sub=(dt == torch.float64))
        if probe is not None:
            out['probe.names'], out['probe.max_abs_score'], out['probe.rel_entropy'] = probe.summary()
            print('temporal scores: max |s| %.2f, min relative entropy %.3f' % (out['probe.max_abs_score'].max(),
                                                                                out['probe.rel_entropy'].min()))
        if dt == torch.float32:
            out['live_param_names'] = np.array([k for k, p in model.named_parameters() if p.grad is not None])
    save('G5c_native_conditioned', **out)


def g6c_fullwidth_conditioned():
    """G6 likewise: DSTTr(19, 1, 1, 8, depth=2) at full width (dim 728) on correlated feature frames, well-conditioned
    recipe; logit, input-gradient subsample, every parameter-gradient norm (f32, f64) + float64 subsamples."""
    out = {}
    x32 = recipe.correlated_frames('g6c.x', (1, 8, 728, 19, 19))
    for dt, suffix in ((torch.float64, '64'), (torch.float32, '')):
        mod = ref_vivit.DSTTr(19, 1, 1, 8, depth=2)
        load_cond(mod, 'vit.')
        mod = mod.to(dt)
        probe = _ScoreProbe(mod, 362) if dt == torch.float64 else None
        x = t(x32).to(dt).requires_grad_(True)
        y = mod(x)
        y.sum().backward()
        out['logits' + suffix] = npy(y)
        out['dx_norm' + suffix] = npy(x.grad.double().norm())
        _capture_grads(mod, out, suffix, store_sub=(dt == torch.float64))
        if dt == torch.float64:
            g = x.grad.reshape(-1)
            out['dxsub64'] = npy(g[torch.from_numpy(recipe.grad_subsample_index(g.numel(), 16384))]).astype(np.float32)
            out['probe.names'], out['probe.max_abs_score'], out['probe.rel_entropy'] = probe.summary()
            print('temporal scores: max |s| %.2f, min relative entropy %.3f' % (out['probe.max_abs_score'].max(),
                                                                                out['probe.rel_entropy'].min()))
    save('G6c_fullwidth_conditioned', **out)


ALL.update({'G5c': g5c_native_conditioned, 'G6c': g6c_fullwidth_conditioned})
ALL.update({'G1b': g1b_stem224, 'G2b': g2b_modules_f17, 'G4b': g4b_dsttr_t16, 'G5b': g5b_native_fp64, 'G7': g7_xception,
            'G8': g8_siblings, 'G9': g9_dualnet_halves})

if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--only', default='')
    a = ap.parse_args()
    todo = [s for s in a.only.split(',') if s] or list(ALL)
    for k in todo:
        print('==', k)
        ALL[k]()
