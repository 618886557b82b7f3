"""Pin the oracle (oracle/istvt_ref.py) to golden vectors G1-G6 captured from the
reference's own modules (tests/golden/make_golden.py).  CPU only."""
import os

import numpy as np
import pytest
import torch

import recipe
from oracle import istvt_ref as R

torch.set_num_threads(min(16, os.cpu_count() or 1))      # the job's CPU share on the GPU box (conftest.py); all 8 here
RTOL = 1e-5          # every tensor <= 1e-5 rel in fp32


def relerr(a, b):
    a = torch.as_tensor(np.asarray(a), dtype=torch.float64)
    b = torch.as_tensor(np.asarray(b), dtype=torch.float64)
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def recipe_params(shapes, prefix):
    return {k: torch.from_numpy(recipe.param_value(prefix + k, s)) for k, s in shapes.items()}


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + '.npz'))


@pytest.mark.parametrize('side', [96, 139])
def test_g1_stem(golden_dir, side):
    g = load(golden_dir, 'G1_stem')
    p = R.with_grad(recipe_params(R.stem_param_shapes(), 'xcep.model.'))
    x = torch.from_numpy(recipe.input_value('g1.x%d' % side, (2, 3, side, side))).requires_grad_(True)
    y = R.stem_forward(p, x)
    assert y.shape[-1] == R.stem_out_side(side)
    coef = torch.from_numpy(recipe.input_value('g1.coef%d' % side, tuple(y.shape)))
    (y * coef).sum().backward()
    tag = 's%d.' % side
    assert relerr(y.detach(), g[tag + 'y']) < RTOL
    assert relerr(x.grad.norm(), g[tag + 'dx_norm']) < 1e-4
    assert relerr(x.grad[0, :, 10:14, 20:28], g[tag + 'dx_slice']) < 1e-4
    for k in ('bn1', 'bn2', 'block1.skipbn', 'block2.rep.2', 'block3.rep.5', 'block3.skipbn'):
        assert relerr(p[k + '.running_mean'], g[tag + k + '.running_mean']) < RTOL
        assert relerr(p[k + '.running_var'], g[tag + k + '.running_var']) < RTOL
    for k, v in p.items():
        if v.requires_grad:
            assert relerr(v.grad.norm(), g[tag + 'gnorm.' + k]) < 1e-4, k
    for k in g.files:
        if k.startswith(tag + 'grad.'):
            name = k[len(tag + 'grad.'):]
            assert relerr(p[name].grad.reshape(-1)[:4096], g[k]) < 1e-4, name


def _module_case(fn, pshapes, prefix, x_name, shape):
    p = R.with_grad(recipe_params(pshapes, prefix))
    x = torch.from_numpy(recipe.input_value(x_name, shape)).requires_grad_(True)
    y = fn(p, x)
    coef = torch.from_numpy(recipe.input_value(x_name + '.coef', tuple(y.shape)))
    (y * coef).sum().backward()
    return p, x, y


DIM, HEADS, DH = 64, 2, 32
INNER = HEADS * DH
ROW_STRIDE = {5: 3, 9: 7}
MODULES = {
    'prenorm_ff': ({'norm.weight': (DIM,), 'norm.bias': (DIM,), 'fn.net.0.weight': (4 * DIM, DIM),
                    'fn.net.0.bias': (4 * DIM,), 'fn.net.3.weight': (DIM, 4 * DIM), 'fn.net.3.bias': (DIM,)},
                   lambda p, x: R.feed_forward(p, 'fn', R.layer_norm(p, 'norm', x))),
    'ff': ({'net.0.weight': (4 * DIM, DIM), 'net.0.bias': (4 * DIM,), 'net.3.weight': (DIM, 4 * DIM),
            'net.3.bias': (DIM,)},
           lambda p, x: R.feed_forward({'f.' + k: v for k, v in p.items()}, 'f', x)),
    'spatial': ({'to_qkv.weight': (3 * INNER, DIM), 'to_out.0.weight': (DIM, INNER), 'to_out.0.bias': (DIM,)},
                lambda p, x: R.spatial_attention({'a.' + k: v for k, v in p.items()}, 'a', x, 362, HEADS)),
    'temporal': ({'to_qk.weight': (2 * INNER, DIM), 'to_v.weight': (INNER, DIM), 'to_out.0.weight': (DIM, INNER),
                  'to_out.0.bias': (DIM,)},
                 lambda p, x: R.temporal_residual_attention({'a.' + k: v for k, v in p.items()}, 'a', x, 362, HEADS)),
}


@pytest.mark.parametrize('frames', [5, 9])
@pytest.mark.parametrize('name', list(MODULES))
def test_g2_modules(golden_dir, name, frames):
    g = load(golden_dir, 'G2_modules')
    shapes, fn = MODULES[name]
    p, x, y = _module_case(fn, shapes, 'g2.%s.' % name, 'g2.%s.F%d' % (name, frames), (1, frames * 362, DIM))
    tag = 'F%d.%s.' % (frames, name)
    st = ROW_STRIDE[frames]
    assert relerr(y.detach()[:, ::st], g[tag + 'y']) < RTOL
    assert relerr(x.grad[:, ::st], g[tag + 'dx']) < 2e-5
    for k, v in p.items():
        assert relerr(v.grad, g[tag + 'grad.' + k]) < 2e-5, k


def test_g3_sttransformer(golden_dir):
    g = load(golden_dir, 'G3_sttransformer')
    shapes = {k[len('transformer.'):]: v for k, v in
              R.dsttr_param_shapes(1, 1, dim=DIM, depth=2, heads=HEADS, dim_head=DH, scale_dim=2).items()
              if k.startswith('transformer.')}
    fn = lambda p, x: R.st_transformer({'t.' + k: v for k, v in p.items()}, 't', x, 2, 362, HEADS)  # noqa: E731
    p, x, y = _module_case(fn, shapes, 'g3.', 'g3.x', (1, 5 * 362, DIM))
    assert relerr(y.detach()[:, ::3], g['y']) < RTOL
    assert relerr(x.grad[:, ::3], g['dx']) < 2e-5
    for k, v in p.items():
        assert relerr(v.grad, g['grad.' + k]) < 5e-5, k


@pytest.mark.parametrize('T', [4, 8])
def test_g4_dsttr(golden_dir, T):
    g = load(golden_dir, 'G4_dsttr')
    shapes = R.dsttr_param_shapes(T, 19, dim=DIM, depth=2, heads=HEADS, dim_head=DH, scale_dim=2)
    p = R.with_grad(recipe_params(shapes, 'g4.'))
    x = torch.from_numpy(recipe.input_value('g4.x.T%d' % T, (2, T, DIM, 19, 19))).requires_grad_(True)
    y = R.dsttr_forward(p, x, depth=2, heads=HEADS)
    coef = torch.from_numpy(recipe.input_value('g4.coef', tuple(y.shape)))
    (y * coef).sum().backward()
    tag = 'T%d.' % T
    assert relerr(y.detach(), g[tag + 'logits']) < RTOL
    assert relerr(x.grad.norm(), g[tag + 'dx_norm']) < 1e-4
    assert relerr(x.grad.flatten(2).norm(dim=2), g[tag + 'dx_frame_norms']) < 1e-4
    for k, v in p.items():
        assert relerr(v.grad.norm(), g[tag + 'gnorm.' + k]) < 1e-4, k
    assert relerr(p['pos_embedding'].grad[0, :, ::37], g[tag + 'grad.pos_embedding']) < 1e-4
    assert relerr(p['space_token'].grad, g[tag + 'grad.space_token']) < 1e-4
    assert relerr(p['temporal_token'].grad, g[tag + 'grad.temporal_token']) < 1e-4


def test_g6_fullwidth(golden_dir):
    g = load(golden_dir, 'G6_fullwidth')
    shapes = R.dsttr_param_shapes(8, 19, depth=2)
    p = R.with_grad(recipe_params(shapes, 'vit.'))
    x = torch.from_numpy(recipe.input_value('g6.x', (1, 8, 728, 19, 19))).requires_grad_(True)
    y = R.dsttr_forward(p, x, depth=2)
    y.sum().backward()
    assert relerr(y.detach(), g['logits']) < RTOL
    assert relerr(x.grad.flatten(2).norm(dim=2), g['dx_frame_norms']) < 1e-4
    for k, v in p.items():
        assert relerr(v.grad.norm(), g['gnorm.' + k]) < 1e-4, k
    assert relerr(p['transformer.layers.0.0.fn.to_qk.weight'].grad.reshape(-1)[:256], g['grad.qk0']) < 1e-4


def test_g5_native_end_to_end(golden_dir):
    """Reference-native geometry (T=6, 300^2 -> 19x19, depth 12): logit, BCE loss, every
    live parameter's grad norm, and one SGD(1e-3, momentum .9) step (train_CNN.py:198-201,526-533)."""
    g = load(golden_dir, 'G5_native')
    shapes = {'xcep.model.' + k: v for k, v in R.stem_param_shapes().items()}
    shapes.update({'vit.' + k: v for k, v in R.dsttr_param_shapes(6, 19).items()})
    p = R.with_grad(recipe_params(shapes, ''))
    x = torch.from_numpy(recipe.input_value('g5.x', (1, 6, 3, 300, 300)))
    logits = R.xception_vidtr_forward(p, x)
    loss = R.bce_with_logits(logits, torch.ones(1))
    loss.backward()
    assert relerr(logits.detach(), g['logits']) < RTOL
    assert relerr(loss.detach(), g['loss']) < 1e-4
    live = [str(s) for s in g['live_param_names']]
    assert sorted(live) == sorted(k for k, v in p.items() if v.requires_grad)   # dead Xception params excluded
    for k in live:
        assert relerr(p[k].grad.norm(), g['gnorm.' + k]) < 2e-4, k
    for k in g.files:
        if k.startswith('grad.'):
            assert relerr(p[k[5:]].grad.reshape(-1)[:64], g[k]) < 2e-4, k
    # first SGD-momentum step: buf = grad ; w -= lr * buf
    for k in g.files:
        if k.startswith('after_sgd.'):
            name = k[len('after_sgd.'):]
            w = p[name].detach() - 1e-3 * p[name].grad
            assert relerr(w.reshape(-1)[:64], g[k]) < 1e-6, k
    assert relerr(p['xcep.model.bn1.running_mean'], g['bn1.running_mean']) < RTOL
    assert relerr(p['xcep.model.bn1.running_var'], g['bn1.running_var']) < RTOL
