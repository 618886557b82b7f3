"""SURVEY 8(f) row 4 on a real MI355X: the ablation attention variants -- Attention, TemporalOnlyAttention
(reference network/vivit/module.py:36-64, 145-172), Transformer, ViViT, VanillaTr (network/vivit/vivit.py:10-25, 29-81,
150-191) -- as index-map variants of the two attention kernels, against golden G8 captured from the reference; plus the
HIP dropout kernel (module.py:24-33,76-79,185-188 accept dropout > 0)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import recipe  # noqa: E402


def _mods():
    import istvt_pkg
    istvt_pkg.load()
    from istvt_amd.network.vivit import module as M, vivit as V
    return M, V


def relerr(a, b):
    a = torch.as_tensor(np.asarray(a.detach().cpu() if torch.is_tensor(a) else a), dtype=torch.float64)
    b = torch.as_tensor(np.asarray(b.detach().cpu() if torch.is_tensor(b) else b), dtype=torch.float64)
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def load_rand(mod, prefix):
    sd = mod.state_dict()
    mod.load_state_dict({k: torch.from_numpy(recipe.rand_param_value(prefix + k, tuple(v.shape))) for k, v in sd.items()})
    return mod.cuda().train()


DIM, HEADS, DH = 64, 2, 32


def _case(name):
    M, V = _mods()
    return {'attention': (lambda: M.Attention(DIM, heads=HEADS, dim_head=DH), (2, 50, DIM)),
            'temporal_only': (lambda: M.TemporalOnlyAttention(DIM, heads=HEADS, dim_head=DH), (1, 5 * 362, DIM)),
            'transformer': (lambda: V.Transformer(DIM, 2, HEADS, DH, 2 * DIM), (2, 50, DIM)),
            'vivit': (lambda: V.ViViT(19, 1, 3, 4, dim=DIM, depth=1, heads=HEADS, dim_head=DH, in_channels=DIM, scale_dim=2),
                      (2, 4, DIM, 19, 19)),
            'vivit_mean': (lambda: V.ViViT(19, 1, 3, 4, dim=DIM, depth=1, heads=HEADS, dim_head=DH, pool='mean',
                                           in_channels=DIM, scale_dim=2), (2, 4, DIM, 19, 19)),
            'vanilla': (lambda: V.VanillaTr(7, 1, 3, 4, dim=DIM, depth=1, heads=HEADS, dim_head=DH, in_channels=DIM,
                                            scale_dim=2), (2, 4, DIM, 7, 7))}[name]


@pytest.mark.parametrize('name', ['attention', 'temporal_only', 'transformer', 'vivit', 'vivit_mean', 'vanilla'])
def test_g8_siblings_hip(golden_dir, name):
    g = np.load(os.path.join(golden_dir, 'G8_siblings.npz'))
    ctor, shape = _case(name)
    mod = load_rand(ctor(), 'g8.%s.' % name)
    x = torch.from_numpy(recipe.rand_input_value('g8.%s.x' % name, shape)).cuda().requires_grad_(True)
    y = mod(x)
    coef = torch.from_numpy(recipe.rand_input_value('g8.%s.coef' % name, tuple(y.shape))).cuda()
    (y.float() * coef).sum().backward()
    tag = name + '.'
    big = y.numel() > 20000
    assert relerr(y[:, ::7] if big else y, g[tag + 'y']) < 1e-4
    gx = x.grad[:, ::7] if (big or x.grad.numel() > 200000) else x.grad
    assert relerr(gx, g[tag + 'dx']) < 1e-3
    for k, p in mod.named_parameters():
        assert p.grad is not None, k
        assert relerr(p.grad.norm(), g[tag + 'gnorm.' + k]) < 1e-3, k
        assert relerr(p.grad.reshape(-1)[:4096], g[tag + 'grad.' + k]) < 2e-3, k


def test_vivit_full_width_bf16_runs():
    """dim 728 / 8 heads (the K = 728 GEMM path, line-padded rows) in bfloat16: finite outputs and gradients, logits
    close to the float32 run."""
    M, V = _mods()
    outs = []
    for dt in (torch.float32, torch.bfloat16):
        torch.manual_seed(0)
        mod = V.ViViT(14, 1, 1, 4, depth=1, compute_dtype=dt).cuda().train()
        x = torch.randn(2, 4, 728, 14, 14, generator=torch.Generator().manual_seed(1)).cuda()
        y = mod(x)
        y.sum().backward()
        assert all(torch.isfinite(p.grad).all() for p in mod.parameters())
        outs.append(y.detach())
    assert float((outs[1] - outs[0]).abs().max()) < 5e-2 * max(1.0, float(outs[0].abs().max()))


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_dropout_kernel(dtype):
    import istvt_pkg
    istvt_pkg.load()
    from istvt_amd import functional as Fn
    x = torch.randn(3000, 728, generator=torch.Generator().manual_seed(0)).cuda().to(dtype).requires_grad_(True)
    p = 0.3
    torch.manual_seed(5)
    y = Fn.dropout(x, p, True)
    torch.manual_seed(5)
    y2 = Fn.dropout(x, p, True)
    torch.manual_seed(6)
    y3 = Fn.dropout(x, p, True)
    assert torch.equal(y, y2) and not torch.equal(y, y3)            # reproducible under torch.manual_seed
    kept = (y != 0) | (x == 0)
    rate = float(kept.float().mean())
    assert abs(rate - (1 - p)) < 5e-3, rate                         # 2.2 M Bernoulli draws: sigma = 3e-4
    # columns and rows are decorrelated: per-column keep rates spread like independent draws
    col = kept.float().mean(0)
    assert float(col.std()) < 3.0 * (p * (1 - p) / 3000) ** 0.5 + 1e-3
    ref = (x.detach().float() / (1 - p)).to(dtype)
    assert float((y.detach()[kept] - ref[kept]).abs().max()) <= (0 if dtype == torch.float32 else 1e-2) * float(ref.abs().max())
    gy = torch.randn(y.shape, generator=torch.Generator().manual_seed(1)).cuda().to(dtype)
    y.backward(gy)
    gref = torch.where(kept, (gy.float() / (1 - p)).to(dtype), torch.zeros_like(gy))
    assert float((x.grad - gref).abs().max()) <= (0 if dtype == torch.float32 else 1e-2) * float(gref.abs().max())
    assert Fn.dropout(x, 0.0, True) is x and Fn.dropout(x, p, False) is x


def test_modules_with_dropout_train_and_eval():
    """dropout > 0 through FeedForward / both attention blocks / STTransformer: training runs on the HIP dropout kernel
    (finite gradients, every parameter reached), eval mode equals the dropout-free module exactly."""
    M, V = _mods()
    torch.manual_seed(0)
    ref = V.STTransformer(DIM, 2, HEADS, DH, 2 * DIM, dropout=0.0).cuda()
    drp = V.STTransformer(DIM, 2, HEADS, DH, 2 * DIM, dropout=0.2).cuda()
    drp.load_state_dict(ref.state_dict())
    x = torch.randn(2, 5 * 37, DIM, generator=torch.Generator().manual_seed(1)).cuda()
    ref.eval(); drp.eval()
    with torch.no_grad():
        assert torch.equal(ref(x, hw=37), drp(x, hw=37))
    drp.train()
    xr = x.clone().requires_grad_(True)
    y = drp(xr, hw=37)
    y.square().mean().backward()
    assert torch.isfinite(y).all() and torch.isfinite(xr.grad).all()
    for k, p in drp.named_parameters():
        assert p.grad is not None and torch.isfinite(p.grad).all() and float(p.grad.abs().max()) > 0, k
    # the expectation over masks equals the dropout-free output to first order: many draws average towards it
    ref.train()
    with torch.no_grad():
        base = ref(x, hw=37)
        acc = torch.zeros_like(base)
        for i in range(24):
            acc += drp(x, hw=37)
    assert relerr(acc / 24, base) < 0.2
