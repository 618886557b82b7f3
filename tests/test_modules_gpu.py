"""HIP nn.Modules vs the golden vectors captured from the reference (G2, G3, G4, G6) and vs
the oracle at geometries the reference cannot run (14x14 / 6x6 grids).  float32 parity mode:
logits rtol 1e-3 (north_star), gradients 1e-2 on norms; observed errors are ~1e-5."""
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


def load_recipe(mod, prefix):
    sd = mod.state_dict()
    mod.load_state_dict({k: torch.from_numpy(recipe.param_value(prefix + k, tuple(v.shape))) for k, v in sd.items()})
    return mod.cuda()


DIM, HEADS, DH = 64, 2, 32
ROW_STRIDE = {5: 3, 9: 7}
TOL32 = 1e-4


def _run(mod, x_name, shape):
    x = torch.from_numpy(recipe.input_value(x_name, shape)).cuda().requires_grad_(True)
    y = mod(x)
    coef = torch.from_numpy(recipe.input_value(x_name + '.coef', tuple(y.shape))).cuda()
    (y * coef).sum().backward()
    return x, y


@pytest.mark.parametrize('frames', [5, 9])
@pytest.mark.parametrize('name', ['prenorm_ff', 'ff', 'spatial', 'temporal'])
def test_g2_modules_hip(golden_dir, name, frames):
    M, V = _mods()
    g = np.load(os.path.join(golden_dir, 'G2_modules.npz'))
    mod = {'prenorm_ff': lambda: M.PreNorm(DIM, M.FeedForward(DIM, 4 * DIM)),
           'ff': lambda: M.FeedForward(DIM, 4 * DIM),
           'spatial': lambda: M.SpatialOnlyAttention(DIM, heads=HEADS, dim_head=DH),
           'temporal': lambda: M.TemporalResidualAttention(DIM, heads=HEADS, dim_head=DH)}[name]()
    load_recipe(mod, 'g2.%s.' % name)
    x, y = _run(mod, 'g2.%s.F%d' % (name, frames), (1, frames * 362, DIM))
    tag = 'F%d.%s.' % (frames, name)
    st = ROW_STRIDE[frames]
    assert relerr(y[:, ::st], g[tag + 'y']) < TOL32
    assert relerr(x.grad[:, ::st], g[tag + 'dx']) < TOL32
    for k, p in mod.named_parameters():
        assert relerr(p.grad, g[tag + 'grad.' + k]) < TOL32, k


@pytest.mark.parametrize('name', ['prenorm_ff', 'spatial', 'temporal'])
def test_g2b_modules_f17_hip(golden_dir, name):
    """F = 17 frames (T = 16, BASELINE configs[3]) DIRECTLY against the reference capture G2b: the two-tile path of the
    temporal kernels (frames 16 .. 31 live in a second 16-row tile; the frame difference and its adjoint cross the tile
    boundary at frame 15 / 16) and 17-frame spatial attention, float32."""
    M, V = _mods()
    g = np.load(os.path.join(golden_dir, 'G2b_modules_F17.npz'))
    mod = {'prenorm_ff': lambda: M.PreNorm(DIM, M.FeedForward(DIM, 4 * DIM)),
           'spatial': lambda: M.SpatialOnlyAttention(DIM, heads=HEADS, dim_head=DH),
           'temporal': lambda: M.TemporalResidualAttention(DIM, heads=HEADS, dim_head=DH)}[name]()
    load_recipe(mod, 'g2.%s.' % name)
    x, y = _run(mod, 'g2.%s.F17' % name, (1, 17 * 362, DIM))
    tag = 'F17.%s.' % name
    assert relerr(y[:, ::13], g[tag + 'y']) < TOL32
    assert relerr(x.grad[:, ::13], g[tag + 'dx']) < TOL32
    for k, p in mod.named_parameters():
        assert relerr(p.grad, g[tag + 'grad.' + k]) < TOL32, k


def test_g2b_temporal_f17_bf16_tracks_reference(golden_dir):
    """the same F = 17 module in bfloat16 (the MFMA temporal kernels, two tiles) against the reference's float32 capture.
    The recipe weights saturate these softmaxes (scores ~ 50), so bf16 rounding of q, k moves the to_qk gradient by
    several per cent whichever way the difference is taken: measured on MI355X, differencing the LayerNorm output before
    the projection (round 2) y 1.32e-2 / dx 4.44e-2 / to_qk 6.68e-2 / to_out.bias 1.15e-1, differencing q and k in the
    kernels (now) 1.33e-2 / 4.45e-2 / 6.69e-2 / 1.15e-1 -- the in-kernel difference costs no accuracy.  The kernels
    themselves are held to 4e-2 against float64 on the same bf16 inputs (gpu_checks.attn_temporal, diff + packed)."""
    M, V = _mods()
    g = np.load(os.path.join(golden_dir, 'G2b_modules_F17.npz'))
    mod = load_recipe(M.TemporalResidualAttention(DIM, heads=HEADS, dim_head=DH), 'g2.temporal.')
    x = torch.from_numpy(recipe.input_value('g2.temporal.F17', (1, 17 * 362, DIM))).cuda().bfloat16().requires_grad_(True)
    y = mod(x)
    coef = torch.from_numpy(recipe.input_value('g2.temporal.F17.coef', tuple(y.shape))).cuda().bfloat16()
    (y * coef).sum().backward()
    tag = 'F17.temporal.'
    assert relerr(y[:, ::13].float(), g[tag + 'y']) < 3e-2
    assert relerr(x.grad[:, ::13].float(), g[tag + 'dx']) < 7e-2
    bound = {'to_qk.weight': 1e-1, 'to_v.weight': 3e-2, 'to_out.0.weight': 2e-3, 'to_out.0.bias': 1.8e-1}
    for k, p in mod.named_parameters():
        assert relerr(p.grad, g[tag + 'grad.' + k]) < bound[k], k


def test_g3_sttransformer_hip(golden_dir):
    M, V = _mods()
    g = np.load(os.path.join(golden_dir, 'G3_sttransformer.npz'))
    mod = load_recipe(V.STTransformer(DIM, 2, HEADS, DH, 2 * DIM), 'g3.')
    x, y = _run(mod, 'g3.x', (1, 5 * 362, DIM))
    assert relerr(y[:, ::3], g['y']) < TOL32
    assert relerr(x.grad[:, ::3], g['dx']) < TOL32
    for k, p in mod.named_parameters():
        assert relerr(p.grad, g['grad.' + k]) < 2e-4, k


@pytest.mark.parametrize('T', [4, 8])
def test_g4_dsttr_hip(golden_dir, T):
    M, V = _mods()
    g = np.load(os.path.join(golden_dir, 'G4_dsttr.npz'))
    mod = load_recipe(V.DSTTr(19, 1, 1, T, dim=DIM, depth=2, heads=HEADS, dim_head=DH, in_channels=DIM, scale_dim=2), 'g4.')
    x = torch.from_numpy(recipe.input_value('g4.x.T%d' % T, (2, T, DIM, 19, 19))).cuda().requires_grad_(True)
    y = mod(x)
    coef = torch.from_numpy(recipe.input_value('g4.coef', tuple(y.shape))).cuda()
    (y * coef).sum().backward()
    tag = 'T%d.' % T
    assert relerr(y, g[tag + 'logits']) < 1e-4          # north_star: logits rtol 1e-3
    assert relerr(x.grad.flatten(2).norm(dim=2), g[tag + 'dx_frame_norms']) < 1e-3
    for k, p in mod.named_parameters():
        assert relerr(p.grad.norm(), g[tag + 'gnorm.' + k]) < 1e-3, k
    assert relerr(mod.pos_embedding.grad[0, :, ::37], g[tag + 'grad.pos_embedding']) < 1e-3
    assert relerr(mod.space_token.grad, g[tag + 'grad.space_token']) < 1e-3
    assert relerr(mod.temporal_token.grad, g[tag + 'grad.temporal_token']) < 1e-3


def test_g4b_dsttr_t16_hip(golden_dir):
    """DSTTr with T = 16 (F = 17, the C4 geometry) DIRECTLY against the reference capture G4b, float32"""
    M, V = _mods()
    g = np.load(os.path.join(golden_dir, 'G4b_dsttr_T16.npz'))
    T = 16
    mod = load_recipe(V.DSTTr(19, 1, 1, T, dim=DIM, depth=2, heads=HEADS, dim_head=DH, in_channels=DIM, scale_dim=2), 'g4.')
    x = torch.from_numpy(recipe.input_value('g4.x.T%d' % T, (2, T, DIM, 19, 19))).cuda().requires_grad_(True)
    y = mod(x)
    coef = torch.from_numpy(recipe.input_value('g4.coef', tuple(y.shape))).cuda()
    (y * coef).sum().backward()
    tag = 'T%d.' % T
    assert relerr(y, g[tag + 'logits']) < 1e-4
    assert relerr(x.grad.flatten(2).norm(dim=2), g[tag + 'dx_frame_norms']) < 1e-3
    for k, p in mod.named_parameters():
        assert relerr(p.grad.norm(), g[tag + 'gnorm.' + k]) < 1e-3, k
    assert relerr(mod.pos_embedding.grad[0, :, ::37], g[tag + 'grad.pos_embedding']) < 1e-3
    assert relerr(mod.space_token.grad, g[tag + 'grad.space_token']) < 1e-3
    assert relerr(mod.temporal_token.grad, g[tag + 'grad.temporal_token']) < 1e-3


def test_g6_fullwidth_hip(golden_dir):
    """dim 728 / 8 heads x 64 at the reference-native 19x19 grid (pins the K=728 tiling)."""
    M, V = _mods()
    g = np.load(os.path.join(golden_dir, 'G6_fullwidth.npz'))
    mod = load_recipe(V.DSTTr(19, 1, 1, 8, depth=2), 'vit.')
    x = torch.from_numpy(recipe.input_value('g6.x', (1, 8, 728, 19, 19))).cuda().requires_grad_(True)
    y = mod(x)
    y.sum().backward()
    assert relerr(y, g['logits']) < 1e-4
    assert relerr(x.grad.flatten(2).norm(dim=2), g['dx_frame_norms']) < 1e-3
    for k, p in mod.named_parameters():
        assert relerr(p.grad.norm(), g['gnorm.' + k]) < 1e-3, k
    qk0 = dict(mod.named_parameters())['transformer.layers.0.0.fn.to_qk.weight'].grad
    assert relerr(qk0.reshape(-1)[:256], g['grad.qk0']) < 1e-3


def test_wrong_geometry_raises():
    """reference failure modes survive: wrong T (vivit.py:138) and indivisible token count (module.py:84)."""
    M, V = _mods()
    mod = V.DSTTr(6, 1, 1, 4, dim=DIM, depth=1, heads=HEADS, dim_head=DH, in_channels=DIM).cuda()
    with pytest.raises(RuntimeError):
        mod(torch.zeros(1, 5, DIM, 6, 6, device='cuda'))
    att = M.SpatialOnlyAttention(DIM, heads=HEADS, dim_head=DH).cuda()
    with pytest.raises(RuntimeError):
        att(torch.zeros(1, 5 * 37, DIM, device='cuda'))
    with pytest.raises(RuntimeError):           # CPU tensors are refused: no fallback path
        M.FeedForward(DIM, 2 * DIM)(torch.zeros(1, 4, DIM))


WQK_CORRELATED_TOL = 0.45         # measured 0.384 (the round-3 data flow: 0.481); DESIGN.md section 4


def test_temporal_bf16_correlated_frames_tracks_float32():
    """consecutive frames of a face video are strongly correlated, so the frame difference of module.py:193
    cancels most of q and k.  PreNorm(TemporalResidualAttention) in bfloat16 must keep the precision of the DIFFERENCE:
    the LayerNorm kernel differences in fp32 before rounding and the q | k column tiles of the one GEMM read that plane
    (round 4).  Checked against the float32 oracle on x[f] = x[f-1] + 0.05 * noise at the model's width, forward and
    input gradient; the round-3 path (un-differenced operand, bf16 q / k differenced in the kernels: what calling the
    attention WITHOUT the PreNorm hook still runs) is measured beside it and must be clearly worse on this input."""
    M, V = _mods()
    from oracle import istvt_ref as R
    torch.manual_seed(0)
    dim, heads, dh, B, F, P = 728, 8, 64, 2, 9, 197
    pre = M.PreNorm(dim, M.TemporalResidualAttention(dim, heads=heads, dim_head=dh, hw=P))
    with torch.no_grad():
        for name, prm in pre.named_parameters():
            if prm.dim() == 2:
                # to_qk large enough that the scores of the DIFFERENCED rows are O(1) (a trained model's regime: the
                # temporal softmax is not uniform); q, k of the un-differenced rows are then ~14x larger
                prm.copy_(torch.randn_like(prm) * (0.5 if 'to_qk' in name else 0.05))
    pre = pre.cuda()
    base = torch.randn(B, 1, P, dim)
    frames = [base]
    for _ in range(F - 1):
        frames.append(frames[-1] + 0.05 * torch.randn(B, 1, P, dim))
    x32 = torch.cat(frames, 1).reshape(B, F * P, dim)
    coef = torch.randn(B, F * P, dim)

    p = {'n.weight': pre.norm.weight.detach().cpu(), 'n.bias': pre.norm.bias.detach().cpu()}
    p.update({'a.' + k: v.detach().cpu() for k, v in pre.fn.state_dict().items()})
    xr = x32.clone().double().requires_grad_(True)
    pd = {k: v.double() for k, v in p.items()}
    for k in ('a.to_qk.weight', 'a.to_v.weight'):
        pd[k].requires_grad_(True)
    ref = R.temporal_residual_attention(pd, 'a', R.layer_norm(pd, 'n', xr), P, heads)
    (ref * coef.double()).sum().backward()

    def run(hook):
        x = x32.cuda().bfloat16().requires_grad_(True)
        if hook:
            y = pre(x, hw=P)
        else:       # the round-3 data flow: plain LayerNorm, attention differences bf16 q / k itself
            from istvt_amd import functional as Fn
            y = pre.fn(Fn.layer_norm(x, pre.norm.weight, pre.norm.bias, pre.norm.eps), hw=P)
        for prm in pre.parameters():
            prm.grad = None
        (y * coef.cuda().bfloat16()).sum().backward()
        torch.cuda.synchronize()
        return (relerr(y.float(), ref), relerr(x.grad.float(), xr.grad),
                relerr(pre.fn.to_qk.weight.grad, pd['a.to_qk.weight'].grad), relerr(pre.fn.to_v.weight.grad, pd['a.to_v.weight'].grad))

    assert pre.fn.frame_diff_geometry(x32.cuda().bfloat16(), P) == (B, F, P)
    new_y, new_dx, new_wqk, new_wv = run(True)
    old_y, old_dx, old_wqk, old_wv = run(False)
    print('correlated frames (5 %%): y err new %.3e old %.3e; dx err new %.3e old %.3e; dW(to_qk) new %.3e old %.3e; dW(to_v) '
          'new %.3e old %.3e' % (new_y, old_y, new_dx, old_dx, new_wqk, old_wqk, new_wv, old_wv))
    # the to_qk weight gradient is formed as dq^T y from the ADJOINTED dq and the un-differenced y (exactly
    # dq'^T diff, but every product dq_f y_f is rounded at |y|, not |diff|): measured here, bounded below
    # Measured: dW(to_v) 5.1e-3 (old 6.9e-3); dW(to_qk) 0.384 (old 0.481) beside dx 0.217 (0.345): the to_qk gradient sits on
    # the same dS = p (dp - delta) floor as dx (dq' itself is that inaccurate at a 5 % frame change); the |y| / |diff|
    # amplification of forming it as dq^T y is 2^-9 x 14 ~ 3 % per operand here, an order below that floor.  At the 30 %
    # frame change of the reference capture G5c every to_qk gradient is within 1.2 % in norm and 0.9998 in direction.
    assert new_wv < 2e-2, new_wv
    assert new_wqk < WQK_CORRELATED_TOL and new_wqk < old_wqk, (new_wqk, old_wqk)
    # Measured on MI355X: y 5.2e-3 (round-3 path 7.1e-3), dx 0.22 (0.35).  The output moves little either way -- with
    # correlated frames the rows of V are nearly equal, so P V hardly depends on P -- but the input gradient does: it goes
    # through dS = p (dp - delta), where the same correlation makes dp - delta a difference of nearly equal numbers built
    # from bf16 V (any bfloat16 attention has that floor); the differenced-operand path removes the q / k share of it.
    assert new_y < 8e-3 and new_y <= old_y * 1.02, (new_y, old_y)
    assert new_dx < 0.75 * old_dx, (new_dx, old_dx)


@pytest.mark.gpu
def test_operand_refresh_spares_operands_held_by_a_live_graph():
    """ops.refresh_stale_operands() re-casts the cached bf16 operand copies (W and W^T) IN PLACE after an
    optimizer step -- but not those a live autograd graph still holds (forward A -> step -> forward B -> backward A must see
    A's weights): such copies are dropped from the caches and left intact; the steady-state loop keeps the one grouped
    launch; a dead parameter releases its copies at once."""
    import istvt_pkg
    istvt_pkg.load()
    from istvt_amd import ops
    w = torch.nn.Parameter(torch.randn(256, 728, device='cuda'))
    op = ops.weight_as(w, torch.bfloat16, pad=True)
    opt = ops._transposed_operand(op)
    key = (id(w), 'p')
    assert key in ops._operands and ops._operands[key][1] is op and ops._operands[key][2] is opt
    before = op.clone()
    with torch.no_grad():
        w.add_(1.0)                                    # "optimizer step": bumps _version
    del op, opt
    assert ops.refresh_stale_operands() == 1           # steady state: rewritten in place by the grouped launch
    op = ops.weight_as(w, torch.bfloat16, pad=True)
    assert ops._operands[key][1] is op and not torch.equal(op, before)
    assert torch.equal(op, w.detach().to(torch.bfloat16))

    class Hold(torch.autograd.Function):              # a Function that keeps the operand for its backward
        @staticmethod
        def forward(ctx, x, o):
            ctx.save_for_backward(o)
            return x * 1.0

        @staticmethod
        def backward(ctx, g):
            return g, None
    x = torch.ones(3, device='cuda', requires_grad=True)
    y = Hold.apply(x, op)
    snap = op.clone()
    with torch.no_grad():
        w.add_(1.0)
    held = op
    del op
    assert ops.refresh_stale_operands() == 0 and key not in ops._operands      # not rewritten: dropped from the caches
    torch.cuda.synchronize()
    assert torch.equal(held, snap)                                              # forward A's operand is intact
    fresh = ops.weight_as(w, torch.bfloat16, pad=True)
    assert fresh is not held and torch.equal(fresh, w.detach().to(torch.bfloat16))
    y.sum().backward()
    del y, held, fresh
    # holders no Python reference count sees.  (a) ONLY a SavedVariable holds the operand (no Python name left)
    op = ops.weight_as(w, torch.bfloat16, pad=True)
    snap = op.clone()
    y = Hold.apply(x, op)
    del op
    with torch.no_grad():
        w.add_(1.0)
    assert ops.refresh_stale_operands() == 0 and key not in ops._operands
    (held,) = y.grad_fn.saved_tensors
    torch.cuda.synchronize()
    assert torch.equal(held, snap)
    y.sum().backward()
    del y, held
    # (b) ONLY a view / slice of the operand (it shares the storage) is held, and (c) only a slice of its transpose
    for which in (1, 2):
        ops.weight_as(w, torch.bfloat16, pad=True)
        part = ops._operands[key][which][:8]
        snap = part.clone()
        with torch.no_grad():
            w.add_(1.0)
        assert ops.refresh_stale_operands() == 0 and key not in ops._operands, which
        torch.cuda.synchronize()
        assert torch.equal(part, snap), which
        del part
    # ... and with nobody holding anything the in-place path is back
    ops.weight_as(w, torch.bfloat16, pad=True)
    with torch.no_grad():
        w.add_(1.0)
    assert ops.refresh_stale_operands() == 1 and torch.equal(ops._operands[key][1], w.detach().to(torch.bfloat16))
    n0 = len(ops._operands)
    assert key in ops._operands
    del w
    import gc
    gc.collect()
    assert key not in ops._operands and len(ops._operands) == n0 - 1            # the parameter died: copies released



def test_gelu_saved_derivative_is_bit_identical_in_float32_and_close_in_bfloat16():
    """FeedForward (module.py:23-34) with the GELU pair exchanging gelu'(u) instead of u (istvt_gemm flags bit 4, the default)
    against the u form: float32 gives the same bits (the same gelu'(fp32 u), computed one kernel earlier); bfloat16 differs
    by the rounding of the derivative (computed from the fp32 pre-activation now, from its bf16 rounding before)."""
    import istvt_pkg
    istvt_pkg.load()
    from istvt_amd import functional as Fn
    g = torch.Generator().manual_seed(3)
    for dt, tol in ((torch.float32, 0.0), (torch.bfloat16, 8e-3)):
        res = []
        for on in (True, False):
            Fn.GELU_SAVE_DERIV[0] = on
            try:
                x = (torch.randn((700, 728), generator=torch.Generator().manual_seed(3)) * 0.7).cuda().to(dt).requires_grad_(True)
                w1 = (torch.randn((2912, 728), generator=torch.Generator().manual_seed(4)) * 728 ** -0.5).cuda().requires_grad_(True)
                b1 = torch.randn((2912,), generator=torch.Generator().manual_seed(5)).cuda().requires_grad_(True)
                w2 = (torch.randn((728, 2912), generator=torch.Generator().manual_seed(6)) * 2912 ** -0.5).cuda().requires_grad_(True)
                b2 = torch.randn((728,), generator=torch.Generator().manual_seed(7)).cuda().requires_grad_(True)
                y = Fn.FeedForwardFn.apply(x, w1, b1, w2, b2, None)
                dy = torch.randn((700, 728), generator=torch.Generator().manual_seed(8)).cuda().to(dt)
                y.backward(dy)
                torch.cuda.synchronize()
                res.append([t.detach().float().clone() for t in (y, x.grad, w1.grad, b1.grad, w2.grad, b2.grad)])
            finally:
                Fn.GELU_SAVE_DERIV[0] = True
        for a, b in zip(*res):
            if tol == 0.0:
                assert torch.equal(a, b)
            else:
                assert float((a - b).norm() / b.norm().clamp_min(1e-30)) < tol
