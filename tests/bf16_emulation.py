"""bf16-storage emulation of the oracle for judging the HIP path's bfloat16 mode.

In bf16 mode every activation (and activation gradient) that reaches HBM is rounded to
bfloat16 while all arithmetic, statistics and parameter gradients stay fp32.  Comparing that
against the fp32 oracle mixes two things: kernel bugs and the path's inherent rounding noise,
which a BN/ReLU chain amplifies.  This module restates the oracle's stem with a rounding
point at every place the HIP pipeline stores a tensor (forward AND backward), using plain torch
ops, so `HIP bf16  vs  emulation` isolates kernel correctness, while `emulation vs fp32 oracle`
is the precision cost of bf16 storage itself (reported, not asserted).
"""
import torch
import torch.nn.functional as F


class _RoundBF16(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).to(g.dtype)


def q(x):
    """storage point: value rounded on the way to HBM, gradient rounded on the way back"""
    return _RoundBF16.apply(x)


def qw(w):
    """GEMM weight operand: bf16 copy of the fp32 master weight (gradient stays fp32)"""
    return w + (w.detach().to(torch.bfloat16).to(w.dtype) - w.detach())


def _bn(p, name, x):
    return F.batch_norm(x, None, None, p[name + '.weight'], p[name + '.bias'], True, 0.1, 1e-5)


# ---- decisions (ReLU masks, max-pool arg-max) that can be recorded or forced ---------------------------------------
# Two bf16 pipelines with different fp32 summation orders decide a small share of the ReLU masks / pooling arg-maxes
# differently (|z| within one bf16 ulp of the kink / two window entries within one ulp of each other); every such flip
# is an O(1) change of that element's gradient.  `record` collects this emulation's own decisions, `force` replaces them
# by given ones (the HIP path's, rebuilt from its saved tensors by gpu_checks.stem_bf16_masked): with equal decisions the
# two pipelines differ by rounding only.
class _MaskedRelu(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, mask):
        ctx.save_for_backward(mask)
        return z * mask

    @staticmethod
    def backward(ctx, g):
        (mask,) = ctx.saved_tensors
        return g * mask, None


def _relu(z, site, force, record):
    if record is not None:
        record[site] = (z > 0)
    if force is not None:
        return _MaskedRelu.apply(z, force[site].to(z.dtype))
    return F.relu(z)


def _taps(z):
    """(9, n, c, Ho, Wo): the 3x3 / stride 2 / padding 1 windows of z, tap t = 3 dy + dx at (2 yo - 1 + dy, 2 xo - 1 + dx)"""
    n, c, H, W = z.shape
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    zp = F.pad(z, (1, 2, 1, 2), value=float('-inf'))
    return torch.stack([zp[:, :, dy:dy + 2 * Ho:2, dx:dx + 2 * Wo:2] for dy in range(3) for dx in range(3)], dim=0)


def _pool(z, site, force, record):
    taps = _taps(z)
    if record is not None:
        record[site] = (taps == taps.max(dim=0, keepdim=True).values).to(torch.uint8).argmax(dim=0)    # first maximum wins
    if force is not None:
        return torch.gather(taps, 0, force[site].long().unsqueeze(0)).squeeze(0)
    return F.max_pool2d(z, 3, 2, 1)


def _block(p, name, inp, relu, force=None, record=None):
    i0 = 1 if relu else 0
    a = _relu(inp, name + '.in', force, record) if relu else inp
    c = a.shape[1]
    d1 = q(F.conv2d(a, p['%s.rep.%d.conv1.weight' % (name, i0)], None, 1, 1, 1, c))
    uA = q(F.conv2d(d1, qw(p['%s.rep.%d.pointwise.weight' % (name, i0)])))
    aA = q(_relu(_bn(p, '%s.rep.%d' % (name, i0 + 1), uA), name + '.A', force, record))
    c = aA.shape[1]
    d2 = q(F.conv2d(aA, p['%s.rep.%d.conv1.weight' % (name, i0 + 3)], None, 1, 1, 1, c))
    uB = q(F.conv2d(d2, qw(p['%s.rep.%d.pointwise.weight' % (name, i0 + 3)])))
    zB = q(_bn(p, '%s.rep.%d' % (name, i0 + 4), uB))
    xs = inp[:, :, ::2, ::2]
    uS = q(F.conv2d(xs, qw(p[name + '.skip.weight'])))
    return q(_pool(zB, name + '.pool', force, record) + _bn(p, name + '.skipbn', uS))


def stem_forward_bf16(p, x, force=None, record=None):
    """p: {name: fp32 tensor (requires_grad)} on any device; x: (n,3,S,S) fp32.  force / record: see above."""
    u1 = q(F.conv2d(q(x), qw(p['conv1.weight']), None, 2, 0))
    a1 = q(_relu(_bn(p, 'bn1', u1), 'bn1', force, record))
    u2 = q(F.conv2d(a1, qw(p['conv2.weight'])))
    a2 = q(_relu(_bn(p, 'bn2', u2), 'bn2', force, record))
    t = _block(p, 'block1', a2, False, force, record)
    t = _block(p, 'block2', t, True, force, record)
    return _block(p, 'block3', t, True, force, record)
