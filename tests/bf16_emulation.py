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


def _block(p, name, inp, relu):
    i0 = 1 if relu else 0
    a = F.relu(inp) if relu else inp
    c = a.shape[1]
    d1 = q(F.conv2d(a, p['%s.rep.%d.conv1.weight' % (name, i0)], None, 1, 1, 1, c))
    uA = q(F.conv2d(d1, qw(p['%s.rep.%d.pointwise.weight' % (name, i0)])))
    aA = q(F.relu(_bn(p, '%s.rep.%d' % (name, i0 + 1), uA)))
    c = aA.shape[1]
    d2 = q(F.conv2d(aA, p['%s.rep.%d.conv1.weight' % (name, i0 + 3)], None, 1, 1, 1, c))
    uB = q(F.conv2d(d2, qw(p['%s.rep.%d.pointwise.weight' % (name, i0 + 3)])))
    zB = q(_bn(p, '%s.rep.%d' % (name, i0 + 4), uB))
    xs = inp[:, :, ::2, ::2]
    uS = q(F.conv2d(xs, qw(p[name + '.skip.weight'])))
    return q(F.max_pool2d(zB, 3, 2, 1) + _bn(p, name + '.skipbn', uS))


def stem_forward_bf16(p, x):
    """p: {name: fp32 tensor (requires_grad)} on any device; x: (n,3,S,S) fp32."""
    u1 = q(F.conv2d(q(x), qw(p['conv1.weight']), None, 2, 0))
    a1 = q(F.relu(_bn(p, 'bn1', u1)))
    u2 = q(F.conv2d(a1, qw(p['conv2.weight'])))
    a2 = q(F.relu(_bn(p, 'bn2', u2)))
    t = _block(p, 'block1', a2, False)
    t = _block(p, 'block2', t, True)
    return _block(p, 'block3', t, True)
