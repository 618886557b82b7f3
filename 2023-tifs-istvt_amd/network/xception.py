"""Xception classes with the reference's constructor / forward signatures and state-dict names
(reference: network/xception.py): SeparableConv2d (:39-49), Block (:52-101), Xception
(:104-220), factories xception() (:386-405) and return_pytorch04_xception() (:422-442).

Every forward runs on the HIP kernels (no torch arithmetic): ``Xception.low_level_features`` (conv1 .. block3, the
ISTVT hot path) is one fused pipeline (istvt_amd.stem.StemFn); ``SeparableConv2d.forward``, ``Block.forward`` (any
reps / strides 1|2 / start_with_relu / grow_first), the middle and exit flow of ``features()``, ``logits()`` and
``forward()`` go through istvt_amd.xblocks.  Module inputs and outputs have the reference's (n, C, H, W) shape;
outputs are channels-last in memory (a free view of the kernels' NHWC layout), and a channels-last input costs no
copy, so chained blocks never permute.  Activations are float32 or bfloat16 (the input's dtype; ``Xception`` casts
its float32 image to ``compute_dtype``).
"""
import os

import torch
import torch.nn as nn

from istvt_amd import functional as _Fn
from istvt_amd import stem as _stem
from istvt_amd import xblocks as _xb

__all__ = ['SeparableConv2d', 'Block', 'Xception', 'xception', 'return_pytorch04_xception']


class SeparableConv2d(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size=1, stride=1, padding=0, dilation=1, bias=False):
        super(SeparableConv2d, self).__init__()
        self.conv1 = nn.Conv2d(in_channels, in_channels, kernel_size, stride, padding, dilation,
                               groups=in_channels, bias=bias)
        self.pointwise = nn.Conv2d(in_channels, out_channels, 1, 1, 0, 1, 1, bias=bias)

    def _check(self):
        c = self.conv1
        if c.kernel_size != (3, 3) or c.stride != (1, 1) or c.padding != (1, 1) or c.dilation != (1, 1) or c.bias is not None \
                or self.pointwise.bias is not None:
            raise NotImplementedError('SeparableConv2d on the HIP path is the 3x3 / stride 1 / padding 1 / no-bias form every '
                                      'Xception layer uses (network/xception.py:66,72,78,139,143)')

    def forward(self, x):
        """reference xception.py:46-49: pointwise(conv1(x)); x (n, Cin, H, W) -> (n, Cout, H, W)"""
        self._check()
        n, c, h, w = x.shape
        xn = _xb.nhwc(x)
        y = _xb.SepConvFn.apply(xn.view(n * h * w, c), self.conv1.weight, self.pointwise.weight, n, h, w)
        return _xb.nchw_view(y.view(n, h, w, -1))


class Block(nn.Module):
    def __init__(self, in_filters, out_filters, reps, strides=1, start_with_relu=True, grow_first=True):
        super(Block, self).__init__()
        if out_filters != in_filters or strides != 1:
            self.skip = nn.Conv2d(in_filters, out_filters, 1, stride=strides, bias=False)
            self.skipbn = nn.BatchNorm2d(out_filters)
        else:
            self.skip = None
        self.relu = nn.ReLU(inplace=True)
        # positional layout of `rep` defines the state-dict keys (rep.0 / rep.1 ...), keep it
        units = []
        filters = in_filters
        if grow_first:
            units.append((in_filters, out_filters))
            filters = out_filters
        units += [(filters, filters)] * (reps - 1)
        if not grow_first:
            units.append((in_filters, out_filters))
        rep = []
        for cin, cout in units:
            rep += [self.relu, SeparableConv2d(cin, cout, 3, stride=1, padding=1, bias=False), nn.BatchNorm2d(cout)]
        if not start_with_relu:
            rep = rep[1:]
        else:
            rep[0] = nn.ReLU(inplace=False)
        if strides != 1:
            rep.append(nn.MaxPool2d(3, strides, 1))
        self.rep = nn.Sequential(*rep)
        self._units = units
        self._start_with_relu = bool(start_with_relu)
        self._strides = strides

    def _chain(self):
        """(ChainSpec, parameters, BatchNorm modules) in the order xblocks.RepChainFn takes them"""
        if self._strides not in (1, 2):
            raise NotImplementedError('Block strides %r: the HIP max-pool is MaxPool2d(3, 2, 1) (every strided Xception '
                                      'block, xception.py:126-128,137)' % (self._strides,))
        seps = [m for m in self.rep if isinstance(m, SeparableConv2d)]
        bns = [m for m in self.rep if isinstance(m, nn.BatchNorm2d)]
        for m in seps:
            m._check()
        params, norms = [], []
        for sep, bn in zip(seps, bns):
            params += [sep.conv1.weight, sep.pointwise.weight, bn.weight, bn.bias]
            norms.append(bn)
        if self.skip is not None:
            params += [self.skip.weight, self.skipbn.weight, self.skipbn.bias]
            norms.append(self.skipbn)
        spec = _xb.ChainSpec(self._units, self._start_with_relu, 'pool' if self._strides == 2 else 'add',
                             self.skip is not None)
        return spec, params, norms

    def forward_nhwc(self, xn, n, h, w):
        """xn: NHWC [n*h*w, Cin] -> ([n*ho*wo, Cout], ho, wo)"""
        spec, params, norms = self._chain()
        buffers = []
        for bn in norms:
            buffers += [bn.running_mean, bn.running_var]
        y = _xb.RepChainFn.apply(xn, spec, n, h, w, self.training, buffers, *params)
        if self.training:
            for bn in norms:
                bn.num_batches_tracked.add_(1)
        if self._strides == 2:
            h, w = (h - 1) // 2 + 1, (w - 1) // 2 + 1
        return y, h, w

    def forward(self, inp):
        """reference xception.py:91-101: rep(inp) + (skipbn(skip(inp)) | inp); inp (n, Cin, H, W)"""
        n, c, h, w = inp.shape
        y, ho, wo = self.forward_nhwc(_xb.nhwc(inp).view(n * h * w, c), n, h, w)
        return _xb.nchw_view(y.view(n, ho, wo, -1))


class Xception(nn.Module):
    def __init__(self, num_classes=1000):
        super(Xception, self).__init__()
        self.num_classes = num_classes
        self.conv1 = nn.Conv2d(3, 32, 3, 2, 0, bias=False)
        self.bn1 = nn.BatchNorm2d(32)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(32, 64, 3, bias=False)
        self.bn2 = nn.BatchNorm2d(64)
        self.block1 = Block(64, 128, 2, 2, start_with_relu=False, grow_first=True)
        self.block2 = Block(128, 256, 2, 2, start_with_relu=True, grow_first=True)
        self.block3 = Block(256, 728, 2, 2, start_with_relu=True, grow_first=True)
        for i in range(4, 12):
            setattr(self, 'block%d' % i, Block(728, 728, 3, 1, start_with_relu=True, grow_first=True))
        self.block12 = Block(728, 1024, 2, 2, start_with_relu=True, grow_first=False)
        self.conv3 = SeparableConv2d(1024, 1536, 3, 1, 1)
        self.bn3 = nn.BatchNorm2d(1536)
        self.conv4 = SeparableConv2d(1536, 2048, 3, 1, 1)
        self.bn4 = nn.BatchNorm2d(2048)
        self.fc = nn.Linear(2048, num_classes)
        self.compute_dtype = torch.float32

    def low_level_features_nhwc(self, input, dtype=None):
        """(n,3,S,S) float32 -> (n,h,w,728) channels-last features in the compute dtype."""
        return _stem.stem_forward(input, self, dtype or self.compute_dtype)

    def low_level_features(self, input):
        """Reference signature (xception.py:193-206): (n,3,S,S) -> (n,728,h,w)."""
        return self.low_level_features_nhwc(input).permute(0, 3, 1, 2)

    def _blocks_nhwc(self, x, n, h, w, first, last):
        """block<first> .. block<last> on NHWC rows [n*h*w, C] -> (rows, h, w)"""
        for i in range(first, last + 1):
            x, h, w = getattr(self, 'block%d' % i).forward_nhwc(x, n, h, w)
        return x, h, w

    def _exit_nhwc(self, x, n, h, w):
        """exit flow after block12: Sep(1024->1536) BN ReLU Sep(1536->2048) BN as one chain whose last BatchNorm is
        materialised (xception.py:180-190; no ReLU after bn4)"""
        for m in (self.conv3, self.conv4):
            m._check()
        spec = _xb.ChainSpec([(self.conv3.conv1.in_channels, self.conv3.pointwise.out_channels),
                              (self.conv4.conv1.in_channels, self.conv4.pointwise.out_channels)], False, 'plain', False)
        params = [self.conv3.conv1.weight, self.conv3.pointwise.weight, self.bn3.weight, self.bn3.bias,
                  self.conv4.conv1.weight, self.conv4.pointwise.weight, self.bn4.weight, self.bn4.bias]
        buffers = [self.bn3.running_mean, self.bn3.running_var, self.bn4.running_mean, self.bn4.running_var]
        x = _xb.RepChainFn.apply(x, spec, n, h, w, self.training, buffers, *params)
        if self.training:
            self.bn3.num_batches_tracked.add_(1)
            self.bn4.num_batches_tracked.add_(1)
        return x

    def features(self, input):
        """reference xception.py:161-191: entry flow, blocks 4-12, conv3/bn3/relu/conv4/bn4 -> (n, 2048, h, w)."""
        x = self.low_level_features_nhwc(input)                    # (n, h, w, 728), compute dtype
        n, h, w, c = x.shape
        x, h, w = self._blocks_nhwc(x.view(n * h * w, c), n, h, w, 4, 12)
        x = self._exit_nhwc(x, n, h, w)
        return _xb.nchw_view(x.view(n, h, w, -1))

    def logits(self, features):
        """reference xception.py:208-215: relu, global average pool, last_linear.  (The reference's ReLU is in place on
        `features`; here `features` is left untouched.)"""
        n, c, h, w = features.shape
        x = _xb.ReluAvgPoolFn.apply(_xb.nhwc(features).view(n, h * w, c), n, h * w, True)
        head = self.last_linear if hasattr(self, 'last_linear') else self.fc
        return _apply_head(head, x).float()

    def forward(self, input):
        return self.logits(self.features(input))


def _apply_head(head, x):
    """nn.Linear or the nn.Sequential(nn.Dropout, nn.Linear) that TransferModel installs (models_copy.py:37-44),
    executed on the HIP kernels."""
    mods = list(head) if isinstance(head, nn.Sequential) else [head]
    for m in mods:
        if isinstance(m, nn.Dropout):
            x = _Fn.dropout(x, m.p, m.training)
        elif isinstance(m, nn.Linear):
            x = _Fn.LinearFn.apply(x, m.weight, m.bias, None)
        else:
            raise NotImplementedError('classifier head module %s' % type(m).__name__)
    return x


def xception(num_classes=1000, pretrained='imagenet'):
    model = Xception(num_classes=num_classes)
    if pretrained:
        raise RuntimeError('pretrained Xception weights (xception-b5690688.pth) are not available offline; '
                           'load a state_dict explicitly')
    model.last_linear = model.fc
    del model.fc
    return model


REFERENCE_WEIGHTS = '/mnt/data/DFD/xception-b5690688.pth'      # the path the reference hard-codes (xception.py:429)


def default_weights_path():
    """Where return_pytorch04_xception(pretrained=True) looks when no path is given: $ISTVT_XCEPTION_WEIGHTS, else the
    reference's own hard-coded location."""
    return os.environ.get('ISTVT_XCEPTION_WEIGHTS', REFERENCE_WEIGHTS)


def return_pytorch04_xception(pretrained=False, weights_path=None):
    """Reference signature (xception.py:422; TransferModel calls it with pretrained=True).  pretrained=True loads the ImageNet
    state dict the way the reference does (xception.py:424-438): pointwise weights stored 2-D are unsqueezed to
    (Cout, Cin, 1, 1), the classifier is ``fc`` in the file and ``last_linear`` on the returned model.  The file comes
    from `weights_path`, else $ISTVT_XCEPTION_WEIGHTS, else the reference's hard-coded path."""
    model = xception(pretrained=False)
    if pretrained:
        if weights_path is None:
            weights_path = default_weights_path()
        if not os.path.exists(weights_path):
            raise FileNotFoundError('pretrained Xception weights not found at %s (set ISTVT_XCEPTION_WEIGHTS or pass '
                                    'weights_path=; the reference reads %s)' % (weights_path, REFERENCE_WEIGHTS))
        model.fc = model.last_linear
        del model.last_linear
        state = torch.load(weights_path, map_location='cpu')
        for name, weights in state.items():
            if 'pointwise' in name and weights.dim() == 2:
                state[name] = weights.unsqueeze(-1).unsqueeze(-1)
        net_dict = model.state_dict()
        net_dict.update({k: v for k, v in state.items() if k in net_dict})
        model.load_state_dict(net_dict)
        model.last_linear = model.fc
        del model.fc
    return model


for _cls in (SeparableConv2d, Block, Xception):
    _cls._replicate_for_data_parallel = _Fn.no_data_parallel      # nn.DataParallel: see functional.no_data_parallel
