"""Xception classes with the reference's constructor signatures and state-dict names
(reference: network/xception.py): SeparableConv2d (:39-49), Block (:52-101), Xception
(:104-220), factories xception() (:386-405) and return_pytorch04_xception() (:422-442).

The modules are parameter containers: the ISTVT hot path only runs
``Xception.low_level_features`` (conv1 .. block3), which executes as one fused HIP pipeline
(istvt_amd.stem) over channels-last activations.  The remaining blocks (block4-12, conv3/4,
fc) are constructed so reference checkpoints load and save unchanged, but -- exactly as in the
reference -- they are never executed on this path.
"""
import torch
import torch.nn as nn

from istvt_amd import stem as _stem

__all__ = ['SeparableConv2d', 'Block', 'Xception', 'xception', 'return_pytorch04_xception']


class SeparableConv2d(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size=1, stride=1, padding=0, dilation=1, bias=False):
        super(SeparableConv2d, self).__init__()
        self.conv1 = nn.Conv2d(in_channels, in_channels, kernel_size, stride, padding, dilation,
                               groups=in_channels, bias=bias)
        self.pointwise = nn.Conv2d(in_channels, out_channels, 1, 1, 0, 1, 1, bias=bias)


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

    def features(self, input):
        raise NotImplementedError('Xception exit flow (block4..conv4) is outside the ISTVT hot path (SURVEY.md 8(f)-3)')

    def forward(self, input):
        raise NotImplementedError('Xception classifier forward is outside the ISTVT hot path (SURVEY.md 8(f)-3)')


def xception(num_classes=1000, pretrained='imagenet'):
    model = Xception(num_classes=num_classes)
    if pretrained:
        raise RuntimeError('pretrained Xception weights (xception-b5690688.pth) are not available offline; '
                           'load a state_dict explicitly')
    model.last_linear = model.fc
    del model.fc
    return model


def return_pytorch04_xception(pretrained=False, weights_path=None):
    """pretrained=True loads `weights_path` the way the reference does (xception.py:424-438):
    pointwise weights stored 2-D are unsqueezed to (Cout, Cin, 1, 1)."""
    model = xception(pretrained=False)
    if pretrained:
        if weights_path is None:
            raise RuntimeError('return_pytorch04_xception(pretrained=True) needs weights_path '
                               '(the reference hard-codes /mnt/data/DFD/xception-b5690688.pth)')
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
