"""The Xception of the reference's DualNet (reference: network/xception_for_dualnet.py): the same layers and state-dict
names as network/xception.py's Xception plus the split points DualNet calls (dual_net.py:210-232):

    fea_0_7  (:215-231)  conv1 .. block7          fea_0_4  (:248-262)  conv1 .. block4
    fea_8_12 (:233-246)  block8 .. bn4            fea_5_8  (:264-270)  block5 .. block8
                                                  fea_9_12 (:272-284)  block9 .. bn4

``logits`` / ``forward`` return the pair (pooled features, head output) with a Dropout(0.2) in front of the head
(:317-329).  Every piece runs on the HIP kernels through the generic block chain (istvt_amd.xblocks.RepChainFn);
inputs and outputs have the reference's (n, C, H, W) shape, channels-last in memory, so chained pieces never permute.
``DualNet`` itself (FAD / LFS frequency heads, MixBlock) and ``ClassBlock`` are out of scope: they need the
un-vendored ``attention_lib`` / ``perceiver_pytorch``.
"""
import torch.nn as nn

from istvt_amd import functional as _Fn
from istvt_amd import xblocks as _xb
from . import xception as _x
from .xception import Block, SeparableConv2d  # noqa: F401  (the reference module defines them too)

__all__ = ['SeparableConv2d', 'Block', 'Xception', 'get_xception']


class Xception(_x.Xception):
    def __init__(self, num_classes=1):
        super(Xception, self).__init__(num_classes=num_classes)
        self.dp = nn.Dropout(p=0.2)

    # ---- split points ------------------------------------------------------------------------------------------
    def _entry_to(self, input, last):
        x = self.low_level_features_nhwc(input)                    # conv1 .. block3: (n, h, w, 728), compute dtype
        n, h, w, c = x.shape
        x, h, w = self._blocks_nhwc(x.view(n * h * w, c), n, h, w, 4, last)
        return _xb.nchw_view(x.view(n, h, w, -1))

    def _middle(self, x, first, last, exit_flow):
        n, c, h, w = x.shape
        if x.dtype != self.compute_dtype:
            x = x.to(self.compute_dtype)
        y, h, w = self._blocks_nhwc(_xb.nhwc(x).view(n * h * w, c), n, h, w, first, last)
        if exit_flow:
            y = self._exit_nhwc(y, n, h, w)
        return _xb.nchw_view(y.view(n, h, w, -1))

    def fea_0_7(self, x):
        return self._entry_to(x, 7)

    def fea_8_12(self, x):
        return self._middle(x, 8, 12, True)

    def fea_0_4(self, x):
        return self._entry_to(x, 4)

    def fea_5_8(self, x):
        return self._middle(x, 5, 8, False)

    def fea_9_12(self, x):
        return self._middle(x, 9, 12, True)

    # ---- head --------------------------------------------------------------------------------------------------
    def logits(self, features):
        """(:317-324) relu, global average pool -> y; last_linear(dp(y)) -> x; returns (y, x).  `last_linear` exists
        once get_xception() has renamed `fc` (:349-351), exactly as in the reference."""
        n, c, h, w = features.shape
        y = _xb.ReluAvgPoolFn.apply(_xb.nhwc(features).view(n, h * w, c), n, h * w, True)
        x = _Fn.dropout(y, self.dp.p, self.dp.training)
        x = _x._apply_head(self.last_linear, x)
        return y.float(), x.float()

    def forward(self, input):
        return self.logits(self.features(input))


def get_xception(num_classes=1000, pretrained=False, weights_path=None):
    """Reference signature (:332-353) with the same loader convention as network/xception.py: the ImageNet file stores
    pointwise weights 2-D (unsqueezed on load), the classifier entries ('fc') of the file are skipped (:345), `fc`
    becomes `last_linear`.  pretrained defaults to False here: the weights file is not part of this repository."""
    model = Xception(num_classes=num_classes)
    if pretrained:
        import os
        import torch
        path = weights_path or _x.default_weights_path()
        if not os.path.exists(path):
            raise FileNotFoundError('pretrained Xception weights not found at %s (set ISTVT_XCEPTION_WEIGHTS or pass '
                                    'weights_path=)' % path)
        state = torch.load(path, map_location='cpu')
        for name, weights in state.items():
            if 'pointwise' in name and weights.dim() == 2:
                state[name] = weights.unsqueeze(-1).unsqueeze(-1)
        model.load_state_dict({k: v for k, v in state.items() if 'fc' not in k}, False)
    model.last_linear = model.fc
    del model.fc
    return model


Xception._replicate_for_data_parallel = _Fn.no_data_parallel      # nn.DataParallel: see functional.no_data_parallel
