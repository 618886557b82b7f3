"""Model registry with the reference's entry point (reference: network/models.py:240-282,
network/models_copy.py:28-45,233-249): ``model_selection(modelname, num_out_classes, dropout,
batch_size)``.  Only the two names on the ISTVT path exist: ``'xception'`` (the stem wrapper
``XceptionVidTr`` builds, vivit.py:196) and ``'resnet_3d'`` (the CLI name that selects the ISTVT
model, train_CNN.py -mn resnet_3d -> models.py:175-180).  Every other name raises the
reference's own error (models.py:184).
"""
import os
import warnings

import torch.nn as nn

from . import xception as _xception
from .xception import return_pytorch04_xception


class TransferModel(nn.Module):
    def __init__(self, modelchoice, num_out_classes=2, dropout=0.5, batch_size=16, pretrained=None, weights_path=None,
                 **istvt_kwargs):
        """pretrained: the reference always loads the ImageNet Xception (models_copy.py:35, pretrained=True, from a
        hard-coded path).  None (default) does the same when the file exists -- at `weights_path`,
        $ISTVT_XCEPTION_WEIGHTS or the reference's own path -- and otherwise keeps the default initialisation with a
        warning (there is no network to fetch it from); True insists (FileNotFoundError), False never loads."""
        super(TransferModel, self).__init__()
        self.modelchoice = modelchoice
        if modelchoice in ['xception']:
            path = weights_path or _xception.default_weights_path()
            if pretrained is None:
                pretrained = os.path.exists(path)
                if not pretrained:
                    warnings.warn('TransferModel(\'xception\'): no pretrained weights at %s; default initialisation '
                                  '(set ISTVT_XCEPTION_WEIGHTS to load them as the reference does)' % path, stacklevel=2)
            self.model = return_pytorch04_xception(pretrained=bool(pretrained), weights_path=path)
            num_ftrs = self.model.last_linear.in_features
            if not dropout:
                self.model.last_linear = nn.Linear(num_ftrs, num_out_classes)
            else:
                self.model.last_linear = nn.Sequential(
                    nn.Dropout(p=dropout),
                    nn.Linear(num_ftrs, num_out_classes)
                )
        elif modelchoice == 'resnet_3d':
            from .vivit.vivit import XceptionVidTr
            self.model = XceptionVidTr(**istvt_kwargs)
        else:
            raise Exception('Choose valid model, e.g. resnet50')

    def get_model(self):
        return self.model

    def features(self, x):
        """reference models_copy.py:227-228: global average pool of the (un-rectified) feature maps -> (n, 2048)"""
        from istvt_amd import xblocks as xb
        f = self.model.features(x)
        n, c, h, w = f.shape
        return xb.ReluAvgPoolFn.apply(xb.nhwc(f).view(n, h * w, c), n, h * w, False).float()

    def feature_maps(self, x):
        return self.model.features(x)

    def low_level_features(self, x):
        return self.model.low_level_features(x)

    def forward(self, x):
        return self.model(x)


def model_selection(modelname, num_out_classes, dropout=None, batch_size=16, **istvt_kwargs):
    if modelname == 'xception':
        kw = {k: istvt_kwargs[k] for k in ('pretrained', 'weights_path') if k in istvt_kwargs}
        return TransferModel(modelchoice='xception', num_out_classes=num_out_classes, **kw)
    return TransferModel(modelchoice=modelname, num_out_classes=num_out_classes, batch_size=batch_size,
                         **istvt_kwargs).get_model()


TransferModel._replicate_for_data_parallel = _xception._Fn.no_data_parallel     # nn.DataParallel: see functional.no_data_parallel
