"""Model registry with the reference's entry point (reference: network/models.py:240-282,
network/models_copy.py:28-45,233-249): ``model_selection(modelname, num_out_classes, dropout,
batch_size)``.  Only the two names on the ISTVT path exist: ``'xception'`` (the stem wrapper
``XceptionVidTr`` builds, vivit.py:196) and ``'resnet_3d'`` (the CLI name that selects the ISTVT
model, train_CNN.py -mn resnet_3d -> models.py:175-180).  Every other name raises the
reference's own error (models.py:184).
"""
import torch.nn as nn

from .xception import return_pytorch04_xception


class TransferModel(nn.Module):
    def __init__(self, modelchoice, num_out_classes=2, dropout=0.5, batch_size=16, **istvt_kwargs):
        super(TransferModel, self).__init__()
        self.modelchoice = modelchoice
        if modelchoice in ['xception']:
            self.model = return_pytorch04_xception(pretrained=False)
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

    def low_level_features(self, x):
        return self.model.low_level_features(x)

    def forward(self, x):
        return self.model(x)


def model_selection(modelname, num_out_classes, dropout=None, batch_size=16, **istvt_kwargs):
    if modelname == 'xception':
        return TransferModel(modelchoice='xception', num_out_classes=num_out_classes)
    return TransferModel(modelchoice=modelname, num_out_classes=num_out_classes, batch_size=batch_size,
                         **istvt_kwargs).get_model()
