"""DB head structure (reference: modules/segmentation_head.py:20-108)."""
from torch import nn

from .basic import BatchNorm2dParams, Conv2dParams, ConvTranspose2dParams, Slot, _Holder


class DBHead(_Holder):
    def __init__(self, in_channels, out_channels, k=50):
        super().__init__()
        self.k = k
        q = in_channels // 4
        # binarize: conv bias=True (segmentation_head.py:25); thresh: bias=False (:64-68)
        self.binarize = nn.Sequential(Conv2dParams(in_channels, q, 3, 1, 1, bias=True), BatchNorm2dParams(q), Slot('ReLU'),
                                      ConvTranspose2dParams(q, q), BatchNorm2dParams(q), Slot('ReLU'),
                                      ConvTranspose2dParams(q, 1), Slot('Sigmoid'))
        self.thresh = nn.Sequential(Conv2dParams(in_channels, q, 3, 1, 1, bias=False), BatchNorm2dParams(q), Slot('ReLU'),
                                    ConvTranspose2dParams(q, q), BatchNorm2dParams(q), Slot('ReLU'),
                                    ConvTranspose2dParams(q, 1), Slot('Sigmoid'))
        self.apply(self.weights_init)

    @staticmethod
    def weights_init(m):
        # segmentation_head.py:47-53: kaiming_normal_ on every "Conv*" class, BN w=1 b=1e-4
        if isinstance(m, (Conv2dParams, ConvTranspose2dParams)):
            nn.init.kaiming_normal_(m.weight.data)
        elif isinstance(m, BatchNorm2dParams):
            m.weight.data.fill_(1.)
            m.bias.data.fill_(1e-4)
