"""FPN neck structure (reference: modules/segmentation_body.py:11-87)."""
from torch import nn

from .basic import BatchNorm2dParams, Conv2dParams, ConvBnRelu, Slot, _Holder


class FPN(_Holder):
    def __init__(self, backbone_out_channels, inner_channels=256):
        super().__init__()
        self.conv_out = inner_channels
        inner = inner_channels // 4
        self.reduce_conv_c2 = ConvBnRelu(backbone_out_channels[0], inner, 1)
        self.reduce_conv_c3 = ConvBnRelu(backbone_out_channels[1], inner, 1)
        self.reduce_conv_c4 = ConvBnRelu(backbone_out_channels[2], inner, 1)
        self.reduce_conv_c5 = ConvBnRelu(backbone_out_channels[3], inner, 1)
        self.smooth_p4 = ConvBnRelu(inner, inner, 3, padding=1)
        self.smooth_p3 = ConvBnRelu(inner, inner, 3, padding=1)
        self.smooth_p2 = ConvBnRelu(inner, inner, 3, padding=1)
        self.conv = nn.Sequential(Conv2dParams(self.conv_out, self.conv_out, 3, 1, 1, bias=True),
                                  BatchNorm2dParams(self.conv_out), Slot('ReLU'))
        self.out_channels = self.conv_out
