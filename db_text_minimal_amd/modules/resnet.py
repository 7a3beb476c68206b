"""ResNet-18 backbone structure (reference: modules/resnet.py:37-91,162-255).

Only the topology and the parameters live here; `engine.Engine` walks it and
launches the HIP kernels.  The dead parameters the reference constructs but
never uses (`fc`, `smooth`, resnet.py:193-195) are kept so that state_dict
keys and `load_state_dict` stay compatible.
"""
import math

from torch import nn

from .basic import BatchNorm2dParams, Conv2dParams, LinearParams, Slot, _Holder


class BasicBlock(_Holder):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = Conv2dParams(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = BatchNorm2dParams(planes)
        self.relu = Slot('ReLU')
        self.conv2 = Conv2dParams(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = BatchNorm2dParams(planes)
        self.downsample = downsample
        self.stride = stride


class ResNet(_Holder):
    def __init__(self, layers=(2, 2, 2, 2)):
        super().__init__()
        self.inplanes = 64
        self.conv1 = Conv2dParams(3, 64, 7, 2, 3, bias=False)
        self.bn1 = BatchNorm2dParams(64)
        self.relu = Slot('ReLU')
        self.maxpool = Slot('MaxPool2d(3,2,1)')
        self.layer1 = self._make_layer(64, layers[0])
        self.layer2 = self._make_layer(128, layers[1], stride=2)
        self.layer3 = self._make_layer(256, layers[2], stride=2)
        self.layer4 = self._make_layer(512, layers[3], stride=2)
        self.avgpool = Slot('AvgPool2d (unused)')
        self.fc = LinearParams(512, 1000)
        self.smooth = Conv2dParams(2048, 256, 1, 1, 1, bias=True)
        # resnet.py:197-203: every Conv2d ~ N(0, sqrt(2/(k*k*Cout))), BN weight 1 / bias 0
        for m in self.modules():
            if isinstance(m, Conv2dParams):
                m.weight.data.normal_(0, math.sqrt(2.0 / (m.k * m.k * m.cout)))
            elif isinstance(m, BatchNorm2dParams):
                m.weight.data.fill_(1)
                m.bias.data.zero_()

    def _make_layer(self, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes:
            downsample = nn.Sequential(Conv2dParams(self.inplanes, planes, 1, stride, 0, bias=False),
                                       BatchNorm2dParams(planes))
        layers = [BasicBlock(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes
        for _ in range(1, blocks):
            layers.append(BasicBlock(planes, planes))
        return nn.Sequential(*layers)


def resnet18(pretrained=False):
    """The reference downloads ImageNet weights here (resnet.py:245-255); this
    build has no network access, so weights come from `load_state_dict`."""
    if pretrained:
        raise RuntimeError('pretrained ImageNet weights are not bundled; load a state_dict instead')
    return ResNet((2, 2, 2, 2))
