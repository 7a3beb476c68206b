"""ResNet backbones: BasicBlock (resnet18, the only one models.py:8 registers) and Bottleneck (resnet50), each
optionally with deformable conv2 in layers 2-4 (reference: modules/resnet.py:37-159,162-306).

Only the topology and the parameters live here; `engine.Engine` walks it and
launches the HIP kernels.  The dead parameters the reference constructs but
never uses (`fc`, `smooth`, resnet.py:193-195) are kept so that state_dict
keys and `load_state_dict` stay compatible.
"""
import math

from torch import nn

from .basic import BatchNorm2dParams, Conv2dParams, LinearParams, Slot, _Holder


class DeformConv2dParams(Conv2dParams):
    """torchvision.ops.DeformConv2d(planes, planes, 3, padding=1, stride, bias=False) as the reference uses it
    (resnet.py:61-65,119-124): weight [O,I,3,3] under the key `conv2.weight`; sampling offsets come from the sibling
    `conv2_offset` conv.  The engine runs it as deformable im2col + GEMM."""
    deform = True


def _conv2(planes, stride, dcn):
    """(conv2_offset | None, conv2): resnet.py:46-65 / 103-124."""
    if dcn is None:
        return None, Conv2dParams(planes, planes, 3, stride, 1, bias=False)
    groups = dcn.get('deformable_groups', 1)
    if groups != 1:
        raise NotImplementedError('deformable_groups != 1 (the reference only builds groups = 1, resnet.py:295-306)')
    return Conv2dParams(planes, 18 * groups, 3, stride, 1, bias=True), DeformConv2dParams(planes, planes, 3, stride, 1, bias=False)


class BasicBlock(_Holder):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None, dcn=None):
        super().__init__()
        self.with_dcn = dcn is not None
        self.conv1 = Conv2dParams(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = BatchNorm2dParams(planes)
        self.relu = Slot('ReLU')
        off, conv2 = _conv2(planes, 1, dcn)
        if off is not None:
            self.conv2_offset = off
        self.conv2 = conv2
        self.bn2 = BatchNorm2dParams(planes)
        self.downsample = downsample
        self.stride = stride


class Bottleneck(_Holder):
    """resnet.py:94-159: 1x1 -> 3x3 (stride; optionally deformable) -> 1x1 (x4) + residual."""
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None, dcn=None):
        super().__init__()
        self.with_dcn = dcn is not None
        self.conv1 = Conv2dParams(inplanes, planes, 1, 1, 0, bias=False)
        self.bn1 = BatchNorm2dParams(planes)
        off, conv2 = _conv2(planes, stride, dcn)
        if off is not None:
            self.conv2_offset = off
        self.conv2 = conv2
        self.bn2 = BatchNorm2dParams(planes)
        self.conv3 = Conv2dParams(planes, planes * 4, 1, 1, 0, bias=False)
        self.bn3 = BatchNorm2dParams(planes * 4)
        self.relu = Slot('ReLU')
        self.downsample = downsample
        self.stride = stride


class ResNet(_Holder):
    def __init__(self, block=BasicBlock, layers=(2, 2, 2, 2), dcn=None):
        super().__init__()
        self.block = block
        self.dcn = dcn
        self.inplanes = 64
        self.conv1 = Conv2dParams(3, 64, 7, 2, 3, bias=False)
        self.bn1 = BatchNorm2dParams(64)
        self.relu = Slot('ReLU')
        self.maxpool = Slot('MaxPool2d(3,2,1)')
        self.layer1 = self._make_layer(block, 64, layers[0])  # resnet.py:176: layer1 never gets dcn
        self.layer2 = self._make_layer(block, 128, layers[1], stride=2, dcn=dcn)
        self.layer3 = self._make_layer(block, 256, layers[2], stride=2, dcn=dcn)
        self.layer4 = self._make_layer(block, 512, layers[3], stride=2, dcn=dcn)
        self.avgpool = Slot('AvgPool2d (unused)')
        self.fc = LinearParams(512 * block.expansion, 1000)
        self.smooth = Conv2dParams(2048, 256, 1, 1, 1, bias=True)
        # resnet.py:197-203: every Conv2d ~ N(0, sqrt(2/(k*k*Cout))), BN weight 1 / bias 0
        for m in self.modules():
            if isinstance(m, DeformConv2dParams):
                continue  # torchvision's DeformConv2d is not an nn.Conv2d: it keeps its own kaiming-uniform init
            if isinstance(m, Conv2dParams):
                m.weight.data.normal_(0, math.sqrt(2.0 / (m.k * m.k * m.cout)))
            elif isinstance(m, BatchNorm2dParams):
                m.weight.data.fill_(1)
                m.bias.data.zero_()
        if dcn is not None:  # resnet.py:204-208: offsets start at zero (DCN == plain conv at init)
            for m in self.modules():
                if isinstance(m, (BasicBlock, Bottleneck)) and hasattr(m, 'conv2_offset'):
                    m.conv2_offset.weight.data.zero_()
                    m.conv2_offset.bias.data.zero_()

    def _make_layer(self, block, planes, blocks, stride=1, dcn=None):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(Conv2dParams(self.inplanes, planes * block.expansion, 1, stride, 0, bias=False),
                                       BatchNorm2dParams(planes * block.expansion))
        layers = [block(self.inplanes, planes, stride, downsample, dcn=dcn)]
        self.inplanes = planes * block.expansion
        for _ in range(1, blocks):
            layers.append(block(self.inplanes, planes, dcn=dcn))
        return nn.Sequential(*layers)


def _no_download(pretrained):
    """The reference downloads ImageNet weights here (resnet.py:245-306); this build has no network access, so weights
    come from `load_state_dict`."""
    if pretrained:
        raise RuntimeError('pretrained ImageNet weights are not bundled; load a state_dict instead')


def resnet18(pretrained=False, **kw):
    _no_download(pretrained)
    return ResNet(BasicBlock, (2, 2, 2, 2), **kw)


def deformable_resnet18(pretrained=False, **kw):
    _no_download(pretrained)
    return ResNet(BasicBlock, (2, 2, 2, 2), dcn=dict(deformable_groups=1), **kw)


def resnet50(pretrained=False, **kw):
    _no_download(pretrained)
    return ResNet(Bottleneck, (3, 4, 6, 3), **kw)


def deformable_resnet50(pretrained=False, **kw):
    _no_download(pretrained)
    return ResNet(Bottleneck, (3, 4, 6, 3), dcn=dict(deformable_groups=1), **kw)
