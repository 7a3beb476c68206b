"""Parameter holders for the DBNet modules.

The reference builds its network out of nn.Conv2d / nn.BatchNorm2d /
nn.ConvTranspose2d modules whose forward dispatches to ATen
(/root/reference/src/modules/basic.py:7-36).  Here those modules only OWN the
parameters and buffers — same attribute names, shapes and initialisation, so
`state_dict()` keys match the reference's 211 entries — while every arithmetic
operation runs in libdbnet_hip.so, driven by `db_text_minimal_amd.engine`.
Calling one of these holders directly is an error by design (no ATen fallback).
"""
import math

import torch
from torch import nn


class _Holder(nn.Module):
    def forward(self, *a, **k):
        raise RuntimeError('%s only holds parameters; the computation runs in libdbnet_hip.so '
                           '(call the enclosing DBTextModel)' % type(self).__name__)


class Conv2dParams(_Holder):
    """weight [O,I,k,k] (+ bias [O]); default init = torch's Conv2d default
    (kaiming_uniform(a=sqrt(5)), bias U(+-1/sqrt(fan_in)))."""

    def __init__(self, cin, cout, k, stride=1, padding=0, bias=True):
        super().__init__()
        self.cin, self.cout, self.k, self.stride, self.padding = cin, cout, k, stride, padding
        self.weight = nn.Parameter(torch.empty(cout, cin, k, k))
        self.bias = nn.Parameter(torch.empty(cout)) if bias else None
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if bias:
            bound = 1.0 / math.sqrt(cin * k * k)
            nn.init.uniform_(self.bias, -bound, bound)

    def extra_repr(self):
        return '%d, %d, k=%d, s=%d, p=%d, bias=%s' % (self.cin, self.cout, self.k, self.stride, self.padding,
                                                       self.bias is not None)


class ConvTranspose2dParams(_Holder):
    """ConvTranspose2d(k=2, s=2): weight [Cin,Cout,2,2], bias [Cout]."""

    def __init__(self, cin, cout, k=2, stride=2, bias=True):
        super().__init__()
        assert k == 2 and stride == 2
        self.cin, self.cout, self.k, self.stride = cin, cout, k, stride
        self.weight = nn.Parameter(torch.empty(cin, cout, k, k))
        self.bias = nn.Parameter(torch.empty(cout)) if bias else None
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if bias:
            bound = 1.0 / math.sqrt(cout * k * k)  # torch computes fan_in from weight.size(1)
            nn.init.uniform_(self.bias, -bound, bound)


class BatchNorm2dParams(_Holder):
    def __init__(self, c, eps=1e-5, momentum=0.1):
        super().__init__()
        self.num_features, self.eps, self.momentum = c, eps, momentum
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))
        self.register_buffer('running_mean', torch.zeros(c))
        self.register_buffer('running_var', torch.ones(c))
        self.register_buffer('num_batches_tracked', torch.tensor(0, dtype=torch.long))


class LinearParams(_Holder):
    """Dead `fc` of the reference's ResNet (resnet.py:193): constructed, never used."""

    def __init__(self, cin, cout):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin))
        self.bias = nn.Parameter(torch.empty(cout))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        bound = 1.0 / math.sqrt(cin)
        nn.init.uniform_(self.bias, -bound, bound)


class Slot(_Holder):
    """Parameter-free position in an nn.Sequential (ReLU / Sigmoid in the reference);
    keeps the child indices — and therefore the state_dict keys — identical."""

    def __init__(self, what):
        super().__init__()
        self.what = what

    def extra_repr(self):
        return self.what


class ConvBnRelu(_Holder):
    """Conv(bias) -> BN -> ReLU (reference: modules/basic.py:7-36)."""

    def __init__(self, cin, cout, kernel_size, stride=1, padding=0, bias=True):
        super().__init__()
        self.conv = Conv2dParams(cin, cout, kernel_size, stride, padding, bias)
        self.bn = BatchNorm2dParams(cout)
        self.relu = Slot('ReLU')
