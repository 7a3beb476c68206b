"""DBTextModel — MI355X-native drop-in for the reference's model class.

Same call surface as /root/reference/src/models.py:13-48: `DBTextModel()` (no
arguments), `forward(x[N,3,H,W]) -> [N,3,H,W]` (train: prob, thresh, approx
binary maps) or `[N,2,H,W]` (eval), the nn.Module contract (`.to`, `.train`,
`.eval`, `.parameters`, `.state_dict`, `.load_state_dict` with the reference's
211 keys), usable under autograd and `torch.no_grad()`.  All arithmetic runs in
libdbnet_hip.so via `engine.Engine`; there is no CPU/ATen fallback.
"""
import torch
from torch import nn

from .engine import Engine
from .modules.resnet import deformable_resnet18, deformable_resnet50, resnet18, resnet50
from .modules.segmentation_body import FPN
from .modules.segmentation_head import DBHead

# models.py:8 registers resnet18 only; the Bottleneck / deformable nets that resnet.py:285-306 defines (BASELINE configs[3])
# get their registry entries here (SURVEY A4').
backbone_dict = {
    'resnet18': {'models': resnet18, 'out': [64, 128, 256, 512]},
    'deformable_resnet18': {'models': deformable_resnet18, 'out': [64, 128, 256, 512]},
    'resnet50': {'models': resnet50, 'out': [256, 512, 1024, 2048]},
    'deformable_resnet50': {'models': deformable_resnet50, 'out': [256, 512, 1024, 2048]},
}
segmentation_body_dict = {'FPN': FPN}
segmentation_head_dict = {'DBHead': DBHead}


class _DBNetFunction(torch.autograd.Function):
    """Whole-network autograd node: forward = engine.forward, backward = engine.backward."""

    @staticmethod
    def forward(ctx, model, x, *params):
        eng = model.engine
        out = eng.forward(x, train=model.training)
        ctx.model = model
        ctx.generation = eng.generation
        ctx.names = model._param_names
        return out

    @staticmethod
    def backward(ctx, dout):
        eng = ctx.model.engine
        if eng.generation != ctx.generation:
            raise RuntimeError('DBTextModel: another forward pass ran before backward(); only one in-flight '
                               'forward/backward per model instance is supported')
        eng.backward(dout)
        grads = tuple(eng.grad_views.get(n) for n in ctx.names)
        return (None, None) + grads


class DBTextModel(nn.Module):
    def __init__(self, backbone='resnet18'):
        """`DBTextModel()` is the reference's constructor (models.py:14, resnet18); `backbone` selects another registry entry."""
        super().__init__()
        backbone_name, body_name, head_name = backbone, 'FPN', 'DBHead'
        # The reference hard-codes pretrained=True and downloads ImageNet weights
        # (models.py:17, resnet.py:253); without network access weights come from load_state_dict.
        self.backbone = backbone_dict[backbone_name]['models'](pretrained=False)
        self.segmentation_body = segmentation_body_dict[body_name](backbone_dict[backbone_name]['out'], inner_channels=256)
        self.segmentation_head = segmentation_head_dict[head_name](self.segmentation_body.out_channels, out_channels=2)
        self.name = '{}_{}_{}'.format(backbone_name, body_name, head_name)
        object.__setattr__(self, 'engine', Engine(self))
        self._param_names = [n for n, _ in self.named_parameters()]

    def forward(self, x):
        """TRAIN mode: prob_map, threshold_map, appro_binary_map; EVAL mode: prob_map, threshold_map."""
        if self.training and torch.is_grad_enabled():
            self.engine.ensure_flat()
            params = [p for _, p in self.named_parameters()]
            return _DBNetFunction.apply(self, x, *params)
        return self.engine.forward(x, train=self.training)

    def state_dict(self, *args, **kwargs):
        self.engine.flush_counters()
        return super().state_dict(*args, **kwargs)

    def load_state_dict(self, state_dict, strict=True, **kw):
        res = super().load_state_dict(state_dict, strict=strict, **kw)
        self.engine.mark_params_dirty()
        return res
