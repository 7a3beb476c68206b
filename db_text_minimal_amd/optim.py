"""Fused flat Adam — the optimizer of /root/reference/src/train.py:114-117
(`torch.optim.Adam(lr=.005, betas=(.9,.999), eps=1e-8, weight_decay=0,
amsgrad=False)`) as ONE HIP launch over the model's flat parameter buffer
(dbn_adam_step: reads p,g,m,v, writes p,m,v = 28 B/param).

It subclasses `torch.optim.Optimizer` only for interface compatibility (so the
reference's `ReduceLROnPlateau` / `WarmupPolyLR` schedulers, train.py:119-139, accept it and drive
`param_groups[0]['lr']`); no torch optimizer arithmetic is used.
"""
import torch

from . import _lib
from ._lib import check


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, model, lr=0.005, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False):
        if weight_decay != 0 or amsgrad:
            raise NotImplementedError('weight_decay=0, amsgrad=False only (the reference configuration, example_config.yaml:68-77)')
        self.model = model
        self.engine = model.engine
        params = [p for _, p in self.engine.live_params]
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=0, amsgrad=False))
        self.step_count = 0
        self.exp_avg = None
        self.exp_avg_sq = None

    def zero_grad(self, set_to_none=True):
        """Every backward pass OVERWRITES the engine's flat gradient buffer completely, so there is nothing to clear there —
        and no accumulation either: `step()` consumes the gradients of ONE backward pass.  Gradient accumulation (a second
        backward pass without a zero_grad() / step() in between, which torch.optim.Adam would sum) is refused by step() with a
        RuntimeError instead of silently dropping the first pass; use torch.optim.Adam over `model.parameters()` for that.
        On the autograd surface the parameters' `.grad` views are dropped here so they do not keep accumulating across
        iterations."""
        for _, p in self.engine.live_params:
            p.grad = None
        self.engine.backwards_since_clear = 0
        return None

    def _ensure_state(self):
        eng = self.engine
        eng.ensure_flat()
        if self.exp_avg is None or self.exp_avg.shape != eng.flat.shape or self.exp_avg.device != eng.flat.device:
            self.exp_avg = torch.zeros_like(eng.flat)
            self.exp_avg_sq = torch.zeros_like(eng.flat)

    @torch.no_grad()
    def step(self, closure=None, grad_scale=1.0):
        """Consumes the engine's flat gradient buffer (filled by DBTrainer / engine.backward)."""
        if closure is not None:
            raise NotImplementedError('closures are not supported')
        self._ensure_state()
        eng = self.engine
        if eng.backwards_since_clear > 1:
            raise RuntimeError('FusedAdam.step(): %d backward passes ran since the last zero_grad()/step(); the flat gradient buffer '
                               'holds only the last one (no gradient accumulation on the fused path — use torch.optim.Adam over '
                               'model.parameters())' % eng.backwards_since_clear)
        eng.backwards_since_clear = 0
        g = self.param_groups[0]
        self.step_count += 1
        check(_lib.lib().dbn_adam_step(eng.flat.data_ptr(), eng.flat_grad.data_ptr(), self.exp_avg.data_ptr(),
                                       self.exp_avg_sq.data_ptr(), eng.flat.numel(), float(g['lr']), float(g['betas'][0]),
                                       float(g['betas'][1]), float(g['eps']), self.step_count, float(grad_scale), eng.stream),
              'adam_step')
        eng.mark_params_dirty()

    def state_dict(self):
        if self.engine.flat is not None or any(p.is_cuda for _, p in self.engine.live_params):
            self._ensure_state()  # moments exist (zeros) before the first step, so a fresh optimizer round-trips
        return {'step': self.step_count, 'exp_avg': self.exp_avg, 'exp_avg_sq': self.exp_avg_sq,
                'param_groups': [{k: v for k, v in self.param_groups[0].items() if k != 'params'}]}

    def load_state_dict(self, sd):
        self._ensure_state()
        self.step_count = int(sd['step'])
        for name in ('exp_avg', 'exp_avg_sq'):
            if sd.get(name) is None:  # saved before any state existed
                getattr(self, name).zero_()
            else:
                getattr(self, name).copy_(sd[name])
        self.param_groups[0].update(sd['param_groups'][0])
