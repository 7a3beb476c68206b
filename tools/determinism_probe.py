"""GPU box: is a native-bf16 training step run-to-run bit-reproducible?  usage: determinism_probe.py [patch 0/1] [math]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from db_text_minimal_amd import DBLoss, DBTextModel, DBTrainer, FusedAdam, _lib
patch = int(sys.argv[1]) if len(sys.argv) > 1 else 1
math = sys.argv[2] if len(sys.argv) > 2 else 'bf16'
_lib.lib().dbn_set_patch_conv(patch)
dev = 'cuda'
g = torch.Generator().manual_seed(5)
img = torch.randn(16, 3, 640, 640, generator=g).to(dev)
gts = (torch.rand(16, 4, 640, 640, generator=g) > 0.5).float().to(dev)
gts[:, 2] = torch.rand(16, 640, 640, generator=g).to(dev) * 0.4 + 0.3
runs = []
for r in range(3):
    torch.manual_seed(0)
    m = DBTextModel().to(dev).train()
    m.engine.set_conv_math(math)
    tr = DBTrainer(m, DBLoss(), FusedAdam(m, lr=0.005))
    preds, losses = tr.step(img, gts)
    torch.cuda.synchronize()
    runs.append((preds.clone(), losses.clone(), m.engine.flat_grad.clone()))
for r in (1, 2):
    print('patch=%d %s run0 vs run%d: preds equal %s, losses equal %s, grads equal %s (max |dgrad| %.3g)' % (
        patch, math, r, torch.equal(runs[0][0], runs[r][0]), torch.equal(runs[0][1], runs[r][1]), torch.equal(runs[0][2], runs[r][2]),
        float((runs[0][2] - runs[r][2]).abs().max())))
