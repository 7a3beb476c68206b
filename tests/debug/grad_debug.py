"""Debug helper (GPU box): per-parameter gradient error of the HIP path vs the CPU oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))  # repo root
import torch
from db_text_minimal_amd import DBLoss, DBTextModel, DBTrainer, FusedAdam
from oracle import dbnet_oracle as O

n, size, seed = int(sys.argv[1]), int(sys.argv[2]), 3
img, gts = O.synthetic_batch(n, size, seed=seed)
sd = O.new_state(seed)
m = DBTextModel(); m.load_state_dict(sd); m = m.cuda().train()
eng = m.engine
preds = eng.forward(img.cuda(), train=True)
tr = DBTrainer(m, DBLoss(), FusedAdam(m))
losses, dpreds = tr._loss(preds, gts.cuda())
eng.backward(dpreds)
torch.cuda.synchronize()
taps = {}
preds_o, losses_o, grads_o = O.loss_and_grads(sd, img, gts)
print('preds err', float((preds.cpu() - preds_o).abs().max()), 'losses', losses.cpu().tolist(), losses_o)
for k, g in grads_o.items():
    if g is None: continue
    mine = eng.grad_views[k].cpu()
    err = float((mine - g).abs().max()); sc = float(g.abs().max())
    cos = float((mine.flatten() @ g.flatten()) / (mine.norm() * g.norm() + 1e-30))
    print('%-55s max|g| %.3e  err %.3e  rel %.3e  cos %.6f' % (k, sc, err, err / (sc + 1e-30), cos))
