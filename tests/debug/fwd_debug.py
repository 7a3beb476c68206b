"""Debug helper (GPU box): per-stage forward error of the HIP path vs the CPU oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))  # repo root
import torch
from db_text_minimal_amd import DBTextModel
from oracle import dbnet_oracle as O

n, size, seed = int(sys.argv[1]), int(sys.argv[2]), 3
train = (len(sys.argv) < 4) or sys.argv[3] == 'train'
img, gts = O.synthetic_batch(n, size, seed=seed)
sd = O.new_state(seed)
m = DBTextModel(); m.load_state_dict(sd); m = m.cuda()
m.train(train)
eng = m.engine
with torch.no_grad():
    preds = eng.forward(img.cuda(), train=train)
torch.cuda.synchronize()
taps = {}
with torch.no_grad():
    preds_o = O.forward(sd, img, training=train, taps=taps)
names = {'pool': 'stem/pool', 'c2': 'backbone.layer1.1/out', 'c3': 'backbone.layer2.1/out', 'c4': 'backbone.layer3.1/out',
         'c5': 'backbone.layer4.1/out', 'p5': 'reduce_conv_c5/z', 'p4': 'smooth_p4/z', 'p3': 'smooth_p3/z', 'p2': 'smooth_p2/z',
         'fpn': 'fpn/z'}
for k, b in names.items():
    mine = eng.bufs[b].permute(0, 3, 1, 2).cpu()
    ref = taps[k]
    err = (mine - ref).abs()
    print('%-5s shape %-22s max|ref| %.3e  max err %.3e  mean err %.3e  rel(max) %.3e' % (k, tuple(ref.shape), float(ref.abs().max()), float(err.max()), float(err.mean()), float(err.max() / ref.abs().max())))
for c, nm in enumerate('PTB'[:preds.shape[1]]):
    err = (preds[:, c].cpu() - preds_o[:, c]).abs()
    print('%-5s max err %.3e mean err %.3e' % (nm, float(err.max()), float(err.mean())))
