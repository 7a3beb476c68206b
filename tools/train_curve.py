"""GPU box: total loss over a few dozen Adam steps on one fixed synthetic batch, in every precision mode (sanity of the training
dynamics: the curves of the 16-bit modes must follow the fp32 one).  usage: train_curve.py [steps] [batch] [size]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from db_text_minimal_amd import DBLoss, DBTextModel, DBTrainer, FusedAdam
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
size = int(sys.argv[3]) if len(sys.argv) > 3 else 320
img, gts = bench.synthetic(n, size, 42, torch.device('cuda'))
for mode in ('f32', 'bf16x3', 'bf16c', 'bf16'):
    torch.manual_seed(0)
    m = DBTextModel().cuda().train()
    m.engine.set_conv_math(mode)
    tr = DBTrainer(m, DBLoss(), FusedAdam(m, lr=0.002))
    curve = []
    for it in range(steps):
        _, losses = tr.step(img, gts)
        if it % 5 == 0 or it == steps - 1:
            curve.append(float(losses[4]))
    print('%-7s' % mode, ' '.join('%.4f' % v for v in curve))
