import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from db_text_minimal_amd import DBLoss, DBTextModel, DBTrainer, FusedAdam
from oracle import dbnet_oracle as O
for arch in ('resnet50', 'deformable_resnet50'):
    for size in (320, ):
        seed = 23
        img, gts = O.synthetic_batch(2, size, seed=seed + 1)
        sd = O.new_state(seed, arch)
        for k in sd:
            if k.endswith('.bn3.weight'): sd[k] = sd[k] * 0.2
        O.BN_MOMENTUM = 1.0
        with torch.no_grad():
            O.forward(sd, img, training=True, update_stats=True)
        O.BN_MOMENTUM = 0.1
        with torch.no_grad():
            ref = O.forward(sd, img, training=False)
            taps = {}
            reft = O.forward(sd, img, training=True, update_stats=False, taps=taps)
        m = DBTextModel(arch); m.load_state_dict(sd); m = m.cuda()
        for math in ('f32', 'bf16x3', 'bf16'):
            m.engine.set_conv_math(math)
            m.eval()
            with torch.no_grad():
                pe = m(img.cuda())
            e = (pe.cpu() - ref).abs()
            m.train()
            with torch.no_grad():
                pt = m.engine.forward(img.cuda(), train=True)
            et = (pt.cpu()[:, :2] - reft[:, :2]).abs()
            # logits-level: feature map error after FPN
            f = m.engine.bufs['fpn/z'].permute(0, 3, 1, 2).cpu()
            fe = (f - taps['fpn']).abs().mean() / taps['fpn'].abs().mean()
            print('%-20s %4d %-6s eval mean %.3e max %.3e | train mean %.3e max %.3e | fpn rel err %.3e' %
                  (arch, size, math, float(e.mean()), float(e.max()), float(et.mean()), float(et.max()), float(fe)), flush=True)
