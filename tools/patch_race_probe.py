"""GPU box: repeat one pixel-patch convolution at full size and compare the runs bit for bit (race screen)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch
from gpu_util import L, rnd, nhwc, DEV
from test_ops_gpu import igemm_t, pack_t
for (dtype, ns, kind) in ((torch.bfloat16, 1, 1), (torch.float32, 3, 3)):
    for (Ci, Co, tile, mode) in ((64, 256, 1, 0), (64, 256, 1, 1), (256, 64, 3, 0), (64, 256, 3, 1), (128, 128, 1, 0), (128, 128, 1, 1)):
        N, H, W = 16, 160, 160
        w = rnd(Co, Ci, 3, 3, seed=2, scale=0.05)
        if mode == 0:
            x = torch.randn(N, H, W, Ci, device=DEV).to(dtype)
            wp = pack_t(w, 0, 1, kind, Ci)
            outs = []
            for r in range(4):
                y = torch.zeros(N, H, W, Co, device=DEV, dtype=dtype)
                igemm_t(x, wp, None, y, 3, 1, 1, 0, ns=ns, tile=tile)
                torch.cuda.synchronize()
                outs.append(y)
        else:
            x = torch.randn(N, H, W, Co, device=DEV).to(dtype)   # dy with Co channels -> dx with Ci channels
            wp = pack_t(w, 1, 1, kind)
            outs = []
            for r in range(4):
                y = torch.zeros(N, H, W, Ci, device=DEV, dtype=dtype)
                igemm_t(x, wp, None, y, 3, 1, 1, 1, ns=ns, tile=tile)
                torch.cuda.synchronize()
                outs.append(y)
        L().dbn_set_patch_conv(0)
        ref = torch.zeros_like(outs[0])
        igemm_t(x, wp, None, ref, 3, 1, 1, mode, ns=ns, tile=tile)
        torch.cuda.synchronize()
        L().dbn_set_patch_conv(1)
        bad = [int((o != ref).sum()) for o in outs]
        print(dtype, 'Ci %d Co %d tile %d mode %d: mismatching elements vs gather form per run: %s' % (Ci, Co, tile, mode, bad))
