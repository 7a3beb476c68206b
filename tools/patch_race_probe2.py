import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch, torch.nn.functional as F
from gpu_util import L, rnd, DEV
from test_ops_gpu import igemm_t, pack_t
dtype, ns, kind = torch.bfloat16, 1, 1
Ci, Co, tile = 128, 128, 1
N, H, W = 16, 160, 160
w = rnd(Co, Ci, 3, 3, seed=2, scale=0.05)
x = torch.randn(N, H, W, Co, device=DEV).to(dtype)
wp = pack_t(w, 1, 1, kind)
ref = F.conv_transpose2d(x.float().permute(0, 3, 1, 2), w.to(dtype).float().to(DEV), None, 1, 1).permute(0, 2, 3, 1)
for patch in (1, 0, 1, 0):
    L().dbn_set_patch_conv(patch)
    y = torch.zeros(N, H, W, Ci, device=DEV, dtype=dtype)
    igemm_t(x, wp, None, y, 3, 1, 1, 1, ns=ns, tile=tile)
    torch.cuda.synchronize()
    e = (y.float() - ref).abs()
    bad = e > 0.05 * ref.abs().max()
    idx = bad.nonzero()
    print('patch', patch, 'max err', float(e.max()), 'bad', int(bad.sum()), 'first bad', idx[:3].tolist(), 'last bad', idx[-3:].tolist() if len(idx) else None)
    if len(idx):
        n_, h_, w_, c_ = idx[0].tolist()
        print('   images with bad px:', sorted(set(idx[:, 0].tolist()))[:20], ' channels:', sorted(set(idx[:, 3].tolist()))[:40])
