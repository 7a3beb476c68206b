"""GPU box, under rocprofv3 --pmc: a handful of launches of ONE matrix kernel (the exact-fp32 weight gradient or the forward
conv of the 64->64 3x3 layer at 160 x 160, batch 16) so that its hardware counters can be read per launch.
usage: rocprofv3 --pmc <counters> --output-format csv -d <dir> -- python3 tools/wgrad_pmc_probe.py [wgrad|conv]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
from db_text_minimal_amd import _lib
L = _lib.lib()
dev = 'cuda'
what = sys.argv[1] if len(sys.argv) > 1 else 'wgrad'
N, H, Ci, Co, k = 16, 160, 64, 64, 3
st = torch.cuda.current_stream().cuda_stream
x = torch.randn(N, H, H, Ci, device=dev)
if what == 'wgrad':
    dy = torch.randn(N, H, H, Co, device=dev)
    g = torch.empty(Co, Ci, k, k, device=dev)
    slab = torch.empty(L.dbn_wgrad_slab_floats(N, H, H, Co, Ci, k, k), device=dev)
    for _ in range(4):
        _lib.check(L.dbn_wgrad_phase_t(1, 0, 0, dy.data_ptr(), x.data_ptr(), slab.data_ptr(), g.data_ptr(), N, H, H, Co, H, H, Ci, Ci, k, k, 1, 1, 1.0, st))
else:
    from gpu_util import pack, rnd
    w = rnd(Co, Ci, k, k, seed=1, scale=0.05)
    wp = pack(w, 0, 1)
    y = torch.empty(N, H, H, Co, device=dev)
    for _ in range(4):
        _lib.check(L.dbn_igemm_f32(x.data_ptr(), wp.data_ptr(), None, y.data_ptr(), N, H, H, Ci, H, H, Co, k, k, 1, 1, 0, 0, 0, st))
torch.cuda.synchronize()
