import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from db_text_minimal_amd import _lib
L = _lib.lib()
dev = 'cuda'
st = torch.cuda.current_stream().cuda_stream
for (N, H, O, C, k) in ((16, 80, 128, 128, 3), (16, 160, 64, 64, 3), (16, 160, 64, 256, 3), (16, 40, 256, 256, 3), (16, 20, 512, 512, 3)):
    dy = torch.randn(N, H, H, O, device=dev); x = torch.randn(N, H, H, C, device=dev)
    slab = torch.empty(L.dbn_wgrad_slab_floats_hw(N, H, H, O, H, H, C, k, k, 4), device=dev)
    g = torch.empty(O, C, k, k, device=dev)
    def run():
        _lib.check(L.dbn_wgrad_f32(dy.data_ptr(), x.data_ptr(), slab.data_ptr(), g.data_ptr(), N, H, H, O, H, H, C, C, k, k, 1, 1, 1.0, st), 'w')
    for _ in range(3): run()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): run()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print('N%d H%d O%d C%d: splits %d slab %.1f MB, wgrad+reduce %.1f us' % (N, H, O, C, L.dbn_wgrad_splitk_hw(N, H, H, O, H, H, C, k, k), slab.numel() * 4 / 1e6, dt * 1e6), flush=True)
