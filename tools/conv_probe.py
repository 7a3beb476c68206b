"""GPU probe: time one igemm / wgrad shape through the C ABI (used under rocprofv3 --pmc too).
usage: conv_probe.py N Cin Cout k stride pad H W [mode] [tile] [iters]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch
from gpu_util import L, DEV, pack, igemm, wgrad, stream
N, Ci, Co, k, s, p, H, W = (int(v) for v in sys.argv[1:9])
mode = sys.argv[9] if len(sys.argv) > 9 else 'fwd'
tile = int(sys.argv[10]) if len(sys.argv) > 10 else 0
iters = int(sys.argv[11]) if len(sys.argv) > 11 else 10
ns = int(os.environ.get('NS', '0'))
Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
g = torch.Generator(device=DEV).manual_seed(0)
x = torch.randn(N, H, W, (Ci + 3) // 4 * 4, device=DEV, generator=g)
y = torch.randn(N, Ho, Wo, Co, device=DEV, generator=g)
w = torch.randn(Co, Ci, k, k) * 0.05
flops = 2.0 * N * Ho * Wo * Co * Ci * k * k
def run():
    if mode == 'fwd':
        igemm(x, wp, None, y, k, s, p, 0, 0, tile, ns)
    elif mode == 'dgrad':
        igemm(y, wp, None, x, k, s, p, 1, 0, tile, ns)
    else:
        wgrad(y, x, Co, Ci, k, s, p, 1.0, ns)
wp = pack(w, 0 if mode == 'fwd' else 1, s, ns)
run(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters): run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / iters
print('ns=%d ' % ns + '%s N%d %d->%d k%d s%d %dx%d tile %d: %.3f ms  %.1f TFLOP/s (algorithmic)' % (mode, N, Ci, Co, k, s, H, W, tile, ms, flops / ms / 1e9))
