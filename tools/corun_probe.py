"""Does an MFMA-bound kernel keep its speed beside an HBM-bound one?  igemm 256->256 3x3 @160^2 bs16 (483 GF) on stream A,
bn_apply over a 420 MB tensor on stream B; each alone, then together."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch
from db_text_minimal_amd import _lib
L = _lib.lib()
dev = 'cuda'
N, H, W, C = 16, 160, 160, 256
x = torch.randn(N, H, W, C, device=dev)
w = torch.randn(C, C, 3, 3, device=dev) * 0.02
wpk = torch.empty(L.dbn_igemm_panel_floats(C, C, 3, 3, 0, 1), device=dev)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
_lib.check(L.dbn_pack_weights(w.data_ptr(), C, C, 3, 3, 0, 1, wpk.data_ptr(), torch.cuda.current_stream().cuda_stream), 'pack')
y = torch.empty(N, H, W, C, device=dev)
big = torch.randn(16, 320, 320, 64, device=dev)
big2 = torch.empty_like(big)
sc, sh = torch.ones(64, device=dev), torch.zeros(64, device=dev)
torch.cuda.synchronize()
def conv(st):
    _lib.check(L.dbn_igemm_f32(x.data_ptr(), wpk.data_ptr(), None, y.data_ptr(), N, H, W, C, H, W, C, 3, 3, 1, 1, 0, 0, 0, st.cuda_stream), 'igemm')
def bn(st):
    _lib.check(L.dbn_bn_apply(big.data_ptr(), sc.data_ptr(), sh.data_ptr(), None, None, None, big2.data_ptr(), big.numel() // 64, 64, 1, st.cuda_stream), 'bn')
def timed(fa, na, fb, nb):
    torch.cuda.synchronize()
    ea0, ea1, eb0, eb1 = (torch.cuda.Event(enable_timing=True) for _ in range(4))
    if fa:
        with torch.cuda.stream(sa):
            ea0.record()
            for _ in range(na): fa(sa)
            ea1.record()
    if fb:
        with torch.cuda.stream(sb):
            eb0.record()
            for _ in range(nb): fb(sb)
            eb1.record()
    torch.cuda.synchronize()
    return (ea0.elapsed_time(ea1) / na if fa else 0.0, eb0.elapsed_time(eb1) / nb if fb else 0.0)
for _ in range(2): timed(conv, 5, bn, 20)
GF = 2.0 * N * H * W * C * C * 9 / 1e9
GB = big.numel() * 8 / 1e9
a, _ = timed(conv, 20, None, 0)
_, b = timed(None, 0, bn, 200)
print('alone: conv %.3f ms (%.1f TF/s), bn_apply %.3f ms (%.2f TB/s)' % (a, GF / a, b, GB / b))
# together: choose counts so both streams are busy for about the same time
na = 20; nb = int(na * a / b)
a2, b2 = timed(conv, na, bn, nb)
print('together (%d conv || %d bn): conv %.3f ms (%.1f TF/s, x%.2f), bn_apply %.3f ms (%.2f TB/s, x%.2f)' % (na, nb, a2, GF / a2, a2 / a, b2, GB / b2, b2 / b))
print('  serial would take %.2f ms, together took about %.2f ms' % (na * a + nb * b, max(na * a2, nb * b2)))
a3, a4 = timed(conv, 20, conv, 20)
print('two conv streams: %.3f / %.3f ms per conv each (alone %.3f): combined %.1f TF/s' % (a3, a4, a, 2 * GF / max(a3, a4)))
