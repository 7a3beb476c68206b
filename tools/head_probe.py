"""GPU box: head_tail_fwd alone at the bench shape (16 x 320 x 320 quarter pixels, 64 channels per branch), HIP-event timed."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from db_text_minimal_amd import _lib

L = _lib.lib()
N, Hq, Wq = 16, 320, 320
d = 'cuda'
g = torch.Generator(device=d).manual_seed(0)
xb = torch.randn(N * Hq * Wq, 64, device=d, generator=g)
xt = torch.randn(N * Hq * Wq, 64, device=d, generator=g)
wb = torch.randn(64, 4, device=d, generator=g) * 0.1
wt = torch.randn(64, 4, device=d, generator=g) * 0.1
bb = torch.zeros(1, device=d); bt = torch.zeros(1, device=d)
sc = torch.ones(64, device=d); sh = torch.zeros(64, device=d)
out = torch.empty(N, 3, 2 * Hq, 2 * Wq, device=d)
p = lambda t: t.data_ptr()
def run():
    _lib.check(L.dbn_head_tail_fwd(p(xb), p(xt), p(wb), p(wt), p(bb), p(bt), p(sc), p(sh), p(sc), p(sh), p(out), N, Hq, Wq, 3, 50.0, None))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
by = xb.numel() * 8 + out.numel() * 4
for blocks in ([4096, 8192, 12288, 16384, 25600, 51200, 102400] if hasattr(L, 'dbn_debug_head_fwd_blocks') else [0]):
    if blocks: L.dbn_debug_head_fwd_blocks(blocks)
    for _ in range(5): run()
    torch.cuda.synchronize()
    ts = []
    for _ in range(20):
        e0.record(); run(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    print('head_tail_fwd blocks %5d  median %.1f us  min %.1f us  %.2f TB/s' % (blocks, ts[10], ts[0], by / ts[10] / 1e6))

# yardsticks on the same box: what a read-only and a copy pass over the same 840 MB reach (torch's own kernels, not the product's)
def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0.record(); f(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[n // 2]
big = torch.cat([xb, xt])
t = timeit(lambda: big.sum())
print('torch sum over 840 MB      %.1f us  %.2f TB/s (read only)' % (t, big.numel() * 4 / t / 1e6))
dst = torch.empty_like(big)
t = timeit(lambda: dst.copy_(big))
print('torch copy of 840 MB       %.1f us  %.2f TB/s (read + write)' % (t, big.numel() * 8 / t / 1e6))
cs = torch.ones(8, 64, device=d); ws = torch.empty(1 << 22, device=d)
t = timeit(lambda: _lib.check(L.dbn_bn_train_stats(p(big), big.shape[0], 64, p(cs[0]), p(cs[1]), 1e-5, 0.1, p(cs[2]), p(cs[3]), p(cs[4]), p(cs[5]), p(cs[6]), p(cs[7]), p(ws), None)))
print('dbn_bn_train_stats 840 MB  %.1f us  %.2f TB/s (read only, product kernel)' % (t, big.numel() * 4 / t / 1e6))
