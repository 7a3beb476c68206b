"""GPU: wall time per train step — or, with `eval` as the sixth argument, per inference forward (BASELINE configs[4]: fp16, 32 x 1280^2) —
of a BASELINE config shape that is not the bench headline (parity cases).
usage: [DBN_TIMING_LR=x] python tools/cfg_timing.py <backbone> <batch> <size> <math> [steps] [eval]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from db_text_minimal_amd import DBLoss, DBTextModel, DBTrainer, FusedAdam
import bench

arch, n, size, math_ = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
steps = int(sys.argv[5]) if len(sys.argv) > 5 else 10
torch.manual_seed(0)
if len(sys.argv) > 6 and sys.argv[6] == 'eval':
    m = DBTextModel(arch).cuda().eval()
    m.engine.set_conv_math(math_)
    img, _ = bench.synthetic(n, size, 42, torch.device('cuda'))
    with torch.no_grad():
        for _ in range(3):
            out = m(img)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            out = m(img)
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print('%s bs%d %dx%d %s eval: %.2f ms/forward, %.1f images/s, out %s, peak mem %.1f GB' %
          (arch, n, size, size, math_, dt * 1e3, n / dt, tuple(out.shape), torch.cuda.max_memory_allocated() / 2**30))
    sys.exit(0)
m = DBTextModel(arch).cuda().train()
m.engine.set_conv_math(math_)
lr = float(os.environ.get('DBN_TIMING_LR', '0.005'))  # (the deformable nets: at the reference's 0.005 on random data the learned offsets reach tens of pixels within ten steps)
tr = DBTrainer(m, DBLoss(), FusedAdam(m, lr=lr))
img, gts = bench.synthetic(n, size, 42, torch.device('cuda'))
for _ in range(6):  # (DBTrainer collects garbage once and freezes the heap at its fourth step: ~100 ms that must not fall into the timed steps)
    p, l = tr.step(img, gts)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    p, l = tr.step(img, gts)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
offs = [float(t.float()[..., :18].abs().max()) for k, t in m.engine.bufs.items() if k.endswith('/offset')]
if offs:
    print('  (lr %g; learned offsets after the last step: max |offset| per deformable layer %s)' % (lr, ' '.join('%.1f' % v for v in offs)))
print('%s bs%d %dx%d %s: %.2f ms/step, %.1f images/s, loss %.4f, peak mem %.1f GB' %
      (arch, n, size, size, math_, dt * 1e3, n / dt, float(l[4]), torch.cuda.max_memory_allocated() / 2**30))
