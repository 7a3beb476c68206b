"""GPU box: the native-bf16 train step's two speeds (1170 / 1445 / 1670 images/s for the same library from
process to process).  One configuration per process (argv[1]), each timed with wall clock AND per-step host enqueue time:
  plain      tools/cfg_timing.py's loop: DBTrainer.step(img, gts), nothing else
  gcoff      ... with the cyclic garbage collector disabled (bench.py does that around its timed region)
  probe      ... with bench.py's clock probe (a 200 us one-wave kernel on its own stream at the start of every step)
  resident   ... step(img, gts, resident=True)
  bench      gcoff + probe + resident (what bench.py's timed region does)
usage: python tools/bimodal_probe.py <config> [math]"""
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from db_text_minimal_amd import DBLoss, DBTextModel, DBTrainer, FusedAdam
import bench

cfg = sys.argv[1]
math_ = sys.argv[2] if len(sys.argv) > 2 else 'bf16'
dev = torch.device('cuda')
torch.manual_seed(42)
m = DBTextModel().to(dev).train()
m.engine.set_conv_math(math_)
tr = DBTrainer(m, DBLoss(alpha=1.0, beta=10.0, negative_ratio=3, reduction='mean'), FusedAdam(m, lr=0.005))
img, gts = bench.synthetic(16, 640, 42, dev)
for _ in range(6):  # (past the trainer's one-time collect-and-freeze at its fourth step)
    tr.step(img, gts)
torch.cuda.synchronize()
probe = bench.ClockProbe(dev, 40) if cfg in ('probe', 'bench') else None
if cfg in ('gcoff', 'bench'):
    gc.collect()
    gc.disable()
host = []
t0 = time.perf_counter()
for _ in range(40):
    th = time.perf_counter()
    if probe:
        probe.sample()
    tr.step(img, gts, resident=cfg in ('resident', 'bench'))
    host.append((time.perf_counter() - th) * 1e3)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 40
host.sort()
print('%-9s %s: %.2f ms/step (%.0f images/s); host enqueue per step: median %.2f ms, max %.2f ms' % (cfg, math_, dt * 1e3, 16 / dt, host[20], host[-1]))
