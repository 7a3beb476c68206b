"""GPU: host-side enqueue time of one train step vs its device time (is the step launch-bound?)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from db_text_minimal_amd import DBLoss, DBTextModel, DBTrainer, FusedAdam
import bench
m = DBTextModel().cuda().train()
tr = DBTrainer(m, DBLoss(), FusedAdam(m))
img, gts = bench.synthetic(16, 640, 42, torch.device('cuda'))
for _ in range(5): tr.step(img, gts)
torch.cuda.synchronize()
host = []
t0 = time.perf_counter()
for _ in range(20):
    a = time.perf_counter(); tr.step(img, gts); host.append(time.perf_counter() - a)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print('host enqueue per step: median %.2f ms (min %.2f, max %.2f); loop %.2f ms/step; drain after loop %.2f ms; total %.2f ms/step' %
      (sorted(host)[10] * 1e3, min(host) * 1e3, max(host) * 1e3, (t1 - t0) / 20 * 1e3, (t2 - t1) * 1e3, (t2 - t0) / 20 * 1e3))
