import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from db_text_minimal_amd import DBLoss, DBTextModel, DBTrainer, FusedAdam
import bench
torch.manual_seed(0)
m = DBTextModel().cuda().train()
tr = DBTrainer(m, DBLoss(), FusedAdam(m))
img, gts = bench.synthetic(16, 640, 42, torch.device('cuda'))
for ov in (False, True, False, True):
    m.engine.overlap_wgrad = ov
    for _ in range(5): tr.step(img, gts)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): tr.step(img, gts)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print('overlap_wgrad=%s: %.3f ms/step %.1f img/s' % (ov, dt * 1e3, 16 / dt))
