"""GPU: per-launch table (time, algorithmic TFLOP/s) of every MFMA launch of one bs16 640x640 train step."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from db_text_minimal_amd import DBLoss, DBTextModel, DBTrainer, FusedAdam
from db_text_minimal_amd.engine import KernelTimer
sys.argv = sys.argv[:1]
import bench
torch.manual_seed(42)
m = DBTextModel().cuda().train()
tr = DBTrainer(m, DBLoss(), FusedAdam(m))
img, gts = bench.synthetic(16, 640, 42, torch.device('cuda'))
for _ in range(2): tr.step(img, gts)
t = KernelTimer(); m.engine.prof = t
tr.step(img, gts); torch.cuda.synchronize(); m.engine.prof = None
tot = 0
for label, flops, nbytes, e0, e1, tag in t.records:
    ms = e0.elapsed_time(e1); tot += ms
    print('%-34s %-52s %8.3f ms %7.1f TF' % (label, tag, ms, flops / ms / 1e9 if flops else 0))
print('total', tot)
