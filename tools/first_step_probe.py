"""GPU box, FIRST process on a fresh box: where does the host time of the first instrumented step go?  (bench.py's first timed step
took 98-117 ms there against 23 ms afterwards)"""
import cProfile, pstats, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from db_text_minimal_amd import DBLoss, DBTextModel, DBTrainer, FusedAdam
from db_text_minimal_amd.engine import KernelTimer
torch.manual_seed(42)
m = DBTextModel().cuda().train()
tr = DBTrainer(m, DBLoss(), FusedAdam(m, lr=0.005))
img, gts = bench.synthetic(16, 640, 42, torch.device('cuda'))
for _ in range(3):
    tr.step(img, gts)
torch.cuda.synchronize()
t = KernelTimer(labels=('igemm_f32_kernel', 'winograd_f32_kernel', 'winograd_wgrad_f32_kernel', 'wgrad_f32_kernel', 'head_tail_fwd_kernel'))
for k in range(3):
    m.engine.prof = t if k != 1 else None
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    pr.enable()
    tr.step(img, gts, resident=True)
    pr.disable()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print('step %d (%s): host %.1f ms, + sync %.1f ms' % (k, 'instrumented' if m.engine.prof else 'plain', (t1 - t0) * 1e3, (t2 - t1) * 1e3))
    if k == 0:
        pstats.Stats(pr).sort_stats('tottime').print_stats(12)
