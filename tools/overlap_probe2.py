import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from db_text_minimal_amd import DBLoss, DBTextModel, DBTrainer, FusedAdam
from db_text_minimal_amd.engine import KernelTimer
import bench
torch.manual_seed(0)
m = DBTextModel().cuda().train()
tr = DBTrainer(m, DBLoss(), FusedAdam(m))
img, gts = bench.synthetic(16, 640, 42, torch.device('cuda'))
def run(tag):
    for _ in range(4): tr.step(img, gts)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): tr.step(img, gts)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    timer = KernelTimer(labels=('igemm_f32_kernel<128,64,2,2,0,0,0>', ))
    m.engine.prof = timer
    for _ in range(3): tr.step(img, gts)
    torch.cuda.synchronize(); m.engine.prof = None
    s = timer.summary(); d = list(s.values())[0]
    print('%-40s %.3f ms/step %.1f img/s | igemm<128,64,2,2,0,0,0>: %.1f TF/s frac %.3f' % (tag, dt * 1e3, 16 / dt, d['flops'] / d['ms'] / 1e9, d['flops'] / d['ms'] / 1e9 / 157.3), flush=True)
for rep in range(2):
    m.engine.overlap_head_branches = True; run('head branches on two streams')
    m.engine.overlap_head_branches = False; run('head branches serial')
