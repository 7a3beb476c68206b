import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from db_text_minimal_amd import DBLoss, DBTextModel, DBTrainer, FusedAdam
import bench
torch.manual_seed(0)
print('priority range (least, greatest):', torch.cuda.Stream.priority_range())
m = DBTextModel().cuda().train()
tr = DBTrainer(m, DBLoss(), FusedAdam(m))
img, gts = bench.synthetic(16, 640, 42, torch.device('cuda'))
def run(tag, main=None):
    def go(n):
        for _ in range(n): tr.step(img, gts)
    ctx = torch.cuda.stream(main) if main is not None else None
    if ctx: ctx.__enter__()
    go(4); torch.cuda.synchronize(); t0 = time.perf_counter(); go(15); torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 15
    if ctx: ctx.__exit__(None, None, None)
    print('%-50s %.3f ms/step %.1f img/s' % (tag, dt * 1e3, 16 / dt), flush=True)
lo, hi = torch.cuda.Stream.priority_range()
run('baseline (default streams)')
for sp in sorted(set([lo, 0, hi])):
    m.engine._side = None; m.engine.side_priority = sp
    run('side priority %d, main default' % sp)
for mp in sorted(set([lo, 0, hi])):
    for sp in sorted(set([lo, 0, hi])):
        if mp == sp: continue
        m.engine._side = None; m.engine.side_priority = sp
        run('main stream priority %d, side %d' % (mp, sp), torch.cuda.Stream(priority=mp))
m.engine._side = None; m.engine.side_priority = None
m.engine.overlap_wgrad = False
run('single stream')
