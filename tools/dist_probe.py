import os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch, torch.distributed as dist
from db_text_minimal_amd import DBLoss, DBTextModel, DBTrainer, FusedAdam
import db_text_minimal_amd.train as T
import bench
def run(tag, tr, img, gts):
    for _ in range(5): tr.step(img, gts)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): tr.step(img, gts)
    torch.cuda.synchronize(); print('%s: %.3f ms/step' % (tag, (time.perf_counter() - t0) / 20 * 1e3), flush=True)
m = DBTextModel().cuda().train()
tr = DBTrainer(m, DBLoss(), FusedAdam(m))
img, gts = bench.synthetic(16, 640, 42, torch.device('cuda'))
run('no process group', tr, img, gts)
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29544')
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
class Off:
    @staticmethod
    def is_available(): return False
real = T.dist
T.dist = Off
run('process group exists, trainer does not use it', tr, img, gts)
T.dist = real
tr.overlap_allreduce = False
run('single all-reduce', tr, img, gts)
tr.overlap_allreduce = True
run('bucketed all-reduce', tr, img, gts)
from db_text_minimal_amd.engine import KernelTimer
m.engine.prof = KernelTimer(labels=('igemm_f32_kernel', ))
run('bucketed all-reduce + KernelTimer(igemm)', tr, img, gts)
T.dist = Off
m.engine.prof = KernelTimer(labels=('igemm_f32_kernel', ))
run('no dist + KernelTimer(igemm)', tr, img, gts)
T.dist = real
m.engine.prof = None
m.engine.overlap_wgrad = False
run('bucketed all-reduce, single compute stream', tr, img, gts)
tr.overlap_allreduce = False
run('single all-reduce, single compute stream', tr, img, gts)
dist.destroy_process_group()
