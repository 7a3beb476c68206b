"""GPU: does the order (process group first, model second — as in bench.py) matter for step time?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29546')
torch.cuda.set_device(0)
if os.environ.get('PG_FIRST', '1') == '1':
    if os.environ.get('EAGER', '1') == '1':
        dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    else:
        dist.init_process_group('nccl', rank=0, world_size=1)  # communicator created lazily at the first collective
from db_text_minimal_amd import DBLoss, DBTextModel, DBTrainer, FusedAdam
import bench
def run(tag, tr, img, gts):
    for _ in range(5): tr.step(img, gts)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): tr.step(img, gts)
    torch.cuda.synchronize(); print('%s: %.3f ms/step' % (tag, (time.perf_counter() - t0) / 20 * 1e3), flush=True)
m = DBTextModel().cuda().train()
tr = DBTrainer(m, DBLoss(), FusedAdam(m))
img, gts = bench.synthetic(16, 640, 42, torch.device('cuda'))
if not dist.is_initialized():
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
run('PG_FIRST=%s bucketed' % os.environ.get('PG_FIRST', '1'), tr, img, gts)
if os.environ.get('BARRIER', '0') == '1':
    dist.barrier(); torch.cuda.synchronize()
    run('after dist.barrier()', tr, img, gts)
dist.destroy_process_group()
