import os, time, torch, torch.distributed as dist
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29545')
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
for i in range(4):
    t = time.perf_counter(); dist.barrier(); torch.cuda.synchronize(); print('barrier() %d: %.2f ms' % (i, (time.perf_counter() - t) * 1e3))
for i in range(3):
    t = time.perf_counter(); dist.barrier(device_ids=[0]); torch.cuda.synchronize(); print('barrier(device_ids) %d: %.2f ms' % (i, (time.perf_counter() - t) * 1e3))
x = torch.zeros(1, device='cuda')
for i in range(3):
    t = time.perf_counter(); dist.all_reduce(x); torch.cuda.synchronize(); print('all_reduce(1) %d: %.2f ms' % (i, (time.perf_counter() - t) * 1e3))
dist.destroy_process_group()
