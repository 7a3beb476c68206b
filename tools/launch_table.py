"""GPU box: every bracketed launch of one serialised train step with its tag, duration and rate.  usage: launch_table.py [math]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from db_text_minimal_amd import DBLoss, DBTextModel, DBTrainer, FusedAdam
from db_text_minimal_amd.engine import KernelTimer
math = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith('--') else 'f32'
torch.manual_seed(42)
m = DBTextModel().cuda().train()
m.engine.set_conv_math(math)
tr = DBTrainer(m, DBLoss(), FusedAdam(m, lr=0.005))
img, gts = bench.synthetic(16, 640, 42, torch.device('cuda'))
for _ in range(3):
    tr.step(img, gts)
import gc
gc.collect()
gc.disable()  # (a generation-2 pass inside the instrumented step shows up as a 60 ms launch)
t = KernelTimer()
m.engine.prof = t
tr.step(img, gts)
torch.cuda.synchronize()
m.engine.prof = None
rows = [(e0.elapsed_time(e1), label, tag, flops, nbytes) for label, flops, nbytes, e0, e1, tag in t.records]
peak = bench.MATH[math][2]
tot = sum(r[0] for r in rows)
print('%d launches, %.2f ms bracketed' % (len(rows), tot))
order = rows if '--order' in sys.argv else sorted(rows, key=lambda r: -r[0])[:90]
for ms, label, tag, flops, nbytes in order:
    rate = ('%6.1f TF/s %.2f' % (flops / ms / 1e9, flops / ms / 1e9 / peak)) if flops else (('%6.0f GB/s' % (nbytes / ms / 1e6)) if nbytes else '')
    print('%7.3f ms  %-44s %-48s %s' % (ms, label[:44], tag[:48], rate))
