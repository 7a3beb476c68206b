"""GPU box: every bracketed launch of one inference forward (default BASELINE configs[4]: resnet18, 32 x 1280^2, fp16) with its tag, duration
and rate.  usage: launch_table_eval.py [arch batch size math]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from db_text_minimal_amd import DBTextModel  # noqa: E402
from db_text_minimal_amd.engine import KernelTimer  # noqa: E402

a = sys.argv[1:]
arch, n, size, math_ = (a[0], int(a[1]), int(a[2]), a[3]) if len(a) >= 4 else ('resnet18', 32, 1280, 'fp16')
torch.manual_seed(0)
m = DBTextModel(arch).cuda().eval()
m.engine.set_conv_math(math_)
img, _ = bench.synthetic(n, size, 42, torch.device('cuda'))
with torch.no_grad():
    for _ in range(3):
        m(img)
    import gc
    gc.collect()
    gc.disable()  # (a generation-2 pass inside the instrumented forward would show up as a long launch)
    t = KernelTimer()
    m.engine.prof = t
    m(img)
    torch.cuda.synchronize()
    m.engine.prof = None
rows = [(e0.elapsed_time(e1), label, tag, flops, nbytes) for label, flops, nbytes, e0, e1, tag in t.records]
peak = bench.MATH['bf16' if math_ == 'fp16' else math_][2]
print('%d launches, %.2f ms bracketed' % (len(rows), sum(r[0] for r in rows)))
for ms, label, tag, flops, nbytes in sorted(rows, key=lambda r: -r[0]):
    rate = ('%6.1f TF/s %.2f' % (flops / ms / 1e9, flops / ms / 1e9 / peak)) if flops else ''
    hb = (' %6.0f GB/s' % (nbytes / ms / 1e6)) if nbytes else ''
    print('%7.3f ms  %-46s %-46s %s%s' % (ms, label[:46], tag[:46], rate, hb))
