#!/usr/bin/env python3
"""GPU: the train step with the MAIN stream (forward, data gradients, BatchNorm passes: the critical chain) at a higher HIP stream priority than
the SIDE stream (weight gradients).  Skipping the BatchNorm-backward apply pass altogether is worth 8 % (fp32) / 15 % (bf16) of the step
(DESIGN section 14): the chain dgrad -> apply -> dgrad is what the step waits for, the side stream has slack — does the dispatcher's
priority shorten the chain?   usage: stream_prio_probe.py [f32|bf16] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from db_text_minimal_amd import DBLoss, DBTextModel, DBTrainer, FusedAdam
math = sys.argv[1] if len(sys.argv) > 1 else 'f32'
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device('cuda', 0)
print('priority range (least, greatest):', torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, 'priority_range') else 'n/a')
g = torch.Generator(device=dev).manual_seed(42)
img = torch.randn(16, 3, 640, 640, device=dev, generator=g)
u = torch.rand(4, 16, 640, 640, device=dev, generator=g)
gts = torch.stack([(u[0] > 0.9).float(), (u[1] > 0.05).float(), 0.3 + 0.4 * u[2], (u[3] > 0.8).float()])
import gc


def run(main_prio, side_prio):
    torch.manual_seed(42)
    model = DBTextModel().to(dev).train()
    model.engine.set_conv_math(math)
    model.engine.side_priority = side_prio
    tr = DBTrainer(model, DBLoss(), FusedAdam(model, lr=0.005))
    st = torch.cuda.Stream(device=dev, priority=main_prio) if main_prio is not None else torch.cuda.current_stream(dev)
    with torch.cuda.stream(st):
        for _ in range(5):
            tr.step(img, gts)
        gc.collect(); gc.disable()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            tr.step(img, gts)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        gc.enable()
    del tr, model
    torch.cuda.empty_cache()
    return 16 * steps / dt


for rep in range(2):
    for mp, sp in ((None, None), (-1, None), (-1, 0), (None, 0), (0, -1)):
        print('%s main priority %s, side priority %s: %.1f images/s' % (math, mp, sp, run(mp, sp)), flush=True)
