"""GPU box, DBN_LIB_PATH=...libdbnet_hip_trace.so: phase timestamps of winograd_f32_kernel (entry, loop start, loop end, exit; ticks
from the last MFMA issue of a block to past its barrier)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np
import torch
from db_text_minimal_amd import _lib
from gpu_util import L, rnd, DEV, stream
for (N, H, Ci, Co, what) in ((16, 160, 64, 64, '64->64 @160'), (16, 160, 256, 64, '256->64 @160')):
    w = rnd(Co, Ci, 3, 3, seed=1, scale=0.05)
    x = torch.randn(N, H, H, Ci, device=DEV)
    y = torch.empty(N, H, H, Co, device=DEV)
    up = torch.empty(L().dbn_winograd_panel_floats(Co, Ci), device=DEV)
    _lib.check(L().dbn_winograd_pack(w.to(DEV).data_ptr(), Co, Ci, Ci, 0, up.data_ptr(), stream()), 'pack')
    run = lambda: _lib.check(L().dbn_winograd_conv_bn_f32(x.data_ptr(), up.data_ptr(), None, y.data_ptr(), N, H, H, Ci, Co, None, None, 0.0, 0.0,
                                                          None, None, None, None, None, None, None, stream()), 'w')
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    nblk = N * (H // 8) * (H // 16) * (Co // 64)
    buf = torch.zeros(nblk * 8, dtype=torch.int64, device=DEV)
    assert L().dbn_set_trace(buf.data_ptr(), nblk) == 1
    run()
    torch.cuda.synchronize()
    L().dbn_set_trace(None, 0)
    t = buf.view(-1, 8).cpu().numpy().astype(np.int64)
    us = lambda a: a / 100.0
    q = lambda a: 'mean %6.2f  p10 %6.2f  p50 %6.2f  p90 %6.2f' % (a.mean(), *np.percentile(a, [10, 50, 90]))
    print('%s: %d workgroups, %d channel blocks' % (what, len(t), Ci // 16))
    print('   prologue %s us\n   loop     %s us\n   of which store+barrier %s us\n   epilogue %s us' % (
        q(us(t[:, 1] - t[:, 0])), q(us(t[:, 2] - t[:, 1])), q(us(t[:, 5])), q(us(t[:, 3] - t[:, 2]))))
    print('   loop per block %.2f us; MFMA time of one block for one wave alone: %.2f us' % (us(t[:, 2] - t[:, 1]).mean() / (Ci // 16), 64 * 64 / 2400.0))
    # Round 5: are the two residents of a CU in LOCKSTEP?  Group the workgroups by the CU they ran on (HW_ID: cu 11:8, sh 12, se 15:13;
    # XCC_ID) and measure, per CU, how long 0 / 1 / 2 of its residents were inside their main loop.
    hw = t[:, 7]
    key = ((hw >> 32) << 8) | ((hw >> 8) & 0xFF)
    t0_, t1_ = t[:, 0].min(), t[:, 3].max()
    occ = np.zeros(3)
    for k_ in np.unique(key):
        sel = t[key == k_]
        ev = sorted([(a, 1) for a in sel[:, 1]] + [(b, -1) for b in sel[:, 2]])
        cur, last = 0, t0_
        for tt, d in ev:
            occ[min(cur, 2)] += tt - last
            cur, last = cur + d, tt
        occ[min(cur, 2)] += t1_ - last
    occ /= occ.sum()
    print('   %d CUs seen, %.1f workgroups per CU; share of the launch with 0 / 1 / 2 residents of a CU inside the main loop: %.3f / %.3f / %.3f'
          % (len(np.unique(key)), len(t) / len(np.unique(key)), occ[0], occ[1], occ[2]))
    print('   launch %.1f us (first entry to last exit)' % us(t1_ - t0_))
