"""GPU box, DBN_LIB_PATH=db_text_minimal_amd/libdbnet_hip_trace.so (make -C db_text_minimal_amd/csrc TRACE=1): per-workgroup phase
timestamps of the exact-fp32 implicit-GEMM kernel (s_memrealtime, 100 MHz): where a tile's time goes — prologue (index arithmetic,
first loads, first barrier), main loop, epilogue (statistics, stores issued), stores completed — and whether the workgroups that share
a CU run in lockstep.  usage: trace_probe.py [patch 0|1] [stagger permille]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np
import torch
from gpu_util import L, rnd, DEV, igemm, pack

patch = int(sys.argv[1]) if len(sys.argv) > 1 else 0
stagger = int(sys.argv[2]) if len(sys.argv) > 2 else 0
L().dbn_set_patch_conv(3 if patch else 2)
L().dbn_set_stagger(stagger)
prio = int(sys.argv[3]) if len(sys.argv) > 3 else 0
L().dbn_set_phase_priority(prio)
print('phase priority', prio)
for (N, H, Ci, Co, what) in ((16, 160, 64, 64, '64->64 @160 (K = 576)'), (16, 160, 256, 64, '256->64 @160 (K = 2304)')):
    w = rnd(Co, Ci, 3, 3, seed=1, scale=0.05)
    x = torch.randn(N, H, H, Ci, device=DEV)
    y = torch.empty(N, H, H, Co, device=DEV)
    wp = pack(w, 0)
    for _ in range(5):
        igemm(x, wp, None, y, 3, 1, 1, 0)
    torch.cuda.synchronize()
    nblk = 16 * H * H // 64 * (Co // 64)
    buf = torch.zeros(nblk * 8, dtype=torch.int64, device=DEV)
    assert L().dbn_set_trace(buf.data_ptr(), nblk) == 1, 'not a TRACE build (DBN_LIB_PATH)'
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); igemm(x, wp, None, y, 3, 1, 1, 0); e1.record()
    torch.cuda.synchronize()
    L().dbn_set_trace(None, 0)
    t = buf.view(-1, 8).cpu().numpy().astype(np.int64)
    t = t[t[:, 0] > 0]
    us = lambda a: a / 100.0  # 100 MHz ticks -> us
    t0 = t[:, 0].min()
    start, pro, loop, epi, drain = us(t[:, 0] - t0), us(t[:, 1] - t[:, 0]), us(t[:, 2] - t[:, 1]), us(t[:, 3] - t[:, 2]), us(t[:, 4] - t[:, 3])
    end = us(t[:, 4] - t0)
    print('%s patch %d stagger %d: %d workgroups, kernel %.1f us (events), last store done %.1f us after the first start' % (
        what, patch, stagger, len(t), e0.elapsed_time(e1) * 1e3, end.max()))
    q = lambda a: 'mean %6.2f  p10 %6.2f  p50 %6.2f  p90 %6.2f  max %6.2f' % (a.mean(), *np.percentile(a, [10, 50, 90]), a.max())
    print('   prologue  %s us\n   main loop %s us\n   epilogue  %s us\n   store drain %s us' % (q(pro), q(loop), q(epi), q(drain)))
    # lockstep: start times of the workgroups on one CU (HW_ID bits: cu 11:8, sh 12, se 15:13 on gfx9) — group by (se, sh, cu) as reported
    hw = t[:, 7]
    cu = (hw >> 8) & 0xFF
    key = cu
    ks, cnt = np.unique(key, return_counts=True)
    k0 = ks[np.argmax(cnt)]
    sel = np.sort(start[key == k0])
    print('   start times (us) of the %d workgroups that report HW_ID cu/sh/se bits %#x (several XCDs share a code): %s' % (
        len(sel), k0, ' '.join('%.1f' % v for v in sel[:48])))
    span = us(t[:, 4] - t[:, 0])
    print('   workgroup lifetime %s us; ideal MFMA time of one tile alone on a CU: %.2f us' % (
        q(span), 2.0 * (64 * 64 if not patch else 128 * 64) * Ci * 9 / (157.3e12 / 256) * 1e6))
