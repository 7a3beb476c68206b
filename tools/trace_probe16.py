"""GPU box, DBN_LIB_PATH=db_text_minimal_amd/libdbnet_hip_trace.so (make -C db_text_minimal_amd/csrc TRACE=1): per-workgroup phase timestamps
(s_memrealtime, 100 MHz) of the 16-bit pixel-patch kernel in its inference form (dbn_igemm_act_t: folded BatchNorm bias + ReLU) at the
shapes of BASELINE configs[4] — where a 128-pixel x 64-channel tile's time goes: prologue (index arithmetic, first DMA stages, first
barrier), main loop, epilogue.  usage: trace_probe16.py [bf16|fp16]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np
import torch
from db_text_minimal_amd import _lib
from gpu_util import L, rnd, DEV

dt = torch.bfloat16 if len(sys.argv) > 1 and sys.argv[1] == 'bf16' else torch.float16
at = 1 if dt == torch.bfloat16 else 2
st = torch.cuda.current_stream().cuda_stream
for (N, H, Ci, Co, what) in ((32, 320, 64, 64, 'layer1 64->64 @320 (K = 576)'), (32, 320, 256, 64, 'head 256->64 @320 (K = 2304)'), (32, 160, 128, 128, 'layer2 128->128 @160')):
    w = rnd(Co, Ci, 3, 3, seed=1, scale=0.05).to(DEV)
    b = rnd(Co, seed=2).to(DEV)
    x = torch.randn(N, H, H, Ci, device=DEV).to(dt)
    y = torch.empty(N, H, H, Co, device=DEV, dtype=dt)
    wp = torch.empty(L().dbn_igemm_panel_floats_t(at, Co, Ci, 3, 3, 0, 1, Ci), device=DEV)
    _lib.check(L().dbn_pack_weights_t(at, w.data_ptr(), Co, Ci, 3, 3, 0, 1, Ci, wp.data_ptr(), st), 'pack')
    call = lambda: _lib.check(L().dbn_igemm_act_t(at, 1, x.data_ptr(), wp.data_ptr(), b.data_ptr(), None, 1, y.data_ptr(), N, H, H, Ci, H, H, Co, 3, 3,
                                                  1, 1, 0, 0, st), 'igemm_act')
    for _ in range(3):
        call()
    torch.cuda.synchronize()
    nblk = N * H * H // 128 * (Co // 64)
    buf = torch.zeros(nblk * 8, dtype=torch.int64, device=DEV)
    assert L().dbn_set_trace(buf.data_ptr(), nblk) == 1, 'not a TRACE build (DBN_LIB_PATH)'
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); call(); e1.record()
    torch.cuda.synchronize()
    L().dbn_set_trace(None, 0)
    t = buf.view(-1, 8).cpu().numpy().astype(np.int64)
    t = t[t[:, 0] > 0]
    us = lambda a: a / 100.0
    pro, loop, epi, drain = us(t[:, 1] - t[:, 0]), us(t[:, 2] - t[:, 1]), us(t[:, 3] - t[:, 2]), us(t[:, 4] - t[:, 3])
    life = us(t[:, 4] - t[:, 0])
    mf = 2.0 * 128 * 64 * 9 * Ci / (2.5e15 / 256) * 1e6
    print('%s %s: %d workgroups traced, kernel %.1f us (events)' % (what, str(dt).split('.')[1], len(t), e0.elapsed_time(e1) * 1e3))
    for name, a in (('prologue', pro), ('main loop', loop), ('epilogue', epi), ('store drain', drain), ('lifetime', life)):
        print('   %-11s mean %6.2f  p10 %6.2f  p50 %6.2f  p90 %6.2f  max %6.2f us' % (name, a.mean(), np.percentile(a, 10), np.percentile(a, 50), np.percentile(a, 90), a.max()))
    print('   MFMA time of one tile alone on a CU: %.2f us; workgroup slots the launch used: %.0f (kernel time x workgroups / mean lifetime)' %
          (mf, len(t) * life.mean() / (e0.elapsed_time(e1) * 1e3)))
