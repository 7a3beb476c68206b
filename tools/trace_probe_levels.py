"""GPU box, DBN_LIB_PATH=db_text_minimal_amd/libdbnet_hip_trace.so: per-workgroup phase timestamps of the 16-bit generic loop on the data
gradients of the FPN output conv's pyramid levels 1-3 (BASELINE configs[2]: dy 16 x 160 x 160 x 256 bf16 -> 64 channels at 80 / 40 / 20,
kernel 4 / 6 / 10, stride 2 / 4 / 8) — the launches at 0.05-0.12 of the matrix peak in profiles/r06_launch_table_bf16.txt.
usage: trace_probe_levels.py [ksplit override]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from db_text_minimal_amd import _lib  # noqa: E402
from gpu_util import DEV, L, rnd  # noqa: E402

dt, at = torch.bfloat16, 1
st = torch.cuda.current_stream().cuda_stream
N, H, Co, Cg = 16, 160, 256, 64
dy = torch.randn(N, H, H, Co, device=DEV).to(dt)
traced = bool(os.environ.get('DBN_LIB_PATH', '').endswith('trace.so'))
for g in (1, 2, 3):
    f, k, Hg = 1 << g, (1 << g) + 2, H >> g
    w = rnd(Cg, Co, k, k, seed=g, scale=0.02).to(DEV)
    wp = torch.empty(L().dbn_igemm_panel_floats_t(at, Cg, Co, k, k, 0, f, 0), device=DEV)
    _lib.check(L().dbn_pack_weights_t(at, w.data_ptr(), Cg, Co, k, k, 0, f, 0, wp.data_ptr(), st), 'pack')
    d = torch.empty(N, Hg, Hg, Cg, device=DEV, dtype=dt)
    plan = L().dbn_igemm_splitk_plan_ns(N * Hg * Hg, Cg, k * k * Co, Co, 1)
    tile = int(os.environ.get('DBN_PROBE_TILE', '0'))  # tile hint of the launch (0 = the library's choice, 1 128x128, 2 256x64, 3 128x64, 4 64x64)
    for ks in ([int(a) for a in sys.argv[1:]] or [plan]):
        slab = torch.empty(L().dbn_igemm_splitk_slab_floats(ks, N, Hg, Hg, Cg), device=DEV) if ks > 1 else None
        call = lambda: _lib.check(L().dbn_igemm_t(at, 1, dy.data_ptr(), wp.data_ptr(), None, d.data_ptr(), N, H, H, Co, Hg, Hg, Cg, k, k, f, 1, 0, 0, tile, ks,
                                                 slab.data_ptr() if slab is not None else None, st), 'igemm_t')
        for _ in range(3):
            call()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            call()
        e1.record()
        torch.cuda.synchronize()
        us_ = e0.elapsed_time(e1) * 100
        flops = 2.0 * N * Hg * Hg * Cg * Co * k * k
        cfg = L().dbn_igemm_kernel_config(at, 1, 0, N, H, H, Co, Hg, Hg, Cg, k, k, f, 1, tile, ks)
        print('level %d (k %d, stride %d, M %d, K %d): planner ksplit %d, run with %d, tile cfg %d: %.1f us, %.0f TFLOP/s' %
              (g, k, f, N * Hg * Hg, k * k * Co, plan, ks, cfg & 15, us_, flops / us_ / 1e6))
        if traced:
            nblk = 8192
            buf = torch.zeros(nblk * 8, dtype=torch.int64, device=DEV)
            assert L().dbn_set_trace(buf.data_ptr(), nblk) == 1
            call()
            torch.cuda.synchronize()
            L().dbn_set_trace(None, 0)
            t = buf.view(-1, 8).cpu().numpy().astype(np.int64)
            t = t[t[:, 0] > 0]
            if len(t):
                us = lambda a: a / 100.0
                for name, a in (('prologue', us(t[:, 1] - t[:, 0])), ('main loop', us(t[:, 2] - t[:, 1])), ('epilogue', us(t[:, 3] - t[:, 2])),
                                ('lifetime', us(t[:, 4] - t[:, 0]))):
                    print('   %-10s mean %7.2f  p10 %7.2f  p50 %7.2f  p90 %7.2f  max %7.2f us  (%d workgroups)' %
                          (name, a.mean(), np.percentile(a, 10), np.percentile(a, 50), np.percentile(a, 90), a.max(), len(t)))
                span = us(t[:, 4].max() - t[:, 0].min())
                print('   first start to last end: %.1f us' % span)
