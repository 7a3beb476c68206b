#!/usr/bin/env python3
"""GPU: the 3x3 / stride-1 convs of the 16-bit storage types, alone (HIP events, 20 launches after 5): the weight-resident kernel
(csrc/wres16.hip) against the pixel-patch kernel it replaces (dbn_set_wres16(0)), at BASELINE configs[2] / [4]'s shapes.
Each line: microseconds, dense TFLOP/s, algorithmic GB/s (source + destination once), fraction of the launch's own roofline
max(FLOPs / 2.5 PFLOP/s, bytes / 6.3 TB/s).   usage: python tools/wres_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
from db_text_minimal_amd import _lib  # noqa: E402

L = _lib.lib()
DEV = 'cuda'
AT = {torch.bfloat16: 1, torch.float16: 2}


def pack(w, mode, kind, cs=0):
    O, I, R, S = w.shape
    out = torch.empty(L.dbn_igemm_panel_floats_t(kind, O, I, R, S, mode, 1, cs), device=DEV)
    _lib.check(L.dbn_pack_weights_t(kind, w.data_ptr(), O, I, R, S, mode, 1, cs, out.data_ptr(), torch.cuda.current_stream().cuda_stream), 'pack')
    return out


def time_conv(dtype, N, H, W, Cs, Cd, mode, reps=20):
    kind = 2 if dtype == torch.float16 else 1
    x = torch.randn(N, H, W, Cs, device=DEV).to(dtype)
    w = torch.randn(Cd, Cs, 3, 3, device=DEV) * 0.05 if mode == 0 else torch.randn(Cs, Cd, 3, 3, device=DEV) * 0.05
    wp = pack(w, mode, kind, Cs if mode == 0 else 0)
    y = torch.empty(N, H, W, Cd, device=DEV, dtype=dtype)
    st = torch.cuda.current_stream().cuda_stream

    def run():
        _lib.check(L.dbn_igemm_t(AT[dtype], 1, x.data_ptr(), wp.data_ptr(), None, y.data_ptr(), N, H, W, Cs, H, W, Cd, 3, 3, 1, 1, mode, 0, 0, 1, None, st), 'igemm')
    # warm-up until the clocks have settled: with 5 launches the first configuration measured after a pause read 20-25 % slow (64 -> 64 at
    # 32 x 320^2: 324 us, 254 us after 100 launches; the forward / data-gradient "difference" of the first tables was this)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        run()
    e1.record()
    torch.cuda.synchronize()
    for _ in range(max(20, int(60.0 / max(e0.elapsed_time(e1) / 5, 1e-3)))):  # >= 60 ms of launches
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        run()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    shapes = [('cfg5 layer1 64->64 @320 x32 fp16', torch.float16, 32, 320, 320, 64, 64),
              ('cfg5 head 256->64 @320 x32 fp16', torch.float16, 32, 320, 320, 256, 64),
              ('cfg5 layer2 128->128 @160 x32 fp16', torch.float16, 32, 160, 160, 128, 128),
              ('cfg3 layer1 64->64 @160 x16 bf16', torch.bfloat16, 16, 160, 160, 64, 64),
              ('cfg3 head 256->64 @160 x16 bf16', torch.bfloat16, 16, 160, 160, 256, 64),
              ('cfg3 layer2 128->128 @80 x16 bf16', torch.bfloat16, 16, 80, 80, 128, 128)]
    for tag, dt, N, H, W, Cs, Cd in shapes:
        for mode in (0, 1):
            fl = 2.0 * N * H * W * Cs * Cd * 9
            by = 2.0 * N * H * W * (Cs + Cd)
            roof = max(fl / 2.5e15, by / 6.3e12) * 1e6
            res = []
            for on in (1, 0):
                L.dbn_set_wres16(on)
                us = time_conv(dt, N, H, W, Cs, Cd, mode)
                res.append(us)
            L.dbn_set_wres16(1)
            print('%-38s mode %d: wres %7.1f us (%6.1f TFLOP/s, %5.0f GB/s, %.2f of own roofline %5.1f us) | patch %7.1f us (%.2f) | x%.2f' % (
                tag, mode, res[0], fl / res[0] / 1e6, by / res[0] / 1e3, roof / res[0], roof, res[1], roof / res[1], res[1] / res[0]))


if __name__ == '__main__':
    main()
