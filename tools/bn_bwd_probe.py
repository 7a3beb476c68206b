"""GPU box: dbn_bn_backward_t alone (sums given: the finalize + apply launches), per tensor size and storage type — the achieved HBM rate
of the BatchNorm-backward apply pass (2 reads + 1 write).  python3 tools/bn_bwd_probe.py [at ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from db_text_minimal_amd import _lib  # noqa: E402

L = _lib.lib()
dev = torch.device('cuda', 0)
st = torch.cuda.current_stream().cuda_stream
DT = {0: torch.float32, 1: torch.bfloat16, 2: torch.float16}
SHAPES = [(16 * 160 * 160, 64), (16 * 80 * 80, 128), (16 * 40 * 40, 256), (16 * 20 * 20, 512), (16 * 320 * 320, 64), (16 * 160 * 160, 256)]
for at in [int(a) for a in sys.argv[1:]] or [0, 1]:
    for M, C in SHAPES:
        y = torch.randn(M, C, device=dev).to(DT[at])
        dout = torch.randn(M, C, device=dev).to(DT[at])
        dy = torch.empty_like(y)
        f = lambda v: torch.full((C, ), v, device=dev)
        mean, rstd, gamma, msc, msh = f(0.1), f(1.2), f(0.9), f(1.0), f(0.05)
        sums = torch.randn(2 * C, device=dev)
        dg, db = torch.empty(C, device=dev), torch.empty(C, device=dev)
        ws = torch.empty(1024 * 2 * C + 4 * C, device=dev)
        for mask in ('bn output', 'own tensor'):
            z = torch.randn(M, C, device=dev).to(DT[at]) if mask == 'own tensor' else None
            args = (at, sums.data_ptr(), 1, y.data_ptr(), z.data_ptr() if z is not None else None, None if z is not None else msc.data_ptr(),
                    None if z is not None else msh.data_ptr(), dout.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(), dy.data_ptr(),
                    None, 0, dg.data_ptr(), db.data_ptr(), None, M, C, 1.0, ws.data_ptr(), st)
            for _ in range(3):
                _lib.check(L.dbn_bn_backward_t(*args), 'bn_backward')
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 20
            e0.record()
            for _ in range(n):
                L.dbn_bn_backward_t(*args)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / n * 1e3
            nbytes = (3 + (z is not None)) * M * C * y.element_size()
            print('at %d  M %7d C %3d  mask from %-10s: %7.1f us  %6.0f GB/s (%d MB)' % (at, M, C, mask, us, nbytes / us / 1e3, nbytes >> 20))
