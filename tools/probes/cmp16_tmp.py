import ctypes, os, sys
ROOT='/root/repo'
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT+'/tests')
import torch
from db_text_minimal_amd import _lib
dev='cuda'
libs = {n: ctypes.CDLL(ROOT+'/tools/probes/dbg/lib_%s.so' % n) for n in ('db16', 'nodb')}
base = _lib.lib()
st = torch.cuda.current_stream().cuda_stream
def run(l, mode, x, w, N, Hs, Ws, Cs, Hd, Wd, Cd, k, s, p, ks):
    n = base.dbn_igemm_panel_floats_t(1, *( (Cd, Cs) if mode == 0 else (Cs, Cd) ), k, k, mode, s, Cs if mode == 0 else 0)
    wp = torch.empty(n, device=dev)
    wd = w.contiguous()
    O, I = wd.shape[0], wd.shape[1]
    _lib.check(base.dbn_pack_weights_t(1, wd.data_ptr(), O, I, k, k, mode, s, Cs if mode == 0 else 0, wp.data_ptr(), st), 'pack')
    y = torch.full((N, Hd, Wd, Cd), float('nan'), device=dev, dtype=torch.bfloat16)
    slab = torch.empty(max(1, ks) * (y.numel() + 1088), device=dev) if ks > 1 else None
    f = l.dbn_igemm_t
    f.argtypes = [ctypes.c_int]*2 + [ctypes.c_void_p]*4 + [ctypes.c_int]*15 + [ctypes.c_void_p]*2
    rc = f(1, 1, x.data_ptr(), wp.data_ptr(), None, y.data_ptr(), N, Hs, Ws, Cs, Hd, Wd, Cd, k, k, s, p, mode, 0, 0, ks, None if slab is None else slab.data_ptr(), st)
    torch.cuda.synchronize()
    return rc, y
bad = 0
ok = 0
rcs = {}
cases = []
for (N, H, Cs, Cd, k, st_, p) in ((8, 200, 64, 64, 3, 1, 1), (8, 200, 256, 64, 1, 1, 0), (8, 200, 64, 256, 1, 1, 0), (8, 100, 128, 128, 3, 1, 1), (8, 200, 128, 128, 3, 2, 1),
                                (8, 100, 1152, 128, 1, 1, 0), (8, 50, 2304, 256, 1, 1, 0), (8, 25, 4608, 512, 1, 1, 0), (8, 50, 256, 256, 3, 1, 1), (8, 25, 512, 512, 3, 1, 1),
                                (8, 100, 256, 512, 1, 2, 0), (8, 50, 1024, 256, 1, 1, 0), (8, 25, 2048, 512, 1, 1, 0), (8, 25, 512, 2048, 1, 1, 0), (8, 200, 64, 64, 1, 1, 0),
                                (8, 100, 128, 64, 3, 1, 1), (8, 50, 256, 64, 3, 1, 1), (8, 25, 512, 64, 3, 1, 1), (8, 200, 256, 64, 3, 1, 1)):
    for mode in (0, 1):
        for acc in (0, ):
            cases.append((N, H, Cs, Cd, k, st_, p, mode))
for (N, H, Cs, Cd, k, s, p, mode) in cases:
    if mode == 0:
        Hd = (H + 2 * p - k) // s + 1
        x = torch.randn(N, H, H, Cs, device=dev).to(torch.bfloat16)
        w = torch.randn(Cd, Cs, k, k, device=dev) * 0.05
    else:
        Hd = H * s
        x = torch.randn(N, H, H, Cs, device=dev).to(torch.bfloat16)
        w = torch.randn(Cs, Cd, k, k, device=dev) * 0.05
    args = (N, H, H, Cs, Hd, Hd, Cd)
    r = {}
    for n, l in libs.items():
        r[n] = run(l, mode, x, w, *args, k, s, p, 1)
    ok += r['nodb'][0] == 0
    rcs[r['nodb'][0]] = rcs.get(r['nodb'][0], 0) + 1
    if r['db16'][0] != r['nodb'][0] or (r['nodb'][0] == 0 and not torch.equal(r['db16'][1].float().nan_to_num(1e30), r['nodb'][1].float().nan_to_num(1e30))):
        bad += 1
        d = (r['db16'][1].float() - r['nodb'][1].float()).abs().nan_to_num(1e30).max().item() if r['nodb'][0] == 0 else -1
        print('MISMATCH N%d H%d Cs%d Cd%d k%d s%d p%d mode%d rc %s maxdiff %.3g' % (N, H, Cs, Cd, k, s, p, mode, (r['db16'][0], r['nodb'][0]), d))
print('mismatches', bad, 'ok', ok, rcs)
