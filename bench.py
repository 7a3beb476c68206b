#!/usr/bin/env python3
"""Headline benchmark: DBNet (ResNet18-FPN-DBHead) training images/s at 640x640, bs 16 per GPU,
fp32, synthetic data (BASELINE.json `metric`, configs[1]; the path of SURVEY.md §8).

    python bench.py --gpus N --steps K --warmup W

A step is one iteration of the reference's loop (train.py:160-172): forward, DBLoss,
backward, [one RCCL all-reduce of the flat gradients when N > 1], Adam — all HIP kernels of
libdbnet_hip.so.  Inputs are resident in HBM before the timed region.  One process per GPU: with --gpus N > 1 and no
torch.distributed.run environment (WORLD_SIZE unset) this script starts `python -m torch.distributed.run --nproc-per-node N
bench.py ...` itself as a fresh child process (before anything here touches the GPU), forwards rank 0's ONE JSON line and exits
with the child's code; fewer than N visible GPUs -> exit code 2.  Launched BY torch.distributed.run it is one of the N ranks.

Extra objects on the JSON line:
  roofline      the dominant kernel (the MFMA kernel — implicit-GEMM conv or weight gradient — with the largest share
                of the step): algorithmic FLOP/s from HIP-event brackets around its launches inside the timed
                region, vs the MFMA peak of the conv math in use (157.3 TFLOP/s for exact fp32)
  kernels       the same figure for every igemm tile variant, the weight-gradient kernels and the
                HBM-bound DB-head kernels (instrumented extra step after the timed region)
  roofline_hbm  the DB-head group (head tail forward / backward, DBLoss forward / backward) against the HBM peak: on the
                fused kernels' own bytes and on SURVEY section 8d's unfused-algorithmic 84 B/px
  cpu_baseline  the CPU oracle (oracle/dbnet_oracle.py, plain PyTorch fp32) timed on this host
  parity        BEFORE the timed region: one step of the benchmarked configuration on the inputs of tests/golden/cfg2_16x640.npz
                (the reference's own train step at 16x3x640x640) — distances of the three maps and the five losses from the
                reference's numbers; the run exits non-zero above the north_star tolerance
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CUs x 2.4 GHz
PEAK_BF16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA (v_mfma_f32_32x32x16_bf16)
PEAK_HBM_GBS = 8000.0
HBM_ACHIEVABLE_GBS = 6300.0  # MI355X_MICROARCH.md: 6.29 TB/s measured (float4 copy): what an HBM-bound conv launch is priced against
WINOGRAD = ('winograd_f32_kernel', 'winograd_wgrad_f32_kernel')  # their bracketed FLOPs are the EXECUTED ones (engine._winograd_conv, engine.wgrad)
TRAIN_GFLOP_PER_IMAGE = 236.75  # fwd + dgrad + wgrad, no stem dgrad (dense convolution count)
# what the matrix pipe computes per conv math mode: dtype of the JSON line, wording of the workload, MFMA peak that bounds it
# (bf16x3 evaluates six bf16 products per fp32 product: its roofline is the bf16 peak / 6)
MATH = {
    'f32': ('f32', 'fp32 (exact-fp32 MFMA products; 3x3/s1 convs and their data / weight gradients: Winograd F(2x2,3x3) in fp32)',
            PEAK_F32_MFMA_TFLOPS),
    'bf16x3': ('bf16x3', 'fp32 tensors, conv products on the bf16 matrix pipe as an exact 3-way operand split (fp32-accurate)',
               PEAK_BF16_MFMA_TFLOPS / 6.0),
    'bf16': ('bf16', 'native bf16: activations, gradients and weight panels stored in bf16, bf16 MFMA, fp32 accumulate / BN statistics / '
                     'loss sums / master weights', PEAK_BF16_MFMA_TFLOPS),
    'bf16c': ('bf16', 'fp32 tensors, conv operands rounded to bf16 when staged, fp32 accumulate', PEAK_BF16_MFMA_TFLOPS),
}


def synthetic(n, size, seed, dev):
    """SURVEY.md §8d inputs, generated on the device (seed 42 + rank)."""
    g = torch.Generator(device=dev).manual_seed(seed)
    img = torch.randn(n, 3, size, size, device=dev, generator=g)
    u = torch.rand(4, n, size, size, device=dev, generator=g)
    gts = torch.stack([(u[0] > 0.9).float(), (u[1] > 0.05).float(), 0.3 + 0.4 * u[2], (u[3] > 0.8).float()])
    return img, gts


def cpu_model():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def cpu_baseline(max_seconds=30.0, n16_budget_s=120.0):
    """Oracle train step (fwd + DBLoss + backward + Adam) on the host cores (SURVEY.md §8d / BASELINE.md §4): BASELINE
    configs[0]'s shape (2x3x640x640; 1 warm-up + up to 3 timed steps, median) and the benchmarked batch (16x3x640x640): 1 warm-up +
    3 timed steps, median (fewer timed steps when the N = 2 timing predicts more than `n16_budget_s` for the four; none when one
    step alone would exceed it)."""
    from oracle import dbnet_oracle as O
    threads = torch.get_num_threads()
    n, size = 2, 640
    img, gts = O.synthetic_batch(n, size, seed=42)
    sd = O.new_state(0)
    opt = O.AdamState(lr=0.005)
    O.train_step(sd, opt, img, gts)
    times = []
    t_all = time.time()
    for _ in range(3):
        t0 = time.time()
        O.train_step(sd, opt, img, gts)
        times.append(time.time() - t0)
        if time.time() - t_all > max_seconds:
            break
    times.sort()
    med = times[len(times) // 2]
    out = {'value': round(n / med, 4), 'unit': 'images/s', 'cores': threads, 'cpu_model': cpu_model(),
           'logical_cpus': os.cpu_count(), 'kind': 'port',
           'sample': '%d train steps (fwd+DBLoss+bwd+Adam) of the CPU oracle at 2x3x640x640 fp32, median; '
                     'torch CPU threads=%d' % (len(times), threads)}
    predicted = 8.0 * med
    if predicted <= n16_budget_s:
        img16, gts16 = O.synthetic_batch(16, size, seed=42)
        t_all = time.time()
        O.train_step(sd, opt, img16, gts16)  # warm-up (allocator, thread pool at this shape)
        t16s = []
        for _ in range(3):
            t0 = time.time()
            O.train_step(sd, opt, img16, gts16)
            t16s.append(time.time() - t0)
            if time.time() - t_all + t16s[-1] > n16_budget_s:
                break
        t16s.sort()
        t16 = t16s[len(t16s) // 2]
        out['n16'] = {'value': round(16 / t16, 4), 'skipped': False, 'unit': 'images/s', 'seconds': round(t16, 2),
                      'seconds_all': [round(t, 2) for t in t16s],
                      'sample': '1 warm-up + %d timed train steps of the CPU oracle at 16x3x640x640 fp32 (the benchmarked batch), '
                                'median' % len(t16s)}
        # the line's baseline figure is the one on the benchmarked workload; BASELINE configs[0]'s shape stays beside it
        out['n2'] = {'value': out['value'], 'unit': 'images/s', 'sample': out['sample']}
        out['value'] = out['n16']['value']
        out['sample'] = out['n16']['sample'] + '; torch CPU threads=%d' % threads
    else:
        out['n16'] = {'value': None, 'skipped': True,
                      'why': 'predicted %.0f s per step on this host (> %.0f s budget)' % (predicted, n16_budget_s)}
    return out


def parity_gate(dev, math):
    """The benchmarked configuration against the REFERENCE's own numbers, in the same run, before the timed region:
    tests/golden/cfg2_16x640.npz holds the reference's train step at 16x3x640x640 (/root/reference/src/train.py:160-172 run by
    tests/golden/make_golden.py --only-cfg2: strided 4096-point samples and L2 norms of the three maps, the five losses).  Its
    inputs are regenerated from seeds (tests/golden/fixture_inputs.py: procedural weights seed 16, synthetic batch seed 116 — no
    oracle import, fixtures only), ONE step of the HIP path runs on them, and the line carries the distances.  north_star
    tolerance on the maps: 1e-3 abs + 1e-2 rel; losses 1e-5 (abs + rel) in the fp32-accurate modes.  The 16-bit modes are held
    to their own yardstick in tests/test_model_gpu.py (test_distance_to_fp64...); here they only report."""
    import importlib.util
    import numpy as np
    from db_text_minimal_amd import DBLoss, DBTextModel, DBTrainer, FusedAdam
    gdir = os.path.join(ROOT, 'tests', 'golden')
    spec = importlib.util.spec_from_file_location('fixture_inputs', os.path.join(gdir, 'fixture_inputs.py'))
    fx = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fx)
    z = np.load(os.path.join(gdir, 'cfg2_16x640.npz'))
    n, size, seed, _ = (int(v) for v in z['meta'])
    model = DBTextModel()
    model.load_state_dict(fx.procedural_fill({k: v.clone() for k, v in model.state_dict().items()}, seed))
    model = model.to(dev).train()
    model.engine.set_conv_math(math)
    img, gts = fx.synthetic_batch(n, size, seed=seed + 100)
    # distributed=False: the gate is a rank-local computation — with N > 1 ranks the process group is up already, and a trainer
    # that joined it would issue its start-up broadcasts and a gradient all-reduce that no other rank answers
    tr = DBTrainer(model, DBLoss(), FusedAdam(model, lr=0.005), distributed=False)
    preds, losses = tr.step(img.to(dev), gts.to(dev))
    torch.cuda.synchronize()
    maps_err, maps_worst, l2_rel, maps_mean = 0.0, 0.0, 0.0, 0.0
    for c, nm in enumerate('PTB'):
        a = preds[:, c].detach().double().cpu().reshape(-1)
        ref = torch.from_numpy(z['preds_%s/sample' % nm]).double()
        got = a[torch.from_numpy(fx.sample_idx(a.numel(), ref.numel()))]
        err = (got - ref).abs()
        maps_err = max(maps_err, float(err.max()))
        maps_mean = max(maps_mean, float(err.mean()))
        maps_worst = max(maps_worst, float((err / (1e-3 + 1e-2 * ref.abs())).max()))
        l2_rel = max(l2_rel, abs(float(a.pow(2).sum().sqrt()) - float(z['preds_%s/stats' % nm][2])) / float(z['preds_%s/stats' % nm][2]))
    lref = torch.from_numpy(z['losses'][0]).double()
    lerr = (losses.detach().double().cpu() - lref).abs()
    loss_worst = float((lerr / (1e-5 + 1e-5 * lref.abs())).max())
    exact = math in ('f32', 'bf16x3')
    # fp32-accurate modes: the north_star tolerance per sample.  16-bit storage: single pixels of the k = 50 step function flip between any
    # two 16-bit evaluations of one net, and in train mode at random-init weights the approximate-binary map differs by ~3e-2 on average
    # (the reference under autocast shows the same, DESIGN section 4: that mode's yardstick is tests/test_model_gpu.py's fp64-distance
    # test) — here the five losses are held to 2 % and the map distances are reported
    # AND gated on loose numeric bounds — mean map error 5e-2 (measured 3.0e-2), L2 norm of each map within 2e-2 (measured 5e-5) — so
    # that ok = true means more than "finite": a wrong 16-bit map path moves both by far more
    ok = (bool(maps_worst <= 1.0 and loss_worst <= 1.0) if exact else
          bool(maps_mean <= 5e-2 and l2_rel <= 2e-2 and float((lerr / lref.abs()).max()) <= 2e-2))
    out = {'golden': 'cfg2_16x640 (the reference\'s train step at 16x3x640x640, tests/golden/make_golden.py --only-cfg2)',
           'maps_max_err': float('%.3e' % maps_err), 'maps_tol': '1e-3 abs + 1e-2 rel on 3 x 4096 strided samples (north_star)',
           'maps_worst_err_over_tol': round(maps_worst, 4), 'maps_mean_err': float('%.3e' % maps_mean), 'maps_l2_rel_err': float('%.3e' % l2_rel),
           'loss_max_err': float('%.3e' % float(lerr.max())), 'loss_tol': '1e-5 abs + 1e-5 rel on the five losses' if exact else '2e-2 rel; maps gated at mean err <= 5e-2 and L2-norm rel err <= 2e-2 (16-bit storage: single pixels flip through k = 50; the fp64-distance test is the yardstick)',
           'conv_math': math, 'ok': ok}
    del tr, model, preds, losses
    torch.cuda.empty_cache()
    return out


def mfma_sustained(dev, math):
    """What the matrix pipe of THIS box sustains on non-zero operands in the arithmetic of `math` (csrc/mfma_probe.hip: back-to-back MFMAs
    from registers, 8 waves per CU, N(0, 0.05) operands): the chip clocks to its power budget, so the nominal peak of
    MI355X_MICROARCH.md (a product of the 2.4 GHz maximum clock) is not reachable on random data — round 6 measured 1650 TFLOP/s fp16
    against 2496 with all-zero operands.  Returned in the units of the line's `peak` (bf16x3: a sixth of the bf16 rate)."""
    from db_text_minimal_amd import _lib
    L = _lib.lib()
    kind = 0 if math == 'f32' else 1
    g = torch.Generator(device=dev).manual_seed(7)
    ops = torch.randn(65536 // 4 if kind == 0 else 65536 // 2, device=dev, generator=g) * 0.05
    if kind == 1:
        ops = ops.to(torch.bfloat16)
    out = torch.empty(524288, device=dev)
    iters = 2000 if kind == 0 else 4000
    st = torch.cuda.current_stream(dev).cuda_stream
    for _ in range(3):  # (the clock settles over the first launches)
        _lib.check(L.dbn_mfma_sustained(kind, ops.data_ptr(), out.data_ptr(), iters, st), 'mfma_sustained')
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    ev[0].record()
    for i in range(3):
        _lib.check(L.dbn_mfma_sustained(kind, ops.data_ptr(), out.data_ptr(), iters, st), 'mfma_sustained')
        ev[i + 1].record()
    torch.cuda.synchronize()
    ms = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(3))[1]
    tf = L.dbn_mfma_sustained_flops(kind, iters) / (ms * 1e-3) / 1e12
    return tf / 6.0 if math == 'bf16x3' else tf


class ClockProbe:
    """Sustained shader clock during the timed region: dbn_clock_probe (one wave, s_memtime vs s_memrealtime over 200 us)
    launched on its own stream at the start of every timed step, beside the step's kernels."""

    def __init__(self, dev, steps):
        from db_text_minimal_amd import _lib
        self.L = _lib.lib()
        self.buf = torch.zeros(steps, 2, dtype=torch.int64, device=dev)
        self.stream = torch.cuda.Stream(device=dev)
        self.khz = self.L.dbn_wall_clock_khz()
        self.i = 0
        self.L.dbn_clock_probe(self.buf[0].data_ptr(), 50, self.stream.cuda_stream)  # first launch loads the code object: not in the timed region
        self.stream.synchronize()

    def sample(self):
        if self.i < self.buf.shape[0]:
            self.L.dbn_clock_probe(self.buf[self.i].data_ptr(), 200, self.stream.cuda_stream)
            self.i += 1

    def result(self):
        v = self.buf[:self.i].cpu().double()
        mhz = sorted((v[:, 0] / v[:, 1].clamp(min=1) * self.khz / 1000.0).tolist())
        if not mhz:
            return None
        return {'median_mhz': round(mhz[len(mhz) // 2], 1), 'min_mhz': round(mhz[0], 1), 'max_mhz': round(mhz[-1], 1),
                'samples': len(mhz), 'nominal_mhz': 2400,
                'how': 's_memtime / s_memrealtime over 200 us, one wave on its own stream at the start of every timed step'}


def self_launch(args):
    """--gpus N > 1 without a torch.distributed.run environment: start the N ranks as a FRESH child process (never an exec:
    this process may not have touched the GPU yet, and must not need to), forward rank 0's JSON line, return the child's code.
    The reference is single-device (src/train.py:96-98): the whole launcher is this build's."""
    if args.backend == 'nccl' and os.environ.get('DBN_DIST_ONE_DEVICE', '0') != '1':
        have = torch.cuda.device_count()  # counting devices does not initialise the GPU
        if have < args.gpus:
            print('bench.py: --gpus %d but only %d GPU(s) visible' % (args.gpus, have), file=sys.stderr)
            return 2
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus), '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'), DBN_BENCH_SELF_LAUNCHED='1')
    res = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith('{"metric"')]
    for ln in res.stdout.splitlines():
        if not ln.startswith('{"metric"'):
            print(ln, file=sys.stderr)
    if lines:
        print(lines[-1], flush=True)
    if res.returncode == 0 and not lines:
        print('bench.py: the %d-rank child printed no JSON line' % args.gpus, file=sys.stderr)
        return 3
    return res.returncode


def dry_run(args):
    """--dry: the launch / rendezvous / exchange plumbing of an N-rank run without a GPU (tests/test_host_cpu.py runs it with
    --backend gloo): every rank joins the process group, sums a flat gradient buffer of the real size with ONE all-reduce per
    step, rank 0 prints a line with the contract's keys, value null and "dry": true.  Nothing is measured."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29511')
        dist.init_process_group(args.backend, rank=rank, world_size=world)
    if world != args.gpus:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d' % (args.gpus, world))
    from db_text_minimal_amd.train import allreduce_flat_grads
    flat = torch.full((12269378, ), float(rank + 1))  # the live gradient elements of ResNet18-FPN-DBHead (49.08 MB)
    for _ in range(args.steps):
        flat.fill_(float(rank + 1))
        scale = allreduce_flat_grads(flat, world)
    seen = dist.get_world_size() if dist.is_initialized() else 1
    ok = abs(float(flat[0]) * scale - (world + 1) / 2.0) < 1e-6  # mean over the ranks of (rank + 1)
    if rank == 0:
        print(json.dumps({'metric': 'train images/sec @640x640 bs=16/GPU', 'value': None, 'unit': 'images/s', 'n_gpus': world,
                          'steps': args.steps, 'warmup': args.warmup, 'dry': True, 'scaling': 'weak',
                          'config': {'parallelism': 'dp%d' % world, 'global_batch': world * args.batch},
                          'data_parallel': {'world_seen': seen, 'backend': args.backend, 'allreduce_mean_ok': bool(ok),
                                            'launched_by': launched_by()}}), flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    return 0 if ok else 4


def launched_by():
    if os.environ.get('DBN_BENCH_SELF_LAUNCHED') == '1':
        return 'bench.py --gpus N (self-launched torch.distributed.run child)'
    return 'torch.distributed.run (external)' if 'WORLD_SIZE' in os.environ else 'single process'


def hbm_group(summ, px, steps):
    """The DB-head group of BASELINE's metric ("DB-head GB/s vs HBM peak"): head tail forward / backward (+ its fold launches),
    DBLoss forward (+ finalize) / backward, from a KernelTimer summary over `steps` instrumented steps.  `achieved` is on the
    bytes these FUSED kernels move (the head tail also applies BatchNorm + ReLU to both 64-channel ConvT outputs and evaluates
    the last ConvT: 2 x 64 channels per quarter-resolution pixel on top of the maps); `achieved_unfused` is on SURVEY 8d's
    unfused-algorithmic 84 B per full-resolution pixel (20 epilogue fwd + 28 loss fwd + 36 loss/epilogue bwd; reference
    segmentation_head.py:106-108, losses.py:18-40,62-64,77-78) so that the figure stays comparable across fusion choices."""
    labels = [k for k in summ if k.startswith(('head_tail_fwd_kernel', 'head_tail_bwd_kernel', 'db_loss_fwd_kernel', 'db_loss_bwd_kernel'))]
    if len(labels) < 4:
        return None
    ms = sum(summ[k]['ms'] for k in labels) / steps
    own = sum(summ[k]['bytes'] for k in labels) / steps
    unfused = 84.0 * px
    a, u = own / (ms * 1e-3) / 1e9, unfused / (ms * 1e-3) / 1e9
    return {'bound': 'hbm', 'kernels': sorted(labels), 'ms_per_step': round(ms, 4), 'achieved': round(a, 1), 'peak': PEAK_HBM_GBS,
            'unit': 'GB/s', 'frac': round(a / PEAK_HBM_GBS, 4), 'bytes_per_step': round(own),
            'achieved_unfused': round(u, 1), 'frac_unfused': round(u / PEAK_HBM_GBS, 4), 'bytes_per_step_unfused': round(unfused),
            'per_kernel': {k: {'ms': round(summ[k]['ms'] / steps, 4), 'GBps': round(summ[k]['bytes'] / (summ[k]['ms'] * 1e-3) / 1e9, 1),
                               'frac': round(summ[k]['bytes'] / (summ[k]['ms'] * 1e-3) / 1e9 / PEAK_HBM_GBS, 4)} for k in sorted(labels)}}


def main():
    import warnings
    warnings.filterwarnings('ignore', message=r'.*barrier\(\).*')  # keep rank 0's stdout/stderr to the one JSON line
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=16, help='images per GPU (BASELINE: 16)')
    ap.add_argument('--size', type=int, default=640)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--math', default='f32', choices=['f32', 'bf16x3', 'bf16', 'bf16c'],
                    help="precision mode: f32 = exact-fp32 MFMA (default, BASELINE configs[1]); bf16x3 = fp32-accurate split; "
                         "bf16 = native bf16 storage + MFMA (BASELINE configs[2]); bf16c = fp32 tensors, bf16 operands")
    ap.add_argument('--single-allreduce', action='store_true', help="(the default since round 3; accepted for compatibility)")
    ap.add_argument('--bucketed-allreduce', action='store_true',
                    help="N>1: time the 4-bucket exchange issued under the backward pass as the headline instead of north_star's one "
                         "all-reduce after it (the other form is always timed too and reported under grad_allreduce_other); results are "
                         "bit-identical (tests/test_dp_gloo.py, tests/test_rccl_gpu.py)")
    ap.add_argument('--graph', action='store_true',
                    help='replay forward + loss + backward as one hipGraph (DBTrainer.use_graph) instead of launching every kernel '
                         'individually.  Measured SLOWER on ROCm 7.2 / MI355X: 33.4 vs 32.0 ms (f32), 10.76 vs 10.07 ms (bf16) — the '
                         'eager step is GPU-bound already (the host enqueues ahead) and the graph executor adds gaps between nodes')
    ap.add_argument('--serial-steps', type=int, default=3, help='instrumented single-stream steps for the kernels[] table (median)')
    ap.add_argument('--no-alt-modes', action='store_true',
                    help='skip timing the other conv-math modes (reported under alt_modes; never part of `value`)')
    ap.add_argument('--backend', default='nccl', choices=['nccl', 'gloo'], help='process-group backend (nccl = RCCL; gloo only with --dry)')
    ap.add_argument('--dry', action='store_true', help='launch / rendezvous / exchange plumbing only, no GPU work (CPU test of --gpus N)')
    ap.add_argument('--no-parity', action='store_true', help='skip the cfg2_16x640 parity gate that runs before the timed region')
    args = ap.parse_args()
    if args.backend != 'nccl' and not args.dry and os.environ.get('DBN_DIST_ONE_DEVICE', '0') != '1':
        raise SystemExit('--backend gloo is for --dry runs (the step itself has no CPU path) and for the one-GPU rehearsal of an '
                         'N-rank run (DBN_DIST_ONE_DEVICE=1: every rank on device 0, device tensors through gloo; tests/test_dist2_gpu.py)')

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:  # nothing has touched the GPU yet
        raise SystemExit(self_launch(args))
    if args.dry:
        raise SystemExit(dry_run(args))

    from db_text_minimal_amd import DBLoss, DBTextModel, DBTrainer, FusedAdam
    from db_text_minimal_amd.engine import KernelTimer
    from db_text_minimal_amd.train import init_distributed

    rank, local, world = init_distributed(args.backend)
    if world != args.gpus:  # (--gpus N > 1 without WORLD_SIZE was self-launched above: a mismatch here is a wrong external launch)
        raise SystemExit('--gpus %d but WORLD_SIZE=%d' % (args.gpus, world))
    dev = torch.device('cuda', local)
    torch.cuda.set_device(dev)

    # parity gate: the benchmarked configuration vs the reference's golden, on this GPU, in this run, before anything is timed
    parity = None
    if rank == 0 and not args.no_parity and (args.batch, args.size) == (16, 640):
        parity = parity_gate(dev, args.math)
        if not parity['ok']:
            print(json.dumps({'metric': 'train images/sec @640x640 bs=16/GPU', 'value': None, 'parity': parity}), flush=True)
            raise SystemExit('parity gate failed: the HIP path does not reproduce the reference golden cfg2_16x640')

    torch.manual_seed(42)  # utils.setup_determinism(42): same initial weights on every rank
    model = DBTextModel().to(dev).train()
    model.engine.set_conv_math(args.math)
    trainer = DBTrainer(model, DBLoss(alpha=1.0, beta=10.0, negative_ratio=3, reduction='mean'), FusedAdam(model, lr=0.005))
    trainer.overlap_allreduce = bool(args.bucketed_allreduce)
    trainer.use_graph = args.graph and not args.bucketed_allreduce  # (the bucket announcements are host callbacks: eager only)
    img, gts = synthetic(args.batch, args.size, 42 + rank, dev)
    eng = model.engine

    for _ in range(args.warmup):
        trainer.step(img, gts)

    def barrier():
        if dist.is_initialized():
            torch.cuda.synchronize()  # (gloo's barrier knows nothing of the device queue)
            dist.barrier(device_ids=[local]) if args.backend == 'nccl' else dist.barrier()
        torch.cuda.synchronize()

    # HIP events around the igemm launches of every TIMED_EVERY-th step of the timed region (an event pair per launch
    # fences the queue: bracketing all ~60 launches of all steps costs 2 % of the step time)
    timer = KernelTimer(labels=('igemm_f32_kernel', 'conv3x3_wres16_kernel', 'stem7x7_b16_kernel', 'convt2x2_b16_kernel', 'winograd_f32_kernel', 'winograd_wgrad_f32_kernel', 'convt2x2_f32_kernel', 'wgrad_f32_kernel', 'wgrad_tr_kernel', 'wgrad_patch_kernel',
                                'head_tail_fwd_kernel', 'head_tail_bwd_kernel', 'db_loss_fwd_kernel', 'db_loss_bwd_kernel'))
    TIMED_EVERY = max(4, args.steps // 2)  # two instrumented steps of the K (at 12 ms/step in bf16 an instrumented step is ~30 % slower)
    clock = ClockProbe(dev, args.steps)
    step_ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    trainer.exchange_events = [] if world > 1 else None
    # The cyclic garbage collector stays out of the timed region: a generation-2 pass of this process (torch + the model's object
    # graph) stalls the enqueueing thread for 70-90 ms — measured on a fresh box as ONE timed step of 98-117 ms against 23 ms
    # (cProfile: the whole stall inside one dict lookup), i.e. 15 % off a 20-step figure, and whether it falls inside the region
    # depends on the allocation history (first process after a fresh checkout compiles its .pyc files and shifts it in).
    # Nothing in a step creates reference cycles; the collector runs again right after the region.
    import gc
    trainer.gc_freeze = False  # (the trainer's own one-time collect + freeze comes at ITS fourth step: with --warmup < 4 that is a timed step)
    gc.collect()
    gc.disable()
    barrier()
    t0 = time.perf_counter()
    host_ms = []
    cprof = None
    for it in range(args.steps):
        th = time.perf_counter()
        step_ev[it].record()
        clock.sample()
        eng.prof = timer if it % TIMED_EVERY == 0 else None
        if it == 0 and os.environ.get('DBN_BENCH_CPROFILE'):  # debugging aid: host profile of the first timed step
            import cProfile
            cprof = cProfile.Profile()
            cprof.enable()
        preds, losses = trainer.step(img, gts, resident=True)  # (the same resident tensors every step: --graph skips its input copy)
        if cprof is not None and it == 0:
            cprof.disable()
        host_ms.append(round((time.perf_counter() - th) * 1e3, 2))
    step_ev[args.steps].record()
    barrier()
    dt = time.perf_counter() - t0
    eng.prof = None  # (the collector stays off until the alternative modes below have been timed as well: round 5 found it to be the
    # whole of the "bimodal" bf16 figure — a generation-2 pass is ~95 ms, half of twenty 10 ms steps; tools/bimodal_probe.py)
    if cprof is not None and rank == 0:
        import pstats
        pstats.Stats(cprof, stream=sys.stderr).sort_stats('tottime').print_stats(14)
    timed_steps = len(range(0, args.steps, TIMED_EVERY))
    per_step = sorted(step_ev[i].elapsed_time(step_ev[i + 1]) for i in range(args.steps))
    if dist.is_initialized():
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    final_loss = float(losses[4])
    if not (final_loss == final_loss):
        raise SystemExit('non-finite loss')

    # ---- N > 1 diagnostics: exposed part of the exchange, the other exchange form, replica divergence ------------------
    def exposed_ms(events):
        v = sorted(a.elapsed_time(b) for a, b in events)
        return round(v[len(v) // 2], 4) if v else None

    dp_diag = {'world_seen': dist.get_world_size() if dist.is_initialized() else 1, 'launched_by': launched_by()}
    if world > 1:
        form = 'bucketed' if trainer.overlap_allreduce else 'single'
        dp_diag.update({'form': form, 'allreduce_exposed_ms': exposed_ms(trainer.exchange_events),
                   'how': 'HIP events on the main stream after the last backward kernel was enqueued and after the exchange was '
                          'joined: the time the step waits for the collective(s); median over the timed steps'})
        # the other form, same number of steps, same bracket
        trainer.overlap_allreduce = not trainer.overlap_allreduce
        trainer.exchange_events = []
        for _ in range(2):
            trainer.step(img, gts)
        trainer.exchange_events = []
        barrier()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            trainer.step(img, gts)
        barrier()
        t_other = torch.tensor([time.perf_counter() - t1], device=dev, dtype=torch.float64)
        dist.all_reduce(t_other, op=dist.ReduceOp.MAX)
        dp_diag['other'] = {'form': 'bucketed' if trainer.overlap_allreduce else 'single',
                            'ms_per_step': round(float(t_other.item()) / args.steps * 1e3, 3),
                            'images_per_s': round(world * args.batch * args.steps / float(t_other.item()), 2),
                            'allreduce_exposed_ms': exposed_ms(trainer.exchange_events)}
        trainer.overlap_allreduce = not trainer.overlap_allreduce
        trainer.exchange_events = None
        (psum, psq), spread = trainer.param_checksum()
        dp_diag['param_checksum'] = {'sum': psum, 'sum_sq': psq, 'max_spread_over_ranks': spread}
        if spread != 0.0:
            raise SystemExit('data-parallel replicas diverged: parameter checksum spread %g over %d ranks' % (spread, world))

    # every kernel family again, outside the timed region, on a single stream (full bracketing serialises the streams):
    # `--serial-steps` instrumented steps, per-label MEDIAN of the step totals (one such step is too noisy to compare runs)
    serial_timers = []
    for _ in range(max(1, args.serial_steps)):
        timer2 = KernelTimer()
        eng.prof = timer2
        trainer.step(img, gts)  # every rank takes the step (it contains the gradient all-reduce); rank 0 reports
        torch.cuda.synchronize()
        eng.prof = None
        serial_timers.append(timer2)
    serial_summ = {}
    serial_runs = [t.summary(MATH[args.math][2], HBM_ACHIEVABLE_GBS) for t in serial_timers]
    for k, v in serial_runs[0].items():
        ms = sorted(r[k]['ms'] for r in serial_runs if k in r)
        serial_summ[k] = dict(v, ms=ms[len(ms) // 2], ms_min=ms[0], ms_max=ms[-1])

    # dominant kernel = the MFMA kernel (any igemm tile variant or weight-gradient variant) with the largest share of the step
    # when every kernel has the device to itself (under two streams concurrent kernels inflate each other's bracketed
    # durations, which would otherwise decide the ranking); its figures below are the ones measured INSIDE the timed region.
    dtype, math_words, peak_mfma = MATH[args.math]
    summ = timer.summary(peak_mfma, HBM_ACHIEVABLE_GBS)
    dname = max((k for k in serial_summ if serial_summ[k]['flops'] > 0 and k in summ), key=lambda k: serial_summ[k]['ms'])
    d = summ[dname]
    achieved = d['flops'] / (d['ms'] * 1e-3) / 1e12
    roofline = {'bound': 'mfma', 'kernel': dname, 'achieved': round(achieved, 2), 'peak': round(peak_mfma, 1),
                'unit': 'TFLOP/s', 'frac': round(achieved / peak_mfma, 4), 'traffic': None,
                'launches_per_step': d['launches'] // timed_steps, 'avg_launch_ms': round(d['ms'] / d['launches'], 4),
                'algorithmic_gflop_per_launch': round(d['flops'] / d['launches'] / 1e9, 3),
                'share_of_step_time': round(d['ms'] / timed_steps / (dt / args.steps * 1e3), 4),
                'timed_steps': timed_steps,
                'note': ('measured in the timed region; the backward-pass launches of this kernel share the CUs with the concurrent '
                         'weight-gradient stream, so their durations (here and in the rocprofv3 trace of this command) include that '
                         'sharing; roofline_serial is the same kernel with the streams serialised')}
    # the same fraction against what the matrix pipe sustains on this box (power-capped clock, non-zero operands): the nominal `peak` is the
    # contract's yardstick, this one says how much of the attainable rate the kernel reaches
    sustained = mfma_sustained(dev, args.math)
    roofline.update(peak_sustained=round(sustained, 1), frac_of_sustained=round(achieved / sustained, 4),
                    peak_sustained_how='dbn_mfma_sustained (csrc/mfma_probe.hip): back-to-back MFMAs of this arithmetic from registers, N(0, 0.05) '
                                       'operands, 8 waves per CU, in this run on this GPU: the chip clocks to its power budget, the nominal peak '
                                       'assumes 2.4 GHz')
    if d.get('hbm_bound', 0) > 0:  # (16-bit modes) some launches of the dominant symbol are HBM-bound at their own roofline
        roofline.update(hbm_bound_launches=d['hbm_bound'], frac_of_own_roofline=round(d['roof_ms'] / d['ms'], 4),
                        own_roofline='per launch max(FLOPs / %.0f TFLOP/s, algorithmic bytes / %.0f GB/s)' % (peak_mfma, HBM_ACHIEVABLE_GBS))
    if dname in WINOGRAD:
        # F(2x2,3x3): the matrix pipe executes 16 products per 2 x 2 output tile and channel pair where the direct convolution has 36.
        # `achieved` / `frac` are on the EXECUTED FLOPs (matrix-pipe utilisation, <= 1 by construction); the figure on the direct
        # convolution's algorithmic 2*M*N*K count is 9/4 of it and is what the step-level TFLOP/s use.
        roofline.update(flops_counted='executed by the MFMA pipe (Winograd F(2x2,3x3): 4/9 of the direct convolution\'s 2*M*N*K)',
                        achieved_algorithmic=round(achieved * 2.25, 2), frac_algorithmic=round(achieved * 2.25 / peak_mfma, 4))

    # HBM traffic of that kernel: the PMC counters cannot be collected inside this run (rocprofv3 --pmc is its own pass, separate
    # for FETCH_SIZE and WRITE_SIZE: tools/profile_round.sh), so the figure comes from the newest committed pass — and ONLY while
    # it still describes these kernels: the summary records the sha256 of csrc/ it was taken on (tools/pmc_traffic.py); when the
    # sources have changed since, or the kernel is not in it, the line says `traffic: null, traffic_stale: true` instead.
    from db_text_minimal_amd._lib import source_stamp
    roofline['traffic_stale'] = True
    suffix = '' if args.math == 'f32' else '_' + args.math  # (the PMC passes of a mode: profiles/rNN_pmc_traffic[_bf16].json)
    for tag in ('r06', 'r05', 'r04', 'r03'):
        try:
            prof = json.load(open(os.path.join(ROOT, 'profiles', tag + '_pmc_traffic' + suffix + '.json')))
        except (OSError, ValueError):
            continue
        ks = prof.get('kernels', {})
        tr = ks.get(dname)
        if tr is None:  # a label that covers several template instances of one kernel (winograd_f32_kernel<false> / <true>): launch-weighted mean
            inst = [v for k, v in ks.items() if k.startswith(dname + '<') or k.startswith(dname.rstrip('>') + ',')]
            n = sum(v['launches_sampled'] for v in inst)
            if n:
                tr = {'hbm_bytes_per_launch': int(sum(v['hbm_bytes_per_launch'] * v['launches_sampled'] for v in inst) / n)}
        if tr and prof.get('csrc_stamp') == source_stamp():
            roofline['traffic'] = tr['hbm_bytes_per_launch']
            roofline['traffic_stale'] = False
            alg = d['bytes'] / d['launches'] if d.get('bytes') else None
            if alg:
                roofline['algorithmic_bytes_per_launch'] = int(alg)
                roofline['traffic_over_algorithmic'] = round(tr['hbm_bytes_per_launch'] / alg, 3)
            roofline['traffic_unit'] = ('bytes/launch (FETCH_SIZE x2 + WRITE_SIZE, rocprofv3 PMC passes of this source tree: '
                                        'profiles/%s_pmc_traffic%s.json, csrc stamp %s)' % (tag, suffix, prof['csrc_stamp']))
            break

    kernels = []
    serial = serial_summ.get(dname)
    if serial and serial['ms'] > 0:
        a = serial['flops'] / (serial['ms'] * 1e-3) / 1e12
        roofline_serial = {'kernel': dname, 'achieved': round(a, 2), 'peak': round(peak_mfma, 1), 'unit': 'TFLOP/s',
                           'frac': round(a / peak_mfma, 4), 'launches': serial['launches'],
                           **({'achieved_algorithmic': round(a * 2.25, 2), 'frac_algorithmic': round(a * 2.25 / peak_mfma, 4)} if dname in WINOGRAD else {}),
                           'avg_launch_ms': round(serial['ms'] / serial['launches'], 4),
                           'how': 'median of %d extra steps outside the timed region, every launch bracketed by HIP events, single stream' % len(serial_runs)}
    else:
        roofline_serial = None
    if rank == 0:
        for name, v in sorted(serial_summ.items(), key=lambda kv: -kv[1]['ms']):
            ent = {'kernel': name, 'launches': v['launches'], 'ms_per_step': round(v['ms'], 3),
                   'ms_min_max': [round(v['ms_min'], 3), round(v['ms_max'], 3)]}
            if v['flops'] > 0:
                a = v['flops'] / (v['ms'] * 1e-3) / 1e12
                ent.update(bound='mfma', achieved=round(a, 2), peak=round(peak_mfma, 1), unit='TFLOP/s',
                           frac=round(a / peak_mfma, 4))
                if v.get('roof_ms', 0) > 0 and v.get('hbm_bound', 0) > 0:
                    # some launches of this symbol are HBM-bound at their roofline (16-bit modes: 64-channel layers): the honest fraction is
                    # measured time against the sum of each launch's own bound, max(FLOPs / matrix peak, bytes / 6.3 TB/s)
                    ent.update(bound='mfma+hbm (per launch)', hbm_bound_launches=v['hbm_bound'], frac_of_own_roofline=round(v['roof_ms'] / v['ms'], 4))
                if name in WINOGRAD:
                    ent.update(frac_algorithmic=round(a * 2.25 / peak_mfma, 4))
            elif v['bytes'] > 0:
                a = v['bytes'] / (v['ms'] * 1e-3) / 1e9
                ent.update(bound='hbm', achieved=round(a, 1), peak=PEAK_HBM_GBS, unit='GB/s', frac=round(a / PEAK_HBM_GBS, 4))
            kernels.append(ent)

    if rank == 0:  # the weight gradient as a whole (every tile variant + the slab reductions), serial
        wg = [v for k, v in serial_summ.items() if k.startswith(('wgrad_f32_kernel', 'wgrad_tr_kernel', 'wgrad_patch_kernel', 'winograd_wgrad_f32_kernel'))]
        red = serial_summ.get('wgrad_reduce_kernel')
        if wg:
            ms_w = sum(v['ms'] for v in wg) + (red['ms'] if red else 0.0)
            fl_w = sum(v['flops'] for v in wg)
            a = fl_w / (ms_w * 1e-3) / 1e12
            kernels.append({'kernel': 'weight gradient: every wgrad_f32_kernel / wgrad_tr_kernel / wgrad_patch_kernel / winograd_wgrad_f32_kernel launch + wgrad_reduce_kernel (FLOPs as executed)', 'launches': sum(v['launches'] for v in wg),
                            'ms_per_step': round(ms_w, 3), 'bound': 'mfma', 'achieved': round(a, 2), 'peak': round(peak_mfma, 1),
                            'unit': 'TFLOP/s', 'frac': round(a / peak_mfma, 4)})
    alt = {}
    if not args.no_alt_modes:
        for mode in ('f32', 'bf16x3', 'bf16'):
            if mode == args.math:
                continue
            eng.set_conv_math(mode)
            for _ in range(2):
                trainer.step(img, gts)
            barrier()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                trainer.step(img, gts)
            barrier()
            alt[mode] = {'images_per_s': round(world * args.batch * args.steps / (time.perf_counter() - t1), 2),
                         'dtype': MATH[mode][0], 'note': MATH[mode][1]}
            if mode == 'bf16':
                # BASELINE configs[2]'s matrix-pipe figure in THIS line (the driver records only the fp32 run): one instrumented step on a
                # single stream, the MFMA kernel with the largest share, against the nominal and the sustained bf16 rate and its own
                # per-launch roofline max(FLOPs / peak, bytes / 6.3 TB/s)
                t_alt = KernelTimer()
                eng.prof = t_alt
                trainer.step(img, gts)
                torch.cuda.synchronize()
                eng.prof = None
                sa = t_alt.summary(MATH[mode][2], HBM_ACHIEVABLE_GBS)
                mf = [k for k in sa if sa[k]['flops'] > 0]
                if mf:
                    kd = max(mf, key=lambda k: sa[k]['ms'])
                    a_ = sa[kd]['flops'] / (sa[kd]['ms'] * 1e-3) / 1e12
                    sus = mfma_sustained(dev, mode)
                    alt[mode]['roofline'] = {'bound': 'mfma', 'kernel': kd, 'achieved': round(a_, 1), 'peak': MATH[mode][2], 'unit': 'TFLOP/s',
                                             'frac': round(a_ / MATH[mode][2], 4), 'peak_sustained': round(sus, 1),
                                             'frac_of_sustained': round(a_ / sus, 4), 'launches': sa[kd]['launches'],
                                             'ms_per_step': round(sa[kd]['ms'], 3),
                                             'frac_of_own_roofline': round(sa[kd]['roof_ms'] / sa[kd]['ms'], 4) if sa[kd].get('roof_ms') else None,
                                             'all_mfma_kernels_frac': round(sum(sa[k]['flops'] for k in mf) / (sum(sa[k]['ms'] for k in mf) * 1e-3) / 1e12 / MATH[mode][2], 4),
                                             'how': 'one extra step, every launch bracketed by HIP events on a single stream'}
        eng.set_conv_math(args.math)
    gc.enable()
    if rank == 0:
        ms = dt / args.steps * 1e3
        value = world * args.batch * args.steps / dt
        line = {
            'metric': 'train images/sec @640x640 bs=16/GPU', 'value': round(value, 2), 'unit': 'images/s', 'n_gpus': world,
            'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(ms, 3), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': dtype, 'data': 'synthetic',
            'config': {'workload': 'ResNet18-FPN-DBHead DBNet train step (fwd+DBLoss+bwd+Adam), %dx%d, bs %d/GPU, %s, '
                                   'random-init weights (BASELINE configs[%d])' % (args.size, args.size, args.batch, math_words,
                                                                                    1 if args.math in ('f32', 'bf16x3') else 2),
                       'global_batch': world * args.batch, 'img_size': args.size, 'parallelism': 'dp%d' % world,
                       'grad_allreduce': (('RCCL sum all-reduce of the flat 49 MB fp32 gradient buffer per step, ' +
                                           ("ONE call after the backward pass (north_star's single collective)" if not trainer.overlap_allreduce else
                                            'issued as 4 contiguous buckets under the backward pass (FPN+head, layer4, layer3, rest; '
                                            '--bucketed-allreduce)'))
                                          if world > 1 else None)},
            'timing': {'value_from': 'wall time of the K steps between two barrier+synchronize brackets, max over ranks (driver contract)',
                       'ms_per_step_median': round(per_step[len(per_step) // 2], 3), 'ms_per_step_min': round(per_step[0], 3),
                       'ms_per_step_max': round(per_step[-1], 3), 'ms_per_step_in_order': [round(step_ev[i].elapsed_time(step_ev[i + 1]), 2) for i in range(args.steps)],
                       'host_enqueue_ms_in_order': host_ms,
                       'instrumented_steps_in_region': timed_steps,
                       'note': 'per-step figures from HIP events on the main stream at the step boundaries (rank 0); '
                               '%d of the K steps carry event brackets around their MFMA launches (roofline), which costs those steps ~2 %%' % timed_steps},
            'engine_clock': clock.result(),
            'step_launch': ('hipGraph replay of forward + DBLoss + backward (captured after %d eager steps), gradient exchange and Adam '
                            'launched eagerly; the %d instrumented steps of the region are eager' % (trainer.graph_warmup, timed_steps)
                            if trainer.use_graph else 'every kernel launched individually (default; --graph replays a captured hipGraph: measured slower)'),
            'data_parallel': dp_diag,
            'roofline': roofline,
            'roofline_serial': roofline_serial,
            # BASELINE metric, second half ("DB-head GB/s vs HBM peak"): in the timed region / with the streams serialised
            'roofline_hbm': hbm_group(summ, args.batch * args.size * args.size, timed_steps),
            'roofline_hbm_serial': hbm_group(serial_summ, args.batch * args.size * args.size, 1),
            # whole-step rates: on the dense convolution count of SURVEY §8d (an EFFECTIVE rate: the structured FPN kernels
            # skip ~0.77 TFLOP/step of it) and on the FLOPs the MFMA kernels actually execute (sum over the instrumented step)
            'step_tflops': round(TRAIN_GFLOP_PER_IMAGE * (args.size / 640.0)**2 * args.batch / ms, 2),
            'step_tflops_executed': round(sum(v['flops'] for v in serial_summ.values()) / 1e9 / ms, 2),
            'step_tflop_dense': round(TRAIN_GFLOP_PER_IMAGE * (args.size / 640.0)**2 * args.batch / 1e3, 3),
            'step_tflop_executed': round(sum(v['flops'] for v in serial_summ.values()) / 1e12, 3),
            'kernels': kernels,
            'conv_math': args.math,
            'alt_modes': alt,
            'final_total_loss': round(final_loss, 5),
            'parity': parity,
        }
        if not args.no_cpu_baseline and world == 1:  # reported at N=1 only
            line['cpu_baseline'] = cpu_baseline()
        print(json.dumps(line), flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
