"""Host-side executor of the DBNet forward/backward on MI355X.

Walks the module tree of `models.DBTextModel` (same topology as the reference:
/root/reference/src/models.py:34-48, modules/resnet.py:70-91,231-242,
modules/segmentation_body.py:64-87, modules/segmentation_head.py:35-45) and
issues the kernels of libdbnet_hip.so on the current HIP stream.  PyTorch is
used for device memory and streams only — there is no ATen arithmetic here.

Data layout in HBM
  * activations: NHWC fp32, kept in a persistent arena (one buffer per named
    tensor, reused every step -> no allocator traffic, graph-capturable);
  * parameters: one flat fp32 buffer in the reference's OIHW layouts (the
    nn.Parameters are views into it), one flat gradient buffer of the same
    shape -> a single Adam launch and a single all-reduce per step;
  * GEMM weight panels ([K/4][Cout][4]) are re-packed from the flat buffer
    whenever the parameters change.
"""
import contextlib
import os

import torch

from . import _lib
from ._lib import check

DEAD_PREFIXES = ('backbone.fc.', 'backbone.smooth.')
K_STEP_DEFAULT = 50.0


def _p(t):
    return None if t is None else t.data_ptr()


def flat_layout(numels):
    """Offsets (in floats, 16-byte aligned) of tensors packed into one flat buffer, and its size."""
    offs, total = [], 0
    for n in numels:
        offs.append(total)
        total += (n + 3) // 4 * 4
    return offs, total


# rocprof names: <BM,BN,WM,WN,MODE,NS,AT,PATCH,EPI,BLK> (EPI = 1: epilogue with the sums of the consuming BatchNorm's backward;
# BLK = false: the stem's K walk across taps)
IGEMM_TILE_NAMES = {1: 'igemm_f32_kernel<128,128,2,2,%d,%d,%d,%s,%d,%s>', 2: 'igemm_f32_kernel<256,64,4,1,%d,%d,%d,%s,%d,%s>',
                    3: 'igemm_f32_kernel<128,64,2,2,%d,%d,%d,%s,%d,%s>', 4: 'igemm_f32_kernel<64,64,2,2,%d,%d,%d,%s,%d,%s>'}
WGRAD_TILE_NAMES = {1: 'wgrad_f32_kernel<64,192,2,2,%d,%d,%d>', 2: 'wgrad_f32_kernel<128,128,2,2,%d,%d,%d>',
                    3: 'wgrad_f32_kernel<64,128,2,2,%d,%d,%d>', 4: 'wgrad_f32_kernel<64,64,2,2,%d,%d,%d>'}
ACT_DTYPES = {0: torch.float32, 1: torch.bfloat16, 2: torch.float16}


# Debug aid (tests/test_model_gpu.py::test_results_do_not_depend_on_uninitialised_memory): what every buffer the engine allocates
# holds BEFORE its first kernel writes it.  '' = whatever the allocator returns (default); 'nan' = NaN / 0xFF bytes; 'rand' =
# finite noise that differs per allocation.  A path that never reads memory it has not written gives the same bits in all three;
# a stale-memory read shows up as a difference (or a NaN) instead of as a once-in-a-while run-to-run mismatch.
POISON = os.environ.get('DBN_POISON', '')
_poison_count = [0]


def device_empty(shape, device, dtype=torch.float32):
    t = torch.empty(shape, device=device, dtype=dtype)
    if POISON and t.numel():
        if POISON == 'nan':
            t.view(torch.uint8).fill_(0xFF) if not t.is_floating_point() else t.fill_(float('nan'))
        elif POISON == 'rand':
            _poison_count[0] += 1
            g = torch.Generator(device=t.device).manual_seed(1000 + _poison_count[0])
            if t.is_floating_point():
                t.copy_((torch.rand(t.shape, device=t.device, generator=g) * 2000.0 - 1000.0).to(t.dtype))
            else:
                t.view(torch.uint8).copy_(torch.randint(0, 255, (t.numel() * t.element_size(), ), device=t.device, generator=g,
                                                        dtype=torch.uint8).view(t.view(torch.uint8).shape))
        else:
            raise ValueError('DBN_POISON: nan | rand')
    return t


class KernelTimer:
    """HIP-event bracket around selected launches (events are recorded on the stream the
    kernels are launched on).  `labels`: None = time everything, else a tuple of label prefixes."""

    def __init__(self, labels=None):
        self.labels = labels
        self.records = []
        self._open = None

    def begin(self, label, flops=0.0, nbytes=0.0, tag=''):
        if self.labels is not None and not label.startswith(self.labels):
            return
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        self._open = (label, flops, nbytes, e0, e1, tag)

    def end(self):
        if self._open is not None:
            self._open[4].record()
            self.records.append(self._open)
            self._open = None

    def summary(self, peak_tflops=None, hbm_gbs=None):
        """label -> dict(launches, ms, flops, bytes); call after a device synchronize.
        With both peaks given every launch is also priced against ITS roofline — max(FLOPs / matrix peak, algorithmic bytes / achievable HBM
        rate): `roof_ms` is the sum of those bounds and `hbm_bound` the number of launches whose bound is the memory one (a 64 -> 64 3x3 layer
        in bf16 has 288 FLOP/B against a ridge of ~400: it must not be reported against the 2.5 PFLOP/s peak)."""
        out = {}
        for label, flops, nbytes, e0, e1, _ in self.records:
            d = out.setdefault(label, dict(launches=0, ms=0.0, flops=0.0, bytes=0.0, roof_ms=0.0, hbm_bound=0))
            d['launches'] += 1
            d['ms'] += e0.elapsed_time(e1)
            d['flops'] += flops
            d['bytes'] += nbytes
            if peak_tflops and hbm_gbs and flops > 0:
                t_m, t_h = flops / (peak_tflops * 1e9), nbytes / (hbm_gbs * 1e6)  # ms
                d['roof_ms'] += max(t_m, t_h)
                d['hbm_bound'] += int(t_h > t_m)
        return out


class _VirtualConv:
    """Conv geometry + a derived weight tensor (zero-padded / permuted copy of a parameter) for the conv helpers."""

    def __init__(self, cin, cout, k, stride, padding, weight, bias):
        self.cin, self.cout, self.k, self.stride, self.padding, self.weight, self.bias = cin, cout, k, stride, padding, weight, bias


class ConvPlan(object):
    """What one conv takes at one (geometry, input shape, conv math, mode) — see Engine.plan."""
    __slots__ = ('winograd', 'winograd_wgrad', 'apply_on_load', 'fold', 'pw16')


class _Shape:
    """A tensor's shape without the tensor (eligibility checks ahead of the launches)."""

    def __init__(self, *shape):
        self.shape = shape


class Engine:
    def __init__(self, model):
        self.model = model
        self.L = _lib.lib()
        if 'DBN_PATCH_F32' in os.environ:  # A/B runs: 0 exact fp32 on the gather loop everywhere, 1 on the pixel-patch kernel wherever eligible
            self.L.dbn_set_patch_conv(2 if os.environ['DBN_PATCH_F32'] == '0' else 3)
        if 'DBN_WINO_PERSISTENT' in os.environ:  # A/B runs: 0 one workgroup per item, 1 persistent workgroups pulling items, 2 static schedule
            self.L.dbn_set_winograd_persistent(int(os.environ['DBN_WINO_PERSISTENT']))
        if 'DBN_WINO_CBS' in os.environ:  # A/B runs: channel blocks per barrier of the Winograd patch form (1 | 2)
            self.L.dbn_set_winograd_blocks_per_barrier(int(os.environ['DBN_WINO_CBS']))
        if 'DBN_WINO_STAGGER' in os.environ:  # A/B runs: permille of one item's matrix time (0 = off)
            self.L.dbn_set_winograd_stagger(int(os.environ['DBN_WINO_STAGGER']))
        if 'DBN_PHASE_PRIO' in os.environ:  # A/B runs
            self.L.dbn_set_phase_priority(int(os.environ['DBN_PHASE_PRIO']))
        if 'DBN_STAGGER' in os.environ:  # A/B runs (permille of the nominal first-round stagger of the fp32 convs)
            self.L.dbn_set_stagger(int(os.environ['DBN_STAGGER']))
        self.bufs = {}
        self.packs = {}
        self.pack_src = {}  # pack key -> parameter tensor, for repack_params()
        self._pack_jobs = None
        self.param_epoch = 0
        self.generation = 0
        self.saved_generation = -1
        self.flat = None
        self.flat_grad = None
        self.offsets = None
        self.views = {}
        self.grad_views = {}
        self._dcn_slots, self._dcn_E = {}, {}
        self._plans = {}
        self.dcn_forms = {'gather': 0, 'scatter': 0}
        self.grad_scale = 1.0
        self._live = None
        self.nbt_pending = {}
        self._bias_done = set()
        self._bnb_sums = {}     # bn name -> (partials [2][C][rows], rows) produced by the data gradient that wrote its dout
        self._reduce_pending = []    # deferred slab reductions of this backward pass: (key, job record)
        self._wgrad_fifo = []        # weight gradients queued for a late launch (late_wgrad)
        self._wino_src = {}          # Winograd panel key -> (parameter, panel, O, I, cs, dgrad)
        self._wino_jobs = None
        self._reduce_job_cache = {}  # (layer, phase-2 arguments) -> dbn_wgrad_reduce_job
        self._reduce_tables = {}     # tuple of keys -> device job table of one grouped launch
        self._by_ptr = {}       # data_ptr -> activation buffer (to find the pre-split planes of an operand)
        self._plane_cache = {}  # data_ptr -> (planes, generation)
        # 'bf16x3' mode, optional (DBN_PRESPLIT=1 / engine.presplit): MFMA operands read from pre-split bf16 planes (made once per
        # tensor and step by dbn_split3) instead of being split when staged.  Bit-identical results — and SLOWER (520 vs 634
        # images/s): the planes are three separate tensors, so a row's k-tile arrives as 6 pieces of 16 B from 3 distant lines
        # instead of one 64-byte run, and the gather, not the split arithmetic, is what bounds these kernels (igemm<128,64> 382 vs
        # 273 us; 165 us would be MFMA-bound).  Kept as an option; an interleaved [.., C/16][3][16] plane layout is the next step.
        self.presplit = os.environ.get('DBN_PRESPLIT', '0') == '1' and bool(self.L.dbn_has_experiments())  # (EXP=1 builds only)
        self.prof = None  # optional KernelTimer
        self.grad_ready_hook = None  # optional callable(stage): a contiguous part of the flat gradient buffer is final (train.GRAD_STAGES)
        self.ns = 0  # conv math: 0 = exact-fp32 MFMA, 3 = fp32-accurate bf16x3 split, 1 = one 16-bit plane
        self.at = 0  # activation storage in HBM: 0 = fp32, 1 = bf16, 2 = fp16 (inference only); see set_conv_math
        # Weight gradients (+ their slab reductions and bias column sums) run on a second HIP stream: they only feed the
        # optimizer, so they are ordered behind the producer of dy and otherwise free.  Two MFMA kernels with different
        # register / LDS footprints co-resident on a CU keep the matrix pipe busier than either alone (occupancy 3 each) and
        # fill each other's ramps and tails: 36.6 -> 34.9 ms/step at bs16 640^2.  (Tried and rejected: letting the weight
        # gradient overlap only the HBM-bound BatchNorm backward of the next layer — slower than a single stream.)
        self.overlap_wgrad = True
        # forward: the threshold branch of the head beside the binarize branch on the second stream?  Two MFMA-bound kernels
        # sharing the matrix pipe is zero-sum (measured: two conv streams take exactly 2x each); what the overlap hides is the few
        # HBM-bound kernels in between: 31.9 vs 32.05 ms/step, at the price of every launch of the two big 256->64 convs running
        # at half speed.  Off: the kernels run undisturbed (the dominant kernel's timed-region rate 76 -> 94 TFLOP/s).
        self.overlap_head_branches = os.environ.get('DBN_OVERLAP_HEAD', '0') == '1'
        self.side_priority = None  # HIP stream priority of the side stream (None: default)
        self._side = None
        self._side_used = False
        self._slab_free = [None, None]  # per slab scratch: event of the reduction that last read it

    # ------------------------------------------------------------------ memory
    @property
    def live_params(self):
        if self._live is None:
            self._live = [(n, p) for n, p in self.model.named_parameters() if not n.startswith(DEAD_PREFIXES)]
        return self._live

    def ensure_flat(self):
        params = self.live_params
        dev = params[0][1].device
        if dev.type != 'cuda':
            raise RuntimeError('DBTextModel parameters must live on a HIP device (model.to("cuda")); '
                               'there is no CPU path')
        if self.flat is not None and self.flat.device == dev:
            base = self.flat.data_ptr()
            if all(p.data_ptr() == base + 4 * off for (_, p), off in zip(params, self.offsets)):
                return
        offs, total = flat_layout([p.numel() for _, p in params])
        flat = torch.zeros(total, device=dev, dtype=torch.float32)
        grad = torch.zeros(total, device=dev, dtype=torch.float32)
        self.views, self.grad_views = {}, {}
        for (n, p), off in zip(params, offs):
            if p.dtype != torch.float32:
                raise RuntimeError('fp32 parameters expected, got %s for %s' % (p.dtype, n))
            v = flat[off:off + p.numel()].view(p.shape)
            v.copy_(p.data)
            p.data = v
            self.views[n] = v
            self.grad_views[n] = grad[off:off + p.numel()].view(p.shape)
        self.flat, self.flat_grad, self.offsets = flat, grad, offs
        self.bufs, self.packs, self.pack_src, self._pack_jobs = {}, {}, {}, None
        self._by_ptr, self._plane_cache = {}, {}
        self._reduce_pending, self._reduce_job_cache, self._reduce_tables = [], {}, {}
        self.param_epoch += 1

    def flush_counters(self):
        """BatchNorm `num_batches_tracked` is bookkeeping only (momentum is fixed); it is
        counted on the host and written to the buffers when the state is read."""
        if not self.nbt_pending:
            return
        mods = dict(self.model.named_modules())
        for name, cnt in self.nbt_pending.items():
            mods[name].num_batches_tracked += cnt
        self.nbt_pending = {}

    MATH_MODES = {'f32': (0, 0), 'bf16x3': (3, 0), 'bf16c': (1, 0), 'bf16': (1, 1), 'fp16': (1, 2)}  # name -> (ns, at)

    def set_conv_math(self, mode):
        """Precision mode of the whole path.
        'f32' (default, BASELINE configs[1]): fp32 tensors, v_mfma_f32_32x32x2_f32 (bit-exact fp32 products).
        'bf16x3': fp32 tensors, every operand split exactly into three bf16 terms, six bf16 MFMAs per product group, fp32
                  accumulate (fp32-accurate).
        'bf16'  (BASELINE configs[2]/[3]): NATIVE bf16 — activations, their gradients and the weight panels are stored in
                  bf16 in HBM (every HBM-bound kernel moves half the bytes), bf16 MFMA with fp32 accumulation; fp32 master
                  weights, BatchNorm statistics (taken from the fp32 accumulators), loss sums, weight gradients, Adam state.
        'fp16'  (BASELINE configs[4]): NATIVE fp16 inference — eval-mode forward only; fp16 activations and weight panels,
                  v_mfma_f32_32x32x16_f16, fp32 accumulation; the output maps are fp32 (what postprocess.py consumes).
        'bf16c': the round-1 compute-only mode (fp32 tensors in HBM, operands rounded to bf16 when staged)."""
        ns, at = self.MATH_MODES[mode]
        if at != self.at:  # other storage type: every activation buffer and weight panel is stale
            self.bufs, self.packs, self.pack_src, self._pack_jobs = {}, {}, {}, None
            self._by_ptr, self._plane_cache = {}, {}
            self._reduce_pending, self._reduce_job_cache, self._reduce_tables = [], {}, {}
            self.saved_generation = -1
        self.ns, self.at = ns, at
        self.math_mode = mode

    math_mode = 'f32'

    @property
    def kind(self):
        """weight-panel kind: 0 fp32, 1 bf16, 3 bf16x3, 2 fp16"""
        return 2 if self.at == 2 else self.ns

    def mark_params_dirty(self):
        self.param_epoch += 1

    def buf(self, name, *shape, dtype=None):
        """Persistent ACTIVATION buffer (stored in the engine's activation type)."""
        dtype = ACT_DTYPES[self.at] if dtype is None else dtype
        t = self.bufs.get(name)
        dev = self.flat.device
        if t is None or tuple(t.shape) != tuple(shape) or t.device != dev or t.dtype != dtype:
            t = device_empty(shape, dev, dtype)
            self.bufs[name] = t
            self._by_ptr[t.data_ptr()] = t
        return t

    def _drop_buf(self, name):
        """Forget an activation buffer (apply-on-load: the tensor is never written) — including the pointer-keyed views of it, which
        would otherwise keep the old allocation alive across train / eval alternations that disagree on lazy_act()."""
        t = self.bufs.pop(name, None)
        if t is not None:
            self._by_ptr.pop(t.data_ptr(), None)
            self._plane_cache.pop(t.data_ptr(), None)

    def fbuf(self, name, *shape):
        """Persistent fp32 buffer (coefficients, statistics, maps, parameter-shaped temporaries)."""
        return self.buf(name, *shape, dtype=torch.float32)

    def scratch(self, name, numel):
        if self._in_side:  # launches on the side stream run concurrently with the main one: private scratch
            name += '#side'
        t = self.bufs.get(name)
        if t is None or t.numel() < numel or t.device != self.flat.device:
            t = device_empty(int(numel), self.flat.device)
            self.bufs[name] = t
        return t

    @property
    def stream(self):
        return torch.cuda.current_stream(self.flat.device).cuda_stream

    def reduce_ws(self):
        # one scratch per stream: the side (weight-gradient) stream reduces bias gradients concurrently
        name = '_reduce_ws_side' if self._in_side else '_reduce_ws'
        return self.scratch(name, self.L.dbn_reduce_ws_floats(2048))  # widest BatchNorm: 512 (resnet18) / 2048 (resnet50)

    _in_side = False

    @contextlib.contextmanager
    def side_stream(self):
        """Run the enclosed launches on the side stream, ordered after everything enqueued so far on
        the current stream (the producer of dy).  Weight gradients only feed the optimizer, so they
        overlap the HBM-bound BatchNorm-backward / dgrad chain; `join_side()` re-joins the streams."""
        if not self.overlap_wgrad or (self.prof is not None and self.prof.labels is None) or self._in_side:
            yield  # single-stream mode, full profiling (serialises), or already on the side stream (re-entrant)
            return
        if self._side is None or self._side.device != self.flat.device:
            self._side = (torch.cuda.Stream(device=self.flat.device) if self.side_priority is None else
                          torch.cuda.Stream(device=self.flat.device, priority=self.side_priority))
        main = torch.cuda.current_stream(self.flat.device)
        self._side.wait_stream(main)
        self._side_used = True
        self._in_side = True
        try:
            with torch.cuda.stream(self._side):
                yield
        finally:
            self._in_side = False

    def _side_waits_for_reductions(self):
        """Before a gradient bucket is announced from the side stream its reductions (third stream) must have been ordered in."""
        if self._side2_used and self._in_side:
            torch.cuda.current_stream(self.flat.device).wait_stream(self._side2)

    def join_side(self):
        if self._side2_used:  # the reduction stream (see reduce_stream) joins first
            torch.cuda.current_stream(self.flat.device).wait_stream(self._side2)
            self._side2_used = False
        if self._side_used:
            torch.cuda.current_stream(self.flat.device).wait_stream(self._side)
            self._side_used = False

    # ------------------------------------------------------------ weight panels
    def pack(self, name, w, mode, stride=1, version=None, cs=0):
        """Weight panels of `w` for (mode, stride), cached until `w` changes.  `version` replaces w._version for tensors
        that are rewritten through raw pointers (the FPN's combined weights).  cs: channels of the source tensor (mode 0)."""
        if self._repack_pending and name != self.STEM:  # (first use of a panel the side stream is still rebuilding)
            self._join_repack()
        ns = self.kind
        key = (name, mode, stride, ns)
        ent = self.packs.get(key)
        stamp = (w._version if version is None else version, self.param_epoch, w.data_ptr())
        if ent is not None and ent[1] == stamp:
            return ent[0]
        O, I, R, S = w.shape
        cs = cs if mode == 0 else 0
        if version is None:  # a parameter (not a derived tensor): remembered for the one-launch repack of later steps
            self.pack_src[key] = (w, cs if cs else (I + 3) // 4 * 4)
        n = self.L.dbn_igemm_panel_floats_t(ns, O, I, R, S, mode, stride, cs)
        out = ent[0] if ent is not None else device_empty(n, w.device)
        check(self.L.dbn_pack_weights_t(ns, w.data_ptr(), O, I, R, S, mode, stride, cs, out.data_ptr(), self.stream), 'pack_weights')
        self.packs[key] = (out, stamp)
        return out

    def _planes(self, t):
        """The pre-split form [3][...] (bf16) of the fp32 activation tensor `t`, made once per step on first use."""
        key = t.data_ptr()
        ent = self._plane_cache.get(key)
        if ent is not None and ent[1] == self.generation:
            return ent[0]
        # (inside a side-stream region this runs on the side stream: fine for tensors produced there — the main stream only sees
        # them after join_side(); operands that BOTH streams read are pre-split on the main stream first, see _presplit)
        pl = self.buf('#planes/%x' % key, 3, *t.shape, dtype=torch.bfloat16)
        check(self.L.dbn_split3(t.data_ptr(), pl.data_ptr(), t.numel(), self.stream), 'split3')
        self._plane_cache[key] = (pl, self.generation)
        return pl

    @property
    def _use_planes(self):
        return self.at == 0 and self.ns == 3 and self.presplit

    def _presplit(self, *tensors):
        """Make sure the planes of these operands exist (on the CURRENT stream) before work is handed to the side stream."""
        if self._use_planes:
            for t in tensors:
                if t is not None and t.dtype == torch.float32 and t.numel() % 4 == 0:
                    self._planes(t)

    def _src(self, ptr, C):
        """(activation type, pointer) of the source operand of a conv: its pre-split planes in the bf16x3 mode."""
        if self._use_planes and C % 16 == 0:
            t = self._by_ptr.get(ptr)
            if t is not None and t.dtype == torch.float32 and t.data_ptr() == ptr:
                return 3, self._planes(t).data_ptr()
        return self.at, ptr

    fuse_bn_bwd_sums = True  # data gradients also reduce the two sums of the BatchNorm backward that consumes their output
    bnb_finalize_in_kernel = os.environ.get('DBN_BNB_FINAL', '0') == '1'  # ... and fold them to the per-channel results themselves (last-arriver, fixed order).  OFF since round 6 (DBN_BNB_FINAL=1: on): the per-workgroup hand-over costs more than the finalize launches it removes — one box, interleaved: fp32 725.9 / 726.9 -> 729.8 / 730.9 images/s, bf16 1708.7 -> 1780.5 (the same finding as for the statistics rows, bn_final_in_kernel)

    def _bnb_eligible(self, args):
        """Can this igemm call (dbn_igemm_f32 argument list) carry the BatchNorm-backward sums of its consumer?"""
        N, Hs, Ws, Cs, Hd, Wd, Cd, R, S, stride, pad, mode = args[4:16]
        if not self.fuse_bn_bwd_sums or self._use_planes or not ((self.at == 0 and self.ns in (0, 1, 3)) or (self.at == 1 and self.ns == 1)):
            return False
        if mode == 1 and stride > 1 and (R < stride or S < stride):  # a parity class without taps: pixels the launch never visits
            return False
        if self.prof is not None and self.prof.labels is None and not self.prof_fused:
            return False
        return not (self.splitk and self.L.dbn_igemm_splitk_plan_ns(N * Hd * Wd, Cd, R * S * Cs, Cs, self.ns) > 1)

    prof_fused = True

    def _igemm(self, what, *args, consumer=None):
        """args = the dbn_igemm_f32 argument list without the trailing stream.
        consumer: (bn_name, y, zmask) — the BatchNorm whose output gradient this call is the LAST writer of (zmask None: ReLU
        directly on that BatchNorm, mask recomputed from y).  When the call is eligible its epilogue also produces that
        BatchNorm-backward's two per-channel sums; they wait in self._bnb_sums[bn_name] for bn_backward()."""
        N, Hs, Ws, Cs, Hd, Wd, Cd, R, S, stride, pad, mode = args[4:16]
        at, srcp = self._src(args[0], Cs)
        if consumer is not None and self._bnb_eligible(args):
            bn_name, y, zmask = consumer[:3]
            second = consumer[3] if len(consumer) > 3 else None  # (bn_name2, y2): a second BatchNorm over the same dz and mask
            assert tuple(y.shape) == (N, Hd, Wd, Cd) and (zmask is None or zmask.shape == y.shape), (what, bn_name)
            hint = args[17]
            rows = self.L.dbn_igemm_bn_rows(at, self.ns, N, Hs, Ws, Cs, Hd, Wd, Cd, R, S, stride, pad, mode, hint)
            part = self.fbuf(bn_name + '/bnb_part', 2 * Cd * rows)
            msc = msh = None
            if zmask is None:
                msc, msh = self.bufs[bn_name + '/scale'], self.bufs[bn_name + '/shift']
            y2 = mean2 = rstd2 = part2 = None
            if second is not None:
                assert zmask is not None and second[1].shape == y.shape
                y2, mean2, rstd2 = second[1], self.bufs[second[0] + '/mean'], self.bufs[second[0] + '/rstd']
                part2 = self.fbuf(second[0] + '/bnb_part', 2 * Cd * rows)
            fin = None
            if self.bnb_finalize_in_kernel:
                # the data gradient's last workgroups also fold the partial rows: no finalize launch between it and the apply pass
                G = self.grad_views
                cnt = self.bufs.get(bn_name + '/bnb_cnt')
                ncnt = self.L.dbn_igemm_bn_final_counters(rows, Cd)
                if cnt is None or cnt.numel() != ncnt or cnt.device != self.flat.device:
                    cnt = torch.zeros(ncnt, device=self.flat.device, dtype=torch.int32)  # (the kernels leave them zero)
                    self.bufs[bn_name + '/bnb_cnt'] = cnt
                grp = self.fbuf(bn_name + '/bnb_grp', self.L.dbn_igemm_bn_final_group_floats(rows, Cd))
                c1c2 = self.fbuf(bn_name + '/bnb_c1c2', 2 * Cd)
                fin = _lib.BnbFinal(cnt.data_ptr(), grp.data_ptr(), c1c2.data_ptr(), G[bn_name + '.weight'].data_ptr(),
                                    G[bn_name + '.bias'].data_ptr(), None, None, None, self.grad_scale)
                if second is not None:
                    c1c2b = self.fbuf(second[0] + '/bnb_c1c2', 2 * Cd)
                    fin.c1c2_2, fin.dgamma_2, fin.dbeta_2 = (c1c2b.data_ptr(), G[second[0] + '.weight'].data_ptr(),
                                                             G[second[0] + '.bias'].data_ptr())
            import ctypes
            check(self.L.dbn_igemm_bnsums_t(at, self.ns, srcp, *args[1:], y.data_ptr(), _p(zmask), _p(msc), _p(msh),
                                            self.bufs[bn_name + '/mean'].data_ptr(), self.bufs[bn_name + '/rstd'].data_ptr(),
                                            part.data_ptr(), _p(y2), _p(mean2), _p(rstd2), _p(part2),
                                            ctypes.byref(fin) if fin is not None else None, self.stream), what)
            self._bnb_sums[bn_name] = (c1c2, -1) if fin is not None else (part, rows)
            if second is not None:
                self._bnb_sums[second[0]] = (c1c2b, -1) if fin is not None else (part2, rows)
            return
        ks, slab = 1, None
        if self.splitk and (mode == 0 or stride == 1):
            ks = self.L.dbn_igemm_splitk_plan_ns(N * Hd * Wd, Cd, R * S * Cs, Cs, self.ns)
            if ks > 1:  # few output tiles, long reduction: split K over workgroup rows, fixed-order slab sum
                slab = self.scratch('_splitk_slab', self.L.dbn_igemm_splitk_slab_floats(ks, N, Hd, Wd, Cd))  # (slabs are padded apart: HBM channel rotation)
        check(self.L.dbn_igemm_t(at, self.ns, srcp, *args[1:], ks, _p(slab), self.stream), what)

    batched_repack = True
    # ... and that launch (HBM-bound, 0.13 ms) runs on the side stream beside the stem conv, whose own panel is packed lazily on
    # the main stream; the main stream waits for it after the stem (forward() -> _join_repack)
    overlap_repack = True
    _repack_pending = False
    STEM = 'backbone.conv1'

    def _join_repack(self):
        if self._repack_pending:
            assert not self._in_side
            self.join_side()
            self._repack_pending = False

    def repack_params(self):
        """After an optimizer step every weight panel is stale.  Instead of ~80 dbn_pack_weights launches sprinkled over the
        next step (one before each conv's first use), all panels that exist already are rebuilt by ONE launch here."""
        if not self.batched_repack:
            return
        ns = self.kind
        beside = self.overlap_repack and self.overlap_wgrad
        stale = []
        for key, (w, cs) in self.pack_src.items():
            if key[3] != ns or (beside and key[0] == self.STEM):
                continue
            ent = self.packs.get(key)
            stamp = (w._version, self.param_epoch, w.data_ptr())
            if ent is not None and ent[1] != stamp:
                stale.append((key, w, ent[0], stamp, cs))
        if len(stale) < 8:  # first step (nothing packed yet) or nothing changed: the lazy path handles it
            return
        sig = tuple((k, w.data_ptr(), out.data_ptr()) for k, w, out, _, _ in stale)
        if self._pack_jobs is None or self._pack_jobs[0] != sig:
            import ctypes

            class Job(ctypes.Structure):
                _fields_ = [('w', ctypes.c_void_p), ('out', ctypes.c_void_p)] + [(f, ctypes.c_int) for f in
                                                                                  ('O', 'I', 'R', 'S', 'mode', 'Cs', 'Cd', 'f')]
            arr = (Job * len(stale))()
            for i, ((name, mode, stride, _), w, out, _, cs) in enumerate(stale):
                O, I, R, S = w.shape
                arr[i] = Job(w.data_ptr(), out.data_ptr(), O, I, R, S, mode, cs if mode == 0 else O,
                             O if mode == 0 else I, stride if (mode == 1 and stride > 1) else 1)
            host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
            self._pack_jobs = (sig, host.to(self.flat.device), len(stale))
        _, table, n = self._pack_jobs
        if beside:
            with self.side_stream():
                check(self.L.dbn_pack_weights_batched(table.data_ptr(), n, ns, self.stream), 'pack_weights_batched')
            self._repack_pending = True
        else:
            check(self.L.dbn_pack_weights_batched(table.data_ptr(), n, ns, self.stream), 'pack_weights_batched')
        for key, _, out, stamp, _ in stale:
            self.packs[key] = (out, stamp)
        self._repack_winograd(beside)

    def _repack_fpn(self):
        """The pyramid conv's combined weights (4 launches) and every panel derived from them — forward, data-gradient, Winograd
        forms: 9 pack launches — on the SIDE stream beside layer1 (called once the stem's repack join is behind: the side stream is
        idle until the first lateral conv, and the main stream joins it again before the pyramid conv), instead of lazily in front of
        their first use on the MAIN stream: there they sat on the critical path, and the data-gradient panels' pack, a small grid, took
        0.44 ms beside the weight-gradient stream (round-4 trace: 0.6 ms of the 22.7 ms step).  (Together with the other panels right
        after the optimizer step was measured first: 700 -> 692 images/s — the first conv behind the stem then waits for all of it.)"""
        src = self._fpn_src
        beside = self.overlap_repack and self.overlap_wgrad
        on = self.fpn_repack_early == '1' or (self.fpn_repack_early == '' and self.at != 0)
        if not on or src is None or not beside or self._in_side:
            return
        name, conv, Cg = src
        ent = self.packs.get((name, 'combined'))
        w = conv.weight
        if ent is None or ent[1] == (w._version, self.param_epoch, w.data_ptr()):
            return
        derived = [k for k in self.packs if isinstance(k[0], str) and k[0].startswith(name + '#') and '#fold' not in k[0]]

        def run():
            wds, wver = self._fpn_combined_weights(name, conv, Cg)
            for k in derived:
                g = int(k[0][-1])  # '#f<g>' forward / '#g<g>' data-gradient panels of level g, '#lv0' / '#g0': level 0
                if k[1] == 'winograd':
                    self._winograd_panel(k[0], wds[g], k[2], dgrad=k[3], version=wver)
                elif k[3] == self.kind:
                    self.pack(k[0], wds[g], k[1], k[2], version=wver)
        with self.side_stream():
            run()

    # measured on one box, interleaved: exact fp32 702.9-704.7 images/s without vs 698.0-701.9 with (the step there is bound by the two
    # streams' total work, not by the main stream's chain); bf16 1600 / 1613 without vs 1628 / 1622 with: '' = in the 16-bit modes only
    fpn_repack_early = os.environ.get('DBN_FPN_REPACK_EARLY', '')
    _fpn_src = None

    def _repack_winograd(self, beside):
        """The Winograd panels (G g G^T of every 3x3 / stride-1 filter, forward and data-gradient form) of all layers in ONE launch
        after the optimizer step — 31 launches of ~10 us sprinkled over the step otherwise (0.34 ms in the round-4 trace)."""
        items = [(k, v) for k, v in self._wino_src.items() if k in self.packs]
        if len(items) < 4:
            return
        sig = tuple((k, v[0].data_ptr(), v[1].data_ptr()) for k, v in items)
        if self._wino_jobs is None or self._wino_jobs[0] != sig:
            import ctypes

            class Job(ctypes.Structure):
                _fields_ = [('w', ctypes.c_void_p), ('out', ctypes.c_void_p)] + [(f, ctypes.c_int) for f in ('O', 'I', 'Cs', 'dgrad')]
            arr = (Job * len(items))()
            for i, (_, (w, out, O, I, cs, dgrad)) in enumerate(items):
                arr[i] = Job(w.data_ptr(), out.data_ptr(), O, I, cs, dgrad)
            self._wino_jobs = (sig, torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(self.flat.device), len(items))
        _, table, n = self._wino_jobs
        if beside:
            with self.side_stream():
                check(self.L.dbn_winograd_pack_batched(table.data_ptr(), n, self.stream), 'winograd_pack_batched')
            self._repack_pending = True
        else:
            check(self.L.dbn_winograd_pack_batched(table.data_ptr(), n, self.stream), 'winograd_pack_batched')
        for k, (w, out, *_r) in items:
            self.packs[k] = (out, (w._version, self.param_epoch, w.data_ptr()))

    # ------------------------------------------------------------------ kernels
    # 3x3 / stride-1 / pad-1 FORWARD convolutions of the exact-fp32 path through Winograd F(2x2, 3x3) (csrc/winograd_f32.hip): fp32
    # arithmetic, 2.25x fewer matrix FLOPs; the result equals the direct convolution up to fp32 rounding (another summation order).
    winograd = os.environ.get('DBN_WINOGRAD', '1') == '1'

    # ---- per-layer plan (round 5, review item 9): WHICH kernel family and fusion form a conv takes at one (geometry, input shape, conv math,
    # mode) is decided in ONE place and cached; the call sites (_winograd_ok, lazy_act, wgrad, _fold_ok, conv_bn_act_eval) read it.  The key
    # holds every engine switch a decision depends on, so A/B switches flipped at run time (tests do) take effect at once.
    def plan(self, conv, shape, train=False, fp32_tensors=True):
        N, H, W, C = shape
        dcn = bool(getattr(conv, 'with_dcn', False))
        key = (conv.cin, conv.cout, conv.k, conv.stride, conv.padding, dcn, N, H, W, C, bool(train), bool(fp32_tensors), self.at, self.ns,
               self.winograd, self.winograd_wgrad, self.fold_eval_bn, self._use_planes, self.splitk, self.pw16, self.apply_on_load)
        pl = self._plans.get(key)
        if pl is None:
            L = self.L
            k, s_, p_ = conv.k, conv.stride, conv.padding
            Ho, Wo = (H + 2 * p_ - k) // s_ + 1, (W + 2 * p_ - k) // s_ + 1
            pl = ConvPlan()
            exact = self.ns == 0 and self.at == 0 and fp32_tensors
            s1 = (k, s_, p_) == (3, 1, 1)
            # exact fp32, 3x3 / stride 1: Winograd F(2x2, 3x3) forward / data gradient (csrc/winograd_f32.hip) ...
            pl.winograd = bool(self.winograd and exact and s1 and L.dbn_winograd_eligible(N, H, W, C, conv.cout))
            # ... and weight gradient (csrc/winograd_wgrad_f32.hip)
            pl.winograd_wgrad = bool(self.winograd_wgrad and exact and s1 and L.dbn_winograd_wgrad_eligible(N, H, W, conv.cout, C, conv.cin))
            # relu(bn(.)) of this conv's INPUT applied while the patches are staged (the activation tensor is never written): the conv and —
            # in training — its weight gradient must both be Winograd launches
            pl.apply_on_load = bool(self.apply_on_load and not self._use_planes and not dcn and pl.winograd and (not train or pl.winograd_wgrad)
                                    and not (not train and self.fold_eval_bn))
            # inference: conv -> BatchNorm -> (+ residual) -> ReLU as ONE launch on weights with the running statistics folded in
            pl.fold = bool(not train and self.fold_eval_bn and not self._use_planes and not dcn
                           and not (self.splitk and L.dbn_igemm_splitk_plan_ns(N * Ho * Wo, conv.cout, k * k * C, C, self.ns) > 1))
            # ... of a pointwise conv from 64 channels on 16-bit storage: the ConvT kernel's construction (csrc/convt16.hip)
            pl.pw16 = bool(pl.fold and self.pw16 and self.at != 0 and (k, s_, p_) == (1, 1, 0) and C == conv.cin
                           and L.dbn_pw16_eligible(self.at, N, H, W, C, conv.cout))
            self._plans[key] = pl
        return pl

    def _winograd_ok(self, x, conv):
        return self.plan(conv, x.shape, fp32_tensors=x.dtype == torch.float32).winograd

    def _winograd_panel(self, name, w, cs, dgrad=0, version=None):
        """G g G^T of every filter (dgrad: of the rotated / transposed filters of the data gradient), re-made when the parameter
        changed (one small launch per layer and step).  version: as in pack() — stamp of a derived tensor rewritten through raw pointers."""
        if self._repack_pending:  # (first use of a panel the side stream is still rebuilding)
            self._join_repack()
        key = (name, 'winograd', cs, dgrad)
        ent = self.packs.get(key)
        stamp = (w._version if version is None else version, self.param_epoch, w.data_ptr())
        if ent is not None and ent[1] == stamp:
            return ent[0]
        O, I = (w.shape[1], w.shape[0]) if dgrad else (w.shape[0], w.shape[1])  # channels out of / into THIS conv
        out = ent[0] if ent is not None else device_empty(self.L.dbn_winograd_panel_floats(O, cs), w.device)
        if version is None:  # a parameter: remembered for the one-launch refresh after the optimizer step (repack_params)
            self._wino_src[key] = (w, out, O, I, cs, dgrad)
        check(self.L.dbn_winograd_pack(w.data_ptr(), O, I, cs, dgrad, out.data_ptr(), self.stream), 'winograd pack ' + name)
        self.packs[key] = (out, stamp)
        return out

    def _winograd_conv(self, name, x, conv, y, bn=None, bn_name=None, x_act=None):
        """x_act = (scale, shift): the conv's input is relu(x * scale + shift), applied while the kernel stages its patches (x is the
        INPUT of the BatchNorm + ReLU in front of the conv; apply_on_load)."""
        N, H, W, C = x.shape
        up = self._winograd_panel(name, conv.weight, C)
        asc, ash = (x_act[0].data_ptr(), x_act[1].data_ptr()) if x_act is not None else (None, None)
        if self.prof:
            # FLOPs the MFMA pipe EXECUTES: 16 products per 2 x 2 output tile and channel pair (the direct form's 36 are what
            # `step_tflops` counts): the roofline fraction of this kernel is matrix-pipe utilisation, not an effective rate
            # (algorithmic bytes: source + destination once — bench.py: roofline.traffic_over_algorithmic)
            self.prof.begin('winograd_f32_kernel', 2.0 * N * H * W * conv.cout * conv.cin * 4, 4.0 * N * H * W * (C + conv.cout), 'fwd ' + name)
        if bn is None:
            check(self.L.dbn_winograd_conv_bn_act_f32(x.data_ptr(), asc, ash, up.data_ptr(), _p(conv.bias), y.data_ptr(), N, H, W, C, conv.cout,
                                                      None, None, 0.0, 0.0, None, None, None, None, None, None, None, self.stream), 'winograd ' + name)
            sc = sh = None
        else:
            Co = conv.cout
            sc, sh = self.fbuf(bn_name + '/scale', Co), self.fbuf(bn_name + '/shift', Co)
            mu, rs = self.fbuf(bn_name + '/mean', Co), self.fbuf(bn_name + '/rstd', Co)
            ws = self.scratch('_conv_bn_ws', self.L.dbn_winograd_ws_floats(N, H, W, Co))
            self._announce_bn_final(bn_name, self.L.dbn_winograd_rows(N, H, W), Co)
            check(self.L.dbn_winograd_conv_bn_act_f32(x.data_ptr(), asc, ash, up.data_ptr(), _p(conv.bias), y.data_ptr(), N, H, W, C, Co,
                                                      bn.weight.data_ptr(), bn.bias.data_ptr(), bn.eps, bn.momentum,
                                                      bn.running_mean.data_ptr(), bn.running_var.data_ptr(), sc.data_ptr(), sh.data_ptr(),
                                                      mu.data_ptr(), rs.data_ptr(), ws.data_ptr(), self.stream), 'winograd+bn ' + name)
            self.nbt_pending[bn_name] = self.nbt_pending.get(bn_name, 0) + 1
        if self.prof:
            self.prof.end()
        return sc, sh

    def conv_fwd(self, name, x, conv, out_name, version=None, x_act=None):
        N, H, W, C = x.shape
        k, s, p = conv.k, conv.stride, conv.padding
        Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
        assert C >= conv.cin and C % 4 == 0, (name, C, conv.cin)
        if version is None and self._winograd_ok(x, conv):
            y = self.buf(out_name, N, Ho, Wo, conv.cout)
            self._winograd_conv(name, x, conv, y, x_act=x_act)
            return y
        assert x_act is None, name  # (only the Winograd kernel applies an activation on load: lazy_act() checks before it defers)
        wpk = self.pack(name, conv.weight, 0, version=version, cs=C)
        y = self.buf(out_name, N, Ho, Wo, conv.cout)
        if self.prof:
            self._prof_igemm(N * Ho * Wo, conv.cout, 2.0 * N * Ho * Wo * conv.cout * conv.cin * k * k, 'fwd ' + name, 0,
                             (N, H, W, C, Ho, Wo, k, s, p))
        self._igemm('igemm fwd ' + name, x.data_ptr(), wpk.data_ptr(), _p(conv.bias), y.data_ptr(), N, H, W, C, Ho, Wo, conv.cout, k,
                    k, s, p, 0, 0, 0)
        if self.prof:
            self.prof.end()
        return y

    splitk = True  # split the reduction of convs with few output tiles (dbn_igemm_splitk_plan)
    fuse_bn_stats = True  # accumulate train-mode BN statistics in the conv epilogue (no separate statistics pass)
    # round 6: ... and fold the per-tile rows in the conv's own last workgroups (dbn_conv_bn_set_final): no bn_finalize_tiles_kernel launch —
    # 7 us of kernel + ~6 us of dependent-dispatch gap, 32 times a step — between a conv and whatever consumes its BatchNorm's coefficients.
    # Built, tested (tests/test_bn_final_gpu.py), measured and OFF: 277 -> 245 launches per fp32 step, but 711.4 / 711.9 images/s against
    # 726.8 / 723.7 with the finalize kernels (one box, interleaved): every workgroup of every conv pays the hand-over (its row stores drained,
    # an atomic round trip, two barriers: ~2-3 us at the end of a 20-70 us lifetime, six rounds per launch) to save one ~13 us launch.
    # DBN_BN_FINAL=1 turns it on.
    bn_final_in_kernel = os.environ.get('DBN_BN_FINAL', '0') == '1'

    def _announce_bn_final(self, bn_name, rows, C):
        """Counters (zero once: the kernels leave them zero) and group scratch of this BatchNorm's in-kernel statistics finalize."""
        if not self.bn_final_in_kernel or rows <= 0:
            return
        ncnt = self.L.dbn_igemm_bn_final_counters(rows, C)
        cnt = self.bufs.get(bn_name + '/bnf_cnt')
        if cnt is None or cnt.numel() != ncnt or cnt.device != self.flat.device:
            cnt = torch.zeros(ncnt, device=self.flat.device, dtype=torch.int32)
            self.bufs[bn_name + '/bnf_cnt'] = cnt
        ng = self.L.dbn_conv_bn_final_group_doubles(rows, C)
        grp = self.bufs.get(bn_name + '/bnf_grp')
        if grp is None or grp.numel() != ng or grp.device != self.flat.device:
            grp = torch.empty(ng, device=self.flat.device, dtype=torch.float64)
            self.bufs[bn_name + '/bnf_grp'] = grp
        check(self.L.dbn_conv_bn_set_final(cnt.data_ptr(), grp.data_ptr()), 'conv_bn_set_final')

    def _conv_bn_call(self, what, bn_name, bn, y, args, mode, stride, accumulate=0):
        """args: dbn_igemm_f32's arguments up to and including `mode` (without accumulate / tile_hint / stream)."""
        C = y.shape[-1]
        sc, sh = self.fbuf(bn_name + '/scale', C), self.fbuf(bn_name + '/shift', C)
        mu, rs = self.fbuf(bn_name + '/mean', C), self.fbuf(bn_name + '/rstd', C)
        N, Hd, Wd = y.shape[0], y.shape[1], y.shape[2]
        ws = self.scratch('_conv_bn_ws', self.L.dbn_conv_bn_ws_floats(N, Hd, Wd, C, mode, stride))
        at, srcp = self._src(args[0], args[7])
        # args = (src, wpk, bias, dst, N, Hs, Ws, Cs, Hd, Wd, Cd, R, S, stride, pad, mode)
        self._announce_bn_final(bn_name, self.L.dbn_igemm_bn_rows(at, self.ns, *args[4:15], mode, 0), C)
        check(self.L.dbn_conv_bn_t(at, srcp, *args[1:], accumulate, 0, self.ns, bn.weight.data_ptr(), bn.bias.data_ptr(), bn.eps, bn.momentum,
                                     bn.running_mean.data_ptr(), bn.running_var.data_ptr(), sc.data_ptr(), sh.data_ptr(),
                                     mu.data_ptr(), rs.data_ptr(), ws.data_ptr(), self.stream), what)
        self.nbt_pending[bn_name] = self.nbt_pending.get(bn_name, 0) + 1
        return sc, sh

    def conv_bn(self, name, x, conv, out_name, bn_name, bn, train, version=None, x_act=None):
        """conv -> BatchNorm coefficients.  Train mode: one fused call (statistics in the conv epilogue)."""
        N, H, W, C = x.shape
        k, s, p = conv.k, conv.stride, conv.padding
        Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
        wino = version is None and self._winograd_ok(x, conv)
        split = self.splitk and self.L.dbn_igemm_splitk_plan_ns(N * Ho * Wo, conv.cout, k * k * C, C, self.ns) > 1
        if not (train and self.fuse_bn_stats) or split:  # split-K convs take their statistics in a (small) separate pass
            y = self.conv_fwd(name, x, conv, out_name, version=version, x_act=x_act)
            sc, sh = self.bn_coef(bn_name, bn, y, train)
            return y, sc, sh
        assert C >= conv.cin and C % 4 == 0, (name, C, conv.cin)
        if wino:
            y = self.buf(out_name, N, Ho, Wo, conv.cout)
            sc, sh = self._winograd_conv(name, x, conv, y, bn, bn_name, x_act=x_act)
            return y, sc, sh
        assert x_act is None, name
        wpk = self.pack(name, conv.weight, 0, version=version, cs=C)
        y = self.buf(out_name, N, Ho, Wo, conv.cout)
        if self.prof:
            self._prof_igemm(N * Ho * Wo, conv.cout, 2.0 * N * Ho * Wo * conv.cout * conv.cin * k * k, 'fwd ' + name, 0,
                             (N, H, W, C, Ho, Wo, k, s, p))
        sc, sh = self._conv_bn_call('conv+bn ' + name, bn_name, bn, y,
                                    (x.data_ptr(), wpk.data_ptr(), _p(conv.bias), y.data_ptr(), N, H, W, C, Ho, Wo, conv.cout, k, k,
                                     s, p, 0), 0, s)
        if self.prof:
            self.prof.end()
        return y, sc, sh

    # ConvTranspose2d(64 -> 64, 2x2, stride 2) forward in 16-bit storage through its own kernel (round 5, csrc/convt16.hip: the layer is
    # HBM-bound and the generic parity-class launch ran it at 2.5x its memory time).  DBN_CONVT16=0: the generic launch.
    stem16_pool = os.environ.get('DBN_STEM16_POOL', '1') == '1'  # inference: the 16-bit stem kernel applies BatchNorm + ReLU and pools (one launch)
    pw16 = os.environ.get('DBN_PW16', '1') == '1'  # inference: pointwise convs 64 -> 64 | 256 on 16-bit storage through csrc/convt16.hip's kernel
    convt16 = os.environ.get('DBN_CONVT16', '1') == '1'

    def _convt16_panel(self, name, ct):
        w = ct.weight
        key = (name, 'convt16', self.kind)
        stamp = (w._version, self.param_epoch, w.data_ptr())
        ent = self.packs.get(key)
        if ent is None or ent[1] != stamp:
            panel = ent[0] if ent is not None else torch.empty(self.L.dbn_convt16_panel_bytes(), device=w.device, dtype=torch.uint8)
            check(self.L.dbn_convt16_pack(self.kind, w.data_ptr(), panel.data_ptr(), self.stream), 'convt16_pack')
            self.packs[key] = (panel, stamp)
        return self.packs[key][0]

    head16 = os.environ.get('DBN_HEAD16', '1') == '1'  # inference, 16-bit storage: ConvT -> BN -> ReLU -> ConvT -> sigmoid of both branches in ONE launch

    def _convt16(self, name, x, ct, out_name, bn_name=None, bn=None):
        """None when the layer does not take the 16-bit ConvT kernel; else (y, scale, shift) — scale / shift None without a BatchNorm."""
        N, H, W, C = x.shape
        L = self.L
        if not (self.convt16 and self.at != 0 and ct.k == 2 and ct.stride == 2 and bool(L.dbn_convt16_eligible(self.at, N, H, W, C, ct.cout))
                and ct.cin == C):
            return None
        panel = self._convt16_panel(name, ct)
        y = self.buf(out_name, N, 2 * H, 2 * W, ct.cout)
        if self.prof:
            self.prof.begin('convt2x2_b16_kernel<%d>' % self.at, 2.0 * N * H * W * C * ct.cout * 4, float(2 * (x.numel() + y.numel())), 'convT fwd ' + name)
        sc = sh = None
        if bn is not None:
            sc, sh = self.fbuf(bn_name + '/scale', ct.cout), self.fbuf(bn_name + '/shift', ct.cout)
            mu, rs = self.fbuf(bn_name + '/mean', ct.cout), self.fbuf(bn_name + '/rstd', ct.cout)
            ws = self.scratch('_conv_bn_ws', (3 * 64 + 1) * L.dbn_convt16_rows())
            check(L.dbn_convt16_bn_t(self.at, x.data_ptr(), panel.data_ptr(), _p(ct.bias), y.data_ptr(), N, H, W, bn.weight.data_ptr(),
                                     bn.bias.data_ptr(), bn.eps, bn.momentum, bn.running_mean.data_ptr(), bn.running_var.data_ptr(),
                                     sc.data_ptr(), sh.data_ptr(), mu.data_ptr(), rs.data_ptr(), ws.data_ptr(), self.stream), 'convt16+bn ' + name)
            self.nbt_pending[bn_name] = self.nbt_pending.get(bn_name, 0) + 1
        else:
            check(L.dbn_convt16_bn_t(self.at, x.data_ptr(), panel.data_ptr(), _p(ct.bias), y.data_ptr(), N, H, W, None, None, 0.0, 0.0, None,
                                     None, None, None, None, None, None, self.stream), 'convt16 ' + name)
        if self.prof:
            self.prof.end()
        return y, sc, sh

    def convT_bn(self, name, x, ct, out_name, bn_name, bn, train):
        if train and self.fuse_bn_stats:
            got = self._convt16(name, x, ct, out_name, bn_name, bn)
            if got is not None:
                return got
        if not (train and self.fuse_bn_stats):
            y = self.convT_fwd(name, x, ct, out_name)
            sc, sh = self.bn_coef(bn_name, bn, y, train)
            return y, sc, sh
        N, H, W, C = x.shape
        wpk = self.pack(name, ct.weight, 1, 2)
        y = self.buf(out_name, N, 2 * H, 2 * W, ct.cout)
        if self.prof:
            self._prof_convT(N, H, W, C, ct.cout, name)
        sc, sh = self._conv_bn_call('convT+bn ' + name, bn_name, bn, y,
                                    (x.data_ptr(), wpk.data_ptr(), _p(ct.bias), y.data_ptr(), N, H, W, C, 2 * H, 2 * W, ct.cout, 2,
                                     2, 2, 0, 1), 1, 2)
        if self.prof:
            self.prof.end()
        return y, sc, sh

    def _prof_convT(self, N, H, W, C, Co, name):
        """label of a ConvTranspose2d(2x2, stride 2) forward: convt2x2_f32_kernel<Cin> (convt_f32.hip) in exact fp32, else the
        parity-class launch of the implicit-GEMM kernel"""
        flops = 2.0 * N * H * W * C * Co * 4
        at = 3 if self._use_planes else self.at
        if self.L.dbn_igemm_kernel_config(at, self.ns, 2, N, H, W, C, 2 * H, 2 * W, Co, 2, 2, 2, 0, 0, 1) & 32:
            self.prof.begin('convt2x2_f32_kernel<%d>' % C, flops, 0.0, 'convT fwd ' + name)
        else:
            self._prof_igemm(N * 4 * H * W, Co, flops, 'convT fwd ' + name, 2)

    def _prof_igemm(self, M, Cd, flops, tag='', mode=0, geom=None, epi=0):
        """geom = (N, Hs, Ws, Cs, Hd, Wd, R, stride, pad) of a forward / stride-1 data-gradient call: only those can take the
        pixel-patch kernel (the last template argument of the symbol)."""
        cfg = self.L.dbn_igemm_tile_config_ns(M, Cd, self.ns)
        at = 3 if self._use_planes else self.at
        if geom is not None and mode < 2:
            N, Hs, Ws, Cs, Hd, Wd, R, stride, pad = geom
            ks = self.L.dbn_igemm_splitk_plan_ns(N * Hd * Wd, Cd, R * R * Cs, Cs, self.ns) if self.splitk else 1
            cfg = self.L.dbn_igemm_kernel_config(at, self.ns, mode, N, Hs, Ws, Cs, Hd, Wd, Cd, R, R, stride, pad, 0, ks)
        blk = 'false' if (geom is not None and mode == 0 and geom[3] % 16 != 0) else 'true'
        nbytes = 0.0
        if geom is not None:  # algorithmic traffic: source and destination once, in the storage type (weights are L2-resident)
            N, Hs, Ws, Cs, Hd, Wd = geom[:6]
            nbytes = float(N) * (Hs * Ws * Cs + Hd * Wd * Cd) * (4 if self.at == 0 else 2)
        if geom is not None and mode < 2 and geom[6] == 3 and geom[7] == 1 and geom[8] == 1 and self.ns == 1 and \
                self.L.dbn_wres16_would_run(at, mode, geom[0], geom[4], geom[5], geom[3], Cd, epi, 0):
            # round 6: the weight-resident kernel (csrc/wres16.hip) — <at, Cs, Cd, mode>; the epilogue flavour is not part of the label
            self.prof.begin('conv3x3_wres16_kernel<%d,%d,%d,%d>' % (at, geom[3], Cd, mode), flops, nbytes, tag)
            return
        label = IGEMM_TILE_NAMES[cfg & 15] % (mode, self.ns, at, 'true' if cfg & 16 else 'false', epi, blk)
        if (cfg & 15) == 1 and at in (1, 2) and self.ns == 1:  # round 6: the 128 x 128 tile of the 16-bit storage types runs as 1 x 4 waves (csrc/igemm_kernel.h DBN_CFG1_WM)
            label = label.replace('<128,128,2,2,', '<128,128,1,4,')
        self.prof.begin(label, flops, nbytes, tag)

    def _winograd_dgrad(self, name, dy, conv, dx, accumulate, consumer, panel=None):
        """The data gradient of a 3x3 / stride-1 / pad-1 conv through the Winograd kernel (the rotated / transposed filters), with the
        BatchNorm-backward sums of its consumer in the epilogue exactly as _igemm arranges them for the implicit-GEMM kernels.
        panel: a ready Winograd panel mapping dy's channels to dx's (the FPN output conv's level-0 gradient: a forward-form conv
        with the combined weights)."""
        import ctypes
        N, H, W, O = dy.shape
        Cd = dx.shape[3]
        up = panel if panel is not None else self._winograd_panel(name, conv.weight, O, dgrad=1)
        L = self.L
        rows = L.dbn_winograd_rows(N, H, W)
        a = dict(y=None, zmask=None, msc=None, msh=None, mean=None, rstd=None, part=None, y2=None, mean2=None, rstd2=None, part2=None)
        fin = None
        if consumer is not None:
            bn_name, y, zmask = consumer[:3]
            second = consumer[3] if len(consumer) > 3 else None
            assert tuple(y.shape) == (N, H, W, Cd) and (zmask is None or zmask.shape == y.shape), (name, bn_name)
            a.update(y=y, zmask=zmask, mean=self.bufs[bn_name + '/mean'], rstd=self.bufs[bn_name + '/rstd'],
                     part=self.fbuf(bn_name + '/bnb_part', 2 * Cd * rows))
            if zmask is None:
                a.update(msc=self.bufs[bn_name + '/scale'], msh=self.bufs[bn_name + '/shift'])
            if second is not None:
                assert zmask is not None and second[1].shape == y.shape
                a.update(y2=second[1], mean2=self.bufs[second[0] + '/mean'], rstd2=self.bufs[second[0] + '/rstd'],
                         part2=self.fbuf(second[0] + '/bnb_part', 2 * Cd * rows))
            if self.bnb_finalize_in_kernel:
                G = self.grad_views
                cnt = self.bufs.get(bn_name + '/bnb_cnt')
                ncnt = L.dbn_igemm_bn_final_counters(rows, Cd)
                if cnt is None or cnt.numel() != ncnt or cnt.device != self.flat.device:
                    cnt = torch.zeros(ncnt, device=self.flat.device, dtype=torch.int32)  # (the kernels leave them zero)
                    self.bufs[bn_name + '/bnb_cnt'] = cnt
                grp = self.fbuf(bn_name + '/bnb_grp', L.dbn_igemm_bn_final_group_floats(rows, Cd))
                c1c2 = self.fbuf(bn_name + '/bnb_c1c2', 2 * Cd)
                fin = _lib.BnbFinal(cnt.data_ptr(), grp.data_ptr(), c1c2.data_ptr(), G[bn_name + '.weight'].data_ptr(),
                                    G[bn_name + '.bias'].data_ptr(), None, None, None, self.grad_scale)
                if second is not None:
                    c1c2b = self.fbuf(second[0] + '/bnb_c1c2', 2 * Cd)
                    fin.c1c2_2, fin.dgamma_2, fin.dbeta_2 = (c1c2b.data_ptr(), G[second[0] + '.weight'].data_ptr(),
                                                             G[second[0] + '.bias'].data_ptr())
        if self.prof:  # (FLOPs the MFMA pipe executes: see _winograd_conv)
            # algorithmic bytes: dy + dx once, + the operands of the fused BatchNorm-backward sums (y, a separate ReLU mask, a second BatchNorm's
            # y) and the accumulated-onto dx, once each
            extra = sum(a[k] is not None for k in ('y', 'zmask', 'y2')) + int(bool(accumulate))
            self.prof.begin('winograd_f32_kernel', 2.0 * N * H * W * O * Cd * 4, 4.0 * N * H * W * (O + Cd * (1 + extra)), 'dgrad ' + name)
        check(L.dbn_winograd_dgrad_bnsums_f32(dy.data_ptr(), up.data_ptr(), dx.data_ptr(), N, H, W, O, Cd, int(accumulate), _p(a['y']),
                                              _p(a['zmask']), _p(a['msc']), _p(a['msh']), _p(a['mean']), _p(a['rstd']), _p(a['part']),
                                              _p(a['y2']), _p(a['mean2']), _p(a['rstd2']), _p(a['part2']),
                                              ctypes.byref(fin) if fin is not None else None, self.stream), 'winograd dgrad ' + name)
        if self.prof:
            self.prof.end()
        if consumer is not None:
            self._bnb_sums[consumer[0]] = (c1c2, -1) if fin is not None else (a['part'], rows)
            if len(consumer) > 3 and consumer[3] is not None:
                self._bnb_sums[consumer[3][0]] = (c1c2b, -1) if fin is not None else (a['part2'], rows)

    def conv_dgrad(self, name, dy, conv, dx, accumulate, version=None, consumer=None):
        """consumer: see _igemm — the BatchNorm that will consume dx, when this call is dx's last writer."""
        N, Ho, Wo, O = dy.shape
        _, H, W, I = dx.shape
        if (version is None and self.winograd and self.ns == 0 and self.at == 0 and conv.k == 3 and conv.stride == 1 and conv.padding == 1
                and (Ho, Wo) == (H, W) and dy.dtype == torch.float32 and I == conv.cin and O == conv.cout
                and bool(self.L.dbn_winograd_eligible(N, H, W, O, I))):
            if consumer is not None and not (self.fuse_bn_bwd_sums and not (self.prof is not None and self.prof.labels is None and not self.prof_fused)):
                consumer = None
            self._winograd_dgrad(name, dy, conv, dx, accumulate, consumer)
            return
        wpk = self.pack(name, conv.weight, 1, conv.stride, version=version)
        args = (dy.data_ptr(), wpk.data_ptr(), None, dx.data_ptr(), N, Ho, Wo, O, H, W, I, conv.k, conv.k, conv.stride, conv.padding, 1,
                int(accumulate), 0)
        if consumer is not None and not self._bnb_eligible(args):
            consumer = None
        if self.prof:  # algorithmic FLOPs of a data gradient = those of the forward conv
            self._prof_igemm(N * H * W, I, 2.0 * N * Ho * Wo * O * I * conv.k * conv.k, 'dgrad ' + name, 2 if conv.stride == 2 else 1,
                             (N, Ho, Wo, O, H, W, conv.k, conv.stride, conv.padding), epi=int(consumer is not None))
        self._igemm('igemm dgrad ' + name, *args, consumer=consumer)
        if self.prof:
            self.prof.end()

    # Slab reductions of the weight gradients as ONE grouped launch per gradient stage (dbn_wgrad_reduce_many) instead of one small
    # launch behind every matrix kernel (round-3 review: 34 launches, 0.46 ms of work, 4.6 ms in flight on the side stream).
    # Built, bit-identical (tested), measured on one box in interleaved runs and OFF: every layer then needs its OWN slab until
    # the grouped launch runs (1.7 GB at bs16 640^2), while the per-layer form re-uses ONE <= 50 MB scratch that never leaves
    # the 256 MB Infinity Cache — f32 545.6 -> 542.1 images/s, bf16 1667 -> 1340-1420 (profiles/r04_ab_grouped_reduce.txt).  The long
    # in-flight time of the small reductions costs nothing: they run beside MFMA kernels on a stream that is not the critical path.
    defer_wgrad_reduce = os.environ.get('DBN_DEFER_REDUCE', '0') == '1'
    winograd_wgrad = os.environ.get('DBN_WINOGRAD_WGRAD', '1') == '1'  # (A/B switch; off: the direct kernels of wgrad_kernels.h)

    def wgrad(self, name, sm, big, O, I, k, stride, pad, gview, defer=False, big_act=None):
        """defer: the gradient is only needed by the optimizer / the gradient exchange, so its slab reduction may wait for
        flush_wgrad_reduces() (conv_wgrad, convT_bwd); False: a kernel of this pass reads gview next (FPN level scatter, DCN)."""
        N, Ho, Wo, _ = sm.shape
        _, H, W, Cb = big.shape
        if ((k, stride, pad) == (3, 1, 1) and (Ho, Wo) == (H, W)
                and self.plan(_VirtualConv(I, O, k, stride, pad, None, None), big.shape, True,
                              fp32_tensors=sm.dtype == torch.float32 and big.dtype == torch.float32).winograd_wgrad):
            # 3x3 / stride 1 in exact fp32: Winograd F(2x2,3x3) over the tiles (csrc/winograd_wgrad_f32.hip), 2.25x fewer matrix FLOPs
            slab = self.scratch('_wgrad_slab', self.L.dbn_winograd_wgrad_slab_floats(N, H, W, O, Cb))
            asc, ash = (big_act[0].data_ptr(), big_act[1].data_ptr()) if big_act is not None else (None, None)
            wargs = (sm.data_ptr(), big.data_ptr(), asc, ash, slab.data_ptr(), gview.data_ptr(), N, H, W, O, Cb, I, self.grad_scale, self.stream)
            if self.prof:  # (FLOPs the matrix pipe executes: 16 products per tile and channel pair; see _winograd_conv)
                self.prof.begin('winograd_wgrad_f32_kernel', 2.0 * N * ((H + 1) // 2) * ((W + 1) // 2) * 16 * O * Cb, 0.0, 'wgrad ' + name)
                check(self.L.dbn_winograd_wgrad_f32(1, *wargs), 'winograd wgrad ' + name)
                self.prof.end()
                self.prof.begin('wgrad_reduce_kernel', 0.0, 0.0, 'wgrad reduce ' + name)
                check(self.L.dbn_winograd_wgrad_f32(2, *wargs), 'winograd wgrad reduce ' + name)
                self.prof.end()
            else:
                check(self.L.dbn_winograd_wgrad_f32(3, *wargs), 'winograd wgrad ' + name)
            return
        assert big_act is None, name  # (lazy_act() defers an activation only where this branch is taken)
        # reductions on their own stream (see reduce_stream): only for gradients no kernel of this pass reads, on the side stream
        async_reduce = (defer and self.reduce_stream and self._in_side and self.prof is None and not self.defer_wgrad_reduce
                        and not torch.cuda.is_current_stream_capturing())
        defer = defer and self.defer_wgrad_reduce and self.prof is None and Cb % 64 == 0
        slab = self.scratch('_wgrad_slab/' + name if defer else ('_wgrad_slab@%d' % self._slab_k if async_reduce else '_wgrad_slab'),
                            self.L.dbn_wgrad_slab_floats_hw(N, Ho, Wo, O, H, W, Cb, k, k, 4 if self.at == 0 else 2))
        at, smp, bigp = self.at, sm.data_ptr(), big.data_ptr()
        if self._use_planes and sm.dtype == torch.float32 and big.dtype == torch.float32:
            at, smp, bigp = 3, self._planes(sm).data_ptr(), self._planes(big).data_ptr()
        args = (at, self.ns, smp, bigp, slab.data_ptr(), gview.data_ptr(), N, Ho, Wo, O, H, W, Cb, I, k, k, stride,
                pad, self.grad_scale, self.stream)
        if self.prof:  # bracket the matrix kernel alone (its rocprofv3 symbol), then the slab reduction
            cfg = self.L.dbn_wgrad_kernel_config_hw(at, self.ns, O, Cb, k, k, stride, pad, Ho, Wo, H, W)
            wname = WGRAD_TILE_NAMES[cfg & 15] % (self.ns, at, (cfg >> 6) & 1)  # <BM,BN,WM,WN,NS,AT,ROW>
            if cfg & 32:  # 3x3 / stride 1 in the 16-bit matrix modes: pixel-patch kernel
                wname = 'wgrad_patch_kernel<%d,%d>' % (self.ns, at)
            elif cfg & 16:  # bf16 tensors: LDS-DMA + transposing LDS reads
                wname = 'wgrad_tr_kernel<' + wname.split('<')[1].rsplit(',', 3)[0] + '>'
            self.prof.begin(wname,
                            2.0 * N * Ho * Wo * O * I * k * k, 0.0, 'wgrad ' + name)
            check(self.L.dbn_wgrad_phase_t(1, *args), 'wgrad ' + name)
            self.prof.end()
            self.prof.begin('wgrad_reduce_kernel', 0.0, 0.0, 'wgrad reduce ' + name)
            check(self.L.dbn_wgrad_phase_t(2, *args), 'wgrad reduce ' + name)
            self.prof.end()
        elif defer:
            check(self.L.dbn_wgrad_phase_t(1, *args), 'wgrad ' + name)
            key = (name, args[:-1])
            job = self._reduce_job_cache.get(key)
            if job is None:
                import ctypes
                job = _lib.WgradReduceJob()
                check(self.L.dbn_wgrad_reduce_describe(*args[:-1], ctypes.byref(job)), 'wgrad reduce describe ' + name)
                self._reduce_job_cache[key] = job
            self._reduce_pending.append((key, job))
        elif async_reduce:
            side = torch.cuda.current_stream(self.flat.device)
            free = self._slab_free[self._slab_k]
            if free is not None:  # this slab's previous reduction (two weight gradients ago) has read it
                side.wait_event(free)
            check(self.L.dbn_wgrad_phase_t(1, *args), 'wgrad ' + name)
            filled = torch.cuda.Event()
            filled.record(side)
            if self._side2 is None or self._side2.device != self.flat.device:
                self._side2 = torch.cuda.Stream(device=self.flat.device)
            self._side2.wait_event(filled)
            with torch.cuda.stream(self._side2):
                check(self.L.dbn_wgrad_phase_t(2, *args[:-1], self._side2.cuda_stream), 'wgrad reduce ' + name)
                done = torch.cuda.Event()
                done.record(self._side2)
            self._slab_free[self._slab_k] = done
            self._slab_k ^= 1
            self._side2_used = True
        else:
            check(self.L.dbn_wgrad_t(*args), 'wgrad ' + name)

    # The slab reduction of a weight gradient (34 launches of 5-260 us per step, latency-bound: 0.46 ms alone) sits BETWEEN the matrix
    # kernels of the side stream, and since the Winograd convs shortened the main stream the side stream is what the step's end waits
    # for (round-4 trace: wgrad_reduce64 1.4 ms in flight, 0.9 ms of it with nothing else running).  reduce_stream = True runs the
    # reductions of gradients that only the optimizer reads on a THIRD stream behind their matrix kernel, over two alternating slab
    # scratches (both stay in the Infinity Cache).  Built, bit-identical, measured on one box and OFF: f32 614.7 -> 606.7 images/s,
    # bf16 1600 -> 1390-1470 (three extra event / wait calls per weight gradient on the host, and the reductions now compete with the
    # NEXT matrix kernel instead of running in its shadow at the end); not used under hipGraph capture (cross-stream events created
    # inside a capture crashed capture_end on ROCm 7.2).
    reduce_stream = os.environ.get('DBN_REDUCE_STREAM', '0') == '1'
    _side2 = None
    _side2_used = False
    _slab_k = 0

    def flush_wgrad_reduces(self):
        """The deferred slab reductions as ONE launch on the side stream (behind the matrix kernels that filled the slabs)."""
        self._flush_wgrads()  # (every weight gradient of the stage is issued before its gradients are announced / joined)
        if not self._reduce_pending:
            return
        pending, self._reduce_pending = self._reduce_pending, []
        sig = tuple(k for k, _ in pending)
        ent = self._reduce_tables.get(sig)
        if ent is None:
            import ctypes
            arr = (_lib.WgradReduceJob * len(pending))(*[j for _, j in pending])
            first = [0]
            for _, j in pending:
                first.append(first[-1] + j.blocks)
            dev = self.flat.device
            ent = (torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev), torch.tensor(first, dtype=torch.int32, device=dev),
                   len(pending), first[-1], max(j.smem_bytes for _, j in pending))
            self._reduce_tables[sig] = ent
        table, first, n, blocks, smem = ent
        with self.side_stream():
            check(self.L.dbn_wgrad_reduce_many(table.data_ptr(), first.data_ptr(), n, blocks, smem, self.stream), 'wgrad_reduce_many')

    # Weight gradients enqueued LATE: conv_wgrad() only queues the launch; it is issued on the side stream right before the next
    # BatchNorm backward of the main stream (or at the end of the pass).  The side stream waits for everything the main stream has
    # enqueued by then — i.e. also for this layer's data gradient — so the weight gradient runs beside the HBM-bound BatchNorm pass
    # and the next layer's data gradient instead of beside its own layer's data gradient (two MFMA kernels side by side are zero-sum,
    # an MFMA kernel beside an HBM-bound one hides it).  Round 3 measured this as a loss (538 -> 532 images/s) when the data
    # gradients were the longer kernels; with the Winograd data gradients the main stream's BatchNorm passes had become exposed
    # (1.28 ms of bn_bwd_apply with no MFMA kernel in flight, round-4 trace).  Measured on one box, interleaved: exact fp32 629 vs
    # 633 images/s (off is better: default off there); bf16 1620-1630 vs 1430-1500 (on is better, and equal to the round-3 tree on the
    # same box — the bf16 step with immediate launches is bimodal from process to process, 1170 / 1445 / 1670 images/s for the SAME
    # library, profiles/r04_late_wgrad.txt — the late order has only shown the fast mode): default on for the 16-bit storage modes.
    late_wgrad = os.environ.get('DBN_LATE_WGRAD', '')  # '' = in the 16-bit storage modes only (measured, see above); '0' / '1' force

    def _flush_wgrads(self):
        if self._wgrad_fifo:
            fifo, self._wgrad_fifo = self._wgrad_fifo, []
            for fn in fifo:
                fn()

    def conv_wgrad(self, name, dy, x, conv, x_act=None):
        self._presplit(dy, x)

        def launch():
            with self.side_stream():
                self.wgrad(name, dy, x, conv.cout, conv.cin, conv.k, conv.stride, conv.padding, self.grad_views[name + '.weight'], defer=True,
                           big_act=x_act)
                if conv.bias is not None and name + '.bias' not in self._bias_done:
                    self.col_sum(dy, self.grad_views[name + '.bias'])
        late = self.late_wgrad == '1' or (self.late_wgrad == '' and self.at != 0)
        if late and self.overlap_wgrad and not self._in_side and self.prof is None:
            self._wgrad_fifo.append(launch)
        else:
            launch()

    def convT_fwd(self, name, x, ct, out_name):
        got = self._convt16(name, x, ct, out_name)
        if got is not None:
            return got[0]
        N, H, W, C = x.shape
        wpk = self.pack(name, ct.weight, 1, 2)
        y = self.buf(out_name, N, 2 * H, 2 * W, ct.cout)
        if self.prof:
            self._prof_convT(N, H, W, C, ct.cout, name)
        self._igemm('igemm convT fwd ' + name, x.data_ptr(), wpk.data_ptr(), _p(ct.bias), y.data_ptr(), N, H, W, C, 2 * H, 2 * W,
                    ct.cout, 2, 2, 2, 0, 1, 0, 0)
        if self.prof:
            self.prof.end()
        return y

    def convT_bwd(self, name, dy, x, ct, dx, consumer=None):
        N, H2, W2, Co = dy.shape
        _, H, W, Ci = x.shape
        wpk = self.pack(name, ct.weight, 0)
        args = (dy.data_ptr(), wpk.data_ptr(), None, dx.data_ptr(), N, H2, W2, Co, H, W, Ci, 2, 2, 2, 0, 0, 0, 0)
        if consumer is not None and not self._bnb_eligible(args):
            consumer = None
        if self.prof:
            self._prof_igemm(N * H * W, Ci, 2.0 * N * H * W * Ci * Co * 4, 'convT dgrad ' + name, 0, epi=int(consumer is not None))
        self._igemm('igemm convT dgrad ' + name, *args, consumer=consumer)
        if self.prof:
            self.prof.end()
        self._presplit(x, dy)
        with self.side_stream():
            self.wgrad(name, x, dy, Ci, Co, 2, 2, 0, self.grad_views[name + '.weight'], defer=True)
            if ct.bias is not None and name + '.bias' not in self._bias_done:
                self.col_sum(dy, self.grad_views[name + '.bias'])

    def col_sum(self, x, out):
        C = x.shape[-1]
        M = x.numel() // C
        check(self.L.dbn_col_sum_t(self.at, x.data_ptr(), M, C, out.data_ptr(), self.grad_scale, self.reduce_ws().data_ptr(),
                                   self.stream), 'col_sum')

    def bn_coef(self, name, bn, y, train):
        C = y.shape[-1]
        M = y.numel() // C
        sc, sh = self.fbuf(name + '/scale', C), self.fbuf(name + '/shift', C)
        if train:
            mu, rs = self.fbuf(name + '/mean', C), self.fbuf(name + '/rstd', C)
            check(self.L.dbn_bn_train_stats_t(self.at, y.data_ptr(), M, C, bn.weight.data_ptr(), bn.bias.data_ptr(), bn.eps, bn.momentum,
                                            bn.running_mean.data_ptr(), bn.running_var.data_ptr(), sc.data_ptr(), sh.data_ptr(),
                                            mu.data_ptr(), rs.data_ptr(), self.reduce_ws().data_ptr(), self.stream),
                  'bn stats ' + name)
            self.nbt_pending[name] = self.nbt_pending.get(name, 0) + 1  # folded into the buffer by flush_counters()
        else:
            check(self.L.dbn_bn_eval_coef(C, bn.weight.data_ptr(), bn.bias.data_ptr(), bn.running_mean.data_ptr(),
                                          bn.running_var.data_ptr(), bn.eps, sc.data_ptr(), sh.data_ptr(), self.stream),
                  'bn eval ' + name)
        return sc, sh

    def _prof_hbm(self, kernel, nbytes, tag=''):
        """Bracket an HBM-bound launch with its ALGORITHMIC bytes (tensors read + written once); pair with self.prof.end()."""
        if self.prof:
            self.prof.begin(kernel, 0.0, float(nbytes), tag)

    def bn_apply(self, y, sc, sh, out_name, relu=True, res=None, rsc=None, rsh=None):
        C = y.shape[-1]
        out = self.buf(out_name, *y.shape)
        self._prof_hbm('bn_apply_kernel', y.numel() * y.element_size() * (2 + (res is not None)), out_name)
        check(self.L.dbn_bn_apply_t(self.at, y.data_ptr(), sc.data_ptr(), sh.data_ptr(), _p(res), _p(rsc), _p(rsh), out.data_ptr(),
                                    y.numel() // C, C, int(relu), self.stream), 'bn apply ' + out_name)
        if self.prof:
            self.prof.end()
        return out

    # ---- inference: eval-mode BatchNorm folded into the weights (round 5; basic.py:32-36, resnet.py:70-91 under model.eval(), test.py:53-59).
    # w' = w * gamma / sqrt(var + eps) and bias' = beta + (bias - mean) * that go through the ordinary pack functions as derived tensors;
    # the conv's epilogue adds the bias (and a residual input), applies the ReLU and writes the ACTIVATION: no scale / shift launch, no
    # bn_apply pass (every activation was written twice before: the raw conv output, then the normalised one).  DBN_FOLD_EVAL_BN=0
    # restores the conv -> coefficients -> bn_apply chain (the A/B and bit-identity baseline of the train path's kernels).
    fold_eval_bn = os.environ.get('DBN_FOLD_EVAL_BN', '1') == '1'
    train_forwards = 0  # counts train-mode forwards: the running statistics change under them through raw pointers

    def _fold_ok(self, x, conv, train):
        return self.plan(conv, x.shape, train).fold

    def _folded(self, name, conv, bn):
        """(virtual conv with the folded weight / bias, version stamp), re-made when a parameter or a running statistic changed."""
        w = conv.weight
        stamp = (w._version, self.param_epoch, w.data_ptr(), self.train_forwards, bn.running_mean._version, bn.running_var._version,
                 bn.weight._version, bn.bias._version)
        ent = self.packs.get((name, 'fold'))
        if ent is None or ent[1] != stamp:
            wf, bf = ent[0][:2] if ent is not None else (device_empty(tuple(w.shape), w.device), device_empty((w.shape[0], ), w.device))
            O = w.shape[0]
            check(self.L.dbn_fold_bn_eval(w.data_ptr(), O, w.numel() // O, _p(conv.bias), bn.weight.data_ptr(), bn.bias.data_ptr(),
                                          bn.running_mean.data_ptr(), bn.running_var.data_ptr(), bn.eps, wf.data_ptr(), bf.data_ptr(),
                                          self.stream), 'fold_bn_eval ' + name)
            self._fold_count += 1  # (the derived tensors are rewritten through raw pointers: this serial is their version)
            ent = ((wf, bf, self._fold_count), stamp)
            self.packs[(name, 'fold')] = ent
        wf, bf, serial = ent[0]
        return _VirtualConv(conv.cin, conv.cout, conv.k, conv.stride, conv.padding, wf, bf), ('fold', serial)

    _fold_count = 0

    def conv_bn_act_eval(self, name, x, conv, out_name, bn, relu=True, res=None):
        """relu(bn(conv(x)) [+ res]) in ONE launch, eval mode (the caller checked _fold_ok)."""
        vconv, ver = self._folded(name, conv, bn)
        N, H, W, C = x.shape
        k, s, p = conv.k, conv.stride, conv.padding
        Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
        assert C >= conv.cin and C % 4 == 0, (name, C, conv.cin)
        z = self.buf(out_name, N, Ho, Wo, conv.cout)
        assert res is None or (tuple(res.shape) == tuple(z.shape) and res.dtype == z.dtype), name
        if res is None and self.plan(conv, x.shape).pw16:
            # 16-bit storage: a pointwise conv 64 -> 64 | 256 (the FPN lateral on c2) on the ConvT kernel's construction — whole panel in
            # registers, A fragments straight from global memory, no ring (round 5)
            key = (name, 'pw16', self.kind)
            ent = self.packs.get(key)
            stamp = (ver, self.param_epoch, vconv.weight.data_ptr())
            if ent is None or ent[1] != stamp:
                panel = ent[0] if ent is not None else torch.empty(self.L.dbn_pw16_panel_bytes(), device=x.device, dtype=torch.uint8)
                check(self.L.dbn_pw16_pack(self.kind, vconv.weight.data_ptr(), conv.cout, panel.data_ptr(), self.stream), 'pw16_pack')
                self.packs[key] = (panel, stamp)
            panel = self.packs[key][0]
            if self.prof:
                self.prof.begin('convt2x2_b16_kernel<%d>' % self.at, 2.0 * N * H * W * conv.cout * C, float(2 * (x.numel() + z.numel())), 'fwd ' + name)
            check(self.L.dbn_pw16_act_t(self.at, x.data_ptr(), panel.data_ptr(), vconv.bias.data_ptr(), int(relu), z.data_ptr(), N, H, W,
                                        conv.cout, self.stream), 'pw16 ' + name)
        elif self._winograd_ok(x, vconv):
            up = self._winograd_panel(name + '#fold', vconv.weight, C, version=ver)
            if self.prof:
                self.prof.begin('winograd_f32_kernel', 2.0 * N * H * W * conv.cout * conv.cin * 4, 0.0, 'fwd ' + name)
            check(self.L.dbn_winograd_conv_act_f32(x.data_ptr(), up.data_ptr(), vconv.bias.data_ptr(), _p(res), int(relu), z.data_ptr(), N, H, W, C,
                                                   conv.cout, self.stream), 'winograd act ' + name)
        else:
            wpk = self.pack(name + '#fold', vconv.weight, 0, version=ver, cs=C)
            if self.prof:
                self._prof_igemm(N * Ho * Wo, conv.cout, 2.0 * N * Ho * Wo * conv.cout * conv.cin * k * k, 'fwd ' + name, 0,
                                 (N, H, W, C, Ho, Wo, k, s, p))
            at, srcp = self._src(x.data_ptr(), C)
            check(self.L.dbn_igemm_act_t(at, self.ns, srcp, wpk.data_ptr(), vconv.bias.data_ptr(), _p(res), int(relu), z.data_ptr(), N, H, W, C,
                                         Ho, Wo, conv.cout, k, k, s, p, 0, 0, self.stream), 'igemm act ' + name)
        if self.prof:
            self.prof.end()
        return z

    # Apply-on-load BatchNorm + ReLU (basic.py:32-36, resnet.py:77-80): where every consumer of relu(bn(y)) is a Winograd conv (forward and
    # weight gradient) the activation tensor is never written — the kernels apply relu(fma(y, scale, shift)) while they stage their
    # patches, with bn_apply's own arithmetic (bit-identical results).  Exact-fp32 mode only (the 16-bit kernels bring their tiles to
    # LDS by DMA: nothing passes through registers).  The BatchNorm backward takes its ReLU mask from y already (mask 'self').
    apply_on_load = os.environ.get('DBN_APPLY_ON_LOAD', '1') == '1'
    # The FPN output conv's level 0 (a plain 3x3 conv of p2, 53 % of the pyramid's FLOPs) through the Winograd kernel, levels 1-3 added by
    # the pyramid launch (dbn_pyramid_conv_from_t): same results to fp32 rounding (test_pyramid_conv_on_a_winograd_level_0), measured
    # on one box in interleaved runs: 702.6 / 703.8 / 703.2 images/s without, 709.6 / 708.0 / 708.6 with (+0.8 %: the 64 -> 256 Winograd
    # launch re-stages every patch four times and the second launch re-reads the 164 MB output it accumulates into, which eats most of the
    # 2.25x).  (A first measurement had shown "no difference": a wrong shape check had kept the switch from taking effect — the test
    # found it.)
    fpn_level0_winograd = os.environ.get('DBN_FPN_LV0_WINOGRAD', '1') == '1'

    def lazy_act(self, y, convs, train):
        """True when relu(bn(y)) may stay unwritten: every conv in `convs` (all read it as their input) runs as a Winograd conv forward
        and — in training — takes its weight gradient through the Winograd kernel too."""
        # (inference with folded BatchNorm: the producer's epilogue writes the activation itself, conv_bn_act_eval — the plan says no)
        return all(self.plan(conv, y.shape, train, fp32_tensors=y.dtype == torch.float32).apply_on_load for conv in convs)

    bias_grad_in_bn = True  # bias gradients of convs that feed a BatchNorm are formed inside its backward apply pass

    def bn_backward(self, name, y, mask, dout, dy_name, gout=None, gout_acc=False, sums=None, conv_bias=None, sums_parts=1):
        """mask: None (no ReLU), 'self' (ReLU directly on this BN's output: recomputed from y with the
        forward's scale/shift, nothing extra is read), or a tensor (saved activation whose sign gates).
        conv_bias: name of the bias parameter of the conv that produced y (its gradient = column sums of dy)."""
        self._flush_wgrads()  # (late_wgrad: the queued weight gradients start beside this HBM-bound pass)
        C = y.shape[-1]
        M = y.numel() // C
        dy = self.buf(dy_name, *y.shape)
        zmask = msc = msh = None
        if isinstance(mask, str):
            msc, msh = self.bufs[name + '/scale'], self.bufs[name + '/shift']
        elif mask is not None:
            zmask = mask
        if sums is None and name in self._bnb_sums:  # produced by the epilogue of the data gradient that wrote dout
            sums, sums_parts = self._bnb_sums.pop(name)
        dbias = None
        if conv_bias is not None and self.bias_grad_in_bn and 256 % (C // 4) == 0:
            dbias = self.grad_views[conv_bias]
            self._bias_done.add(conv_bias)
        # sums: [2][C] reductions already produced by the kernel that wrote dout
        nb = y.numel() * y.element_size()
        rd = 2 + (zmask is not None)  # tensors a pass reads: dout, y (+ the ReLU mask source)
        self._prof_hbm('bn_bwd_reduce_kernel + bn_bwd_finalize_kernel + bn_bwd_apply_kernel', nb * ((0 if sums is not None else rd) + rd + 1 + (gout is not None) * (1 + bool(gout_acc))),
                       name + (' [sums given]' if sums is not None else ''))
        check(self.L.dbn_bn_backward_t(self.at, _p(sums), int(sums_parts), y.data_ptr(), _p(zmask), _p(msc), _p(msh), dout.data_ptr(),
                                        self.bufs[name + '/mean'].data_ptr(), self.bufs[name + '/rstd'].data_ptr(),
                                        self.views[name + '.weight'].data_ptr(), dy.data_ptr(), _p(gout), int(gout_acc),
                                        self.grad_views[name + '.weight'].data_ptr(), self.grad_views[name + '.bias'].data_ptr(),
                                        _p(dbias), M, C, self.grad_scale, self.reduce_ws().data_ptr(), self.stream),
              'bn backward ' + name)
        if self.prof:
            self.prof.end()
        return dy

    def up_fwd(self, src, addend, dst, coff=0):
        N, Hs, Ws, C = src.shape
        _, H, W, Cd = dst.shape
        check(self.L.dbn_nearest_up_fwd_t(self.at, src.data_ptr(), _p(addend), dst.data_ptr(), N, Hs, Ws, C, H, W, Cd, coff, self.stream),
              'nearest_up_fwd')

    def up_bwd(self, dbig, dsrc, coff, accumulate):
        N, Hs, Ws, C = dsrc.shape
        _, H, W, Cb = dbig.shape
        check(self.L.dbn_nearest_up_bwd_t(self.at, dbig.data_ptr(), dsrc.data_ptr(), N, Hs, Ws, C, H, W, Cb, coff, int(accumulate),
                                        self.stream), 'nearest_up_bwd')

    # ------------------------------------------------------------------ forward
    def forward(self, x, train):
        m = self.model
        if x.dim() != 4 or x.size(1) != 3:
            raise AssertionError('expected input [N,3,H,W]')
        if not x.is_cuda:
            raise RuntimeError('DBTextModel runs on MI355X only: input must be a HIP tensor')
        N, _, H, W = x.shape
        if H < 32 or W < 32:
            raise ValueError('input must be at least 32x32 (five stride-2 stages); got %dx%d' % (H, W))
        self.ensure_flat()
        if self.at == 2 and train:
            raise RuntimeError("conv math 'fp16' is the inference path (BASELINE configs[4]): call model.eval() first")
        self.repack_params()
        if train:
            self.train_forwards += 1
        x = x.contiguous().float()
        L, st = self.L, self.stream
        self.generation += 1
        bb = m.backbone
        stem16 = self._stem16_ok(bb.conv1, N, H, W)
        if stem16:
            y0, sc, sh = self._stem16_conv_bn(x, bb.conv1, bb.bn1, train)
        x4 = None if stem16 else self.buf('x4', N, H, W, 4 if self.at == 0 else 16)  # 16-bit storage: 16-channel blocks (channels 3.. are zero)
        if stem16:
            pass
        elif self.at == 0:
            check(L.dbn_nchw3_to_nhwc4_t(self.at, x.data_ptr(), x4.data_ptr(), N, H, W, st), 'nchw3_to_nhwc4')
        else:  # ... and, for the stem's weight gradient, (tap, channel) columns without the 13 zero channels per block: one pass writes both
            x4w = self.buf('x4w', N, H, W, 4) if train else None
            check(L.dbn_nchw3_to_nhwc16_and_4_t(self.at, x.data_ptr(), x4.data_ptr(), _p(x4w), N, H, W, st), 'nchw3_to_nhwc16_and_4')
        if not stem16:
            y0, sc, sh = self.conv_bn('backbone.conv1', x4, bb.conv1, 'stem/y', 'backbone.bn1', bb.bn1, train)
        if y0 is None:
            pool = sc  # (the stem kernel pooled already: _stem16_conv_bn's inference form)
        else:
            H0, W0 = y0.shape[1], y0.shape[2]
            pool = self.buf('stem/pool', N, (H0 - 1) // 2 + 1, (W0 - 1) // 2 + 1, 64)  # MaxPool2d(3, 2, 1)
            if train and self.pool_argmax:
                # (late in round 5) the forward records each window's first maximum and the pre-BatchNorm value there: the backward then
                # needs no gradient tensor at the conv's resolution and no separate BatchNorm apply pass (backward(): _stem_pool_backward)
                idx = self.buf('stem/pool_idx', *pool.shape, dtype=torch.uint8)
                ypool = self.buf('stem/pool_y', *pool.shape)
                self._prof_hbm('bnrelu_maxpool_fwd_arg_kernel', (y0.numel() + 2 * pool.numel()) * y0.element_size() + idx.numel())
                check(L.dbn_bnrelu_maxpool_fwd_arg_t(self.at, y0.data_ptr(), sc.data_ptr(), sh.data_ptr(), pool.data_ptr(), idx.data_ptr(),
                                                     ypool.data_ptr(), N, H0, W0, 64, st), 'maxpool fwd (argmax)')
                self._pool_arg_generation = self.generation
            else:
                self._prof_hbm('bnrelu_maxpool_fwd_kernel', (y0.numel() + pool.numel()) * y0.element_size())
                check(L.dbn_bnrelu_maxpool_fwd_t(self.at, y0.data_ptr(), sc.data_ptr(), sh.data_ptr(), pool.data_ptr(), N, H0, W0, 64, st),
                      'maxpool fwd')
            if self.prof:
                self.prof.end()
        self._join_repack()
        self._repack_fpn()
        fpn = m.segmentation_body
        pre = 'segmentation_body.'

        def cbr(name, mod, xin):
            if self._fold_ok(xin, mod.conv, train):  # inference: one launch writes relu(bn(conv(x)))
                return self.conv_bn_act_eval(pre + name + '.conv', xin, mod.conv, name + '/z', mod.bn)
            y, s_, h_ = self.conv_bn(pre + name + '.conv', xin, mod.conv, name + '/y', pre + name + '.bn', mod.bn, train)
            return self.bn_apply(y, s_, h_, name + '/z')

        feats, lateral = [], {}
        cur = pool
        for li in range(1, 5):
            layer = getattr(bb, 'layer%d' % li)
            for bi, blk in enumerate(layer):
                cur = self._block_fwd('backbone.layer%d.%d' % (li, bi), blk, cur, train)
            feats.append(cur)
            if li < 4:  # the FPN's lateral 1x1 conv of this stage (tiny, under-filled grid) runs on the second stream
                self._presplit(cur)
                with self.side_stream():  # beside the next backbone stage
                    lateral[li] = cbr('reduce_conv_c%d' % (li + 1), getattr(fpn, 'reduce_conv_c%d' % (li + 1)), cur)
        c2, c3, c4, c5 = feats
        p5 = cbr('reduce_conv_c5', fpn.reduce_conv_c5, c5)
        self.join_side()
        r2, r3, r4 = lateral[1], lateral[2], lateral[3]
        p4pre = self.buf('p4pre', *r4.shape)
        self.up_fwd(p5, r4, p4pre)
        p4 = cbr('smooth_p4', fpn.smooth_p4, p4pre)
        p3pre = self.buf('p3pre', *r3.shape)
        self.up_fwd(p4, r3, p3pre)
        p3 = cbr('smooth_p3', fpn.smooth_p3, p3pre)
        p2pre = self.buf('p2pre', *r2.shape)
        self.up_fwd(p3, r2, p2pre)
        p2 = cbr('smooth_p2', fpn.smooth_p2, p2pre)
        Hq, Wq = p2.shape[1], p2.shape[2]
        zs = (p2, p3, p4, p5)
        self.fpn_exact = self.fpn_structured and all(Hq == zs[g].shape[1] << g and Wq == zs[g].shape[2] << g for g in range(4))
        f_folded = None
        if self.fpn_exact and not train and self.fold_eval_bn and not self._use_planes and fpn.conv[0].cout % 128 == 0 and zs[0].shape[3] % 16 == 0:
            f_folded = self._fpn_conv_eval(pre + 'conv.0', fpn.conv[0], zs, 'fpn/z', fpn.conv[1])
        elif self.fpn_exact:
            # conv over [p2 | up2(p3) | up4(p4) | up8(p5)] without building the concat: per level a transposed conv
            # (k = f+2, stride f, pad 1) with combined weights, accumulated into one output (47 % of the dense MACs)
            fy, s_, h_ = self._fpn_conv_forward(pre + 'conv.0', fpn.conv[0], zs, 'fpn/y', pre + 'conv.1', fpn.conv[1], train)
        else:
            cat = self.buf('cat', N, Hq, Wq, 256)
            for i, t in enumerate(zs):
                self.up_fwd(t, None, cat, coff=64 * i)
            fy, s_, h_ = self.conv_bn(pre + 'conv.0', cat, fpn.conv[0], 'fpn/y', pre + 'conv.1', fpn.conv[1], train)
        head = m.segmentation_head
        f_act = None
        if f_folded is not None:
            f = f_folded
        elif self.lazy_act(fy, (head.binarize[0], head.thresh[0]), train):
            self._drop_buf('fpn/z')  # (never written: the two head convs and their weight gradients read fpn/y through its BatchNorm's coefficients)
            f, f_act = fy, (s_, h_)
        else:
            f = self.bn_apply(fy, s_, h_, 'fpn/z')
        z1 = {}
        fused_tail = (not train and self.head16 and self.convt16 and self.at != 0 and f.shape[3] == 256 and head.binarize[3].k == 2
                      and head.binarize[3].cin == 64 and head.binarize[3].cout == 64
                      and bool(L.dbn_head16_eligible(self.at, N, f.shape[1], f.shape[2])))

        def branch(br):
            seq = getattr(head, br)
            hp = 'segmentation_head.%s.' % br
            if f_act is None and self._fold_ok(f, seq[0], train):
                za = self.conv_bn_act_eval(hp + '0', f, seq[0], br + '/z0', seq[1])
            else:
                ya, s_, h_ = self.conv_bn(hp + '0', f, seq[0], br + '/y0', hp + '1', seq[1], train, x_act=f_act)
                za = self.bn_apply(ya, s_, h_, br + '/z0')
            if fused_tail:
                z1[br] = (za, None, None)  # (inference, 16-bit storage: the whole tail of the branch is one kernel below)
                return
            yb, s_, h_ = self.convT_bn(hp + '3', za, seq[3], br + '/y1', hp + '4', seq[4], train)
            z1[br] = (yb, s_, h_)  # BN + ReLU of the two largest activations is applied inside the head-tail kernels

        # the two branches are independent until the head-tail kernel: the threshold branch may run on the second stream
        if self.overlap_head_branches:
            self._presplit(f)
            with self.side_stream():
                branch('thresh')
            branch('binarize')
            self.join_side()
        else:
            branch('thresh')
            branch('binarize')
        ch = 3 if train else 2
        (yb_, sb_, hb_), (yt_, st_, ht_) = z1['binarize'], z1['thresh']
        if fused_tail:
            Hq, Wq = yb_.shape[1], yb_.shape[2]
            resample = (4 * Hq, 4 * Wq) != (H, W)
            out = device_empty((N, 2, H, W), x.device)
            head_out = self.fbuf('head/out', N, 2, 4 * Hq, 4 * Wq) if resample else out
            coef = []
            for br in ('binarize', 'thresh'):
                seq, hp = getattr(head, br), 'segmentation_head.%s.' % br
                bn = seq[4]
                sc_, sh_ = self.fbuf(hp + '4/scale', 64), self.fbuf(hp + '4/shift', 64)
                check(L.dbn_bn_eval_coef(64, bn.weight.data_ptr(), bn.bias.data_ptr(), bn.running_mean.data_ptr(), bn.running_var.data_ptr(), bn.eps,
                                         sc_.data_ptr(), sh_.data_ptr(), st), 'bn eval ' + hp + '4')
                coef.append((self._convt16_panel(hp + '3', seq[3]), seq[3].bias, sc_, sh_, seq[6]))
            (pb, b1b, scb, shb, b6), (pt, b1t, sct, sht, t6) = coef
            if self.prof:
                self.prof.begin('head16_tail_eval_kernel<%d>' % self.at, 0.0, float(yb_.element_size()) * N * Hq * Wq * 128 + 4.0 * N * 16 * Hq * Wq * 2)
            check(L.dbn_head16_tail_eval_t(self.at, yb_.data_ptr(), yt_.data_ptr(), pb.data_ptr(), pt.data_ptr(), _p(b1b), _p(b1t), scb.data_ptr(),
                                           shb.data_ptr(), sct.data_ptr(), sht.data_ptr(), b6.weight.data_ptr(), t6.weight.data_ptr(),
                                           b6.bias.data_ptr(), t6.bias.data_ptr(), head_out.data_ptr(), N, Hq, Wq, st), 'head16_tail_eval')
            if self.prof:
                self.prof.end()
            if resample:
                check(L.dbn_bilinear_fwd(head_out.data_ptr(), out.data_ptr(), N * 2, 4 * Hq, 4 * Wq, H, W, st), 'bilinear_fwd')
            return out
        Hh, Wh = yb_.shape[1], yb_.shape[2]
        resample = (2 * Hh, 2 * Wh) != (H, W)  # only when H or W is not a multiple of 32 (models.py:43-46)
        out = device_empty((N, ch, H, W), x.device)
        head_out = self.fbuf('head/out', N, ch, 2 * Hh, 2 * Wh) if resample else out
        b6, t6 = head.binarize[6], head.thresh[6]
        if self.prof:  # reads 2 x 64ch at half resolution, writes `ch` full-resolution maps
            self.prof.begin('head_tail_fwd_kernel', 0.0, float(yb_.element_size()) * N * Hh * Wh * 128 + 4.0 * N * 4 * Hh * Wh * ch)
        check(L.dbn_head_tail_fwd_t(self.at, yb_.data_ptr(), yt_.data_ptr(), b6.weight.data_ptr(), t6.weight.data_ptr(), b6.bias.data_ptr(),
                                  t6.bias.data_ptr(), sb_.data_ptr(), hb_.data_ptr(), st_.data_ptr(), ht_.data_ptr(),
                                  head_out.data_ptr(), N, Hh, Wh, ch, float(head.k), st), 'head_tail_fwd')
        if self.prof:
            self.prof.end()
        if resample:
            check(L.dbn_bilinear_fwd(head_out.data_ptr(), out.data_ptr(), N * ch, 2 * Hh, 2 * Wh, H, W, st), 'bilinear_fwd')
        if train:
            self.saved_generation = self.generation
            self.saved_out = head_out
            self.saved_shape = (N, H, W, Hh, Wh, resample)
        return out

    # ---- the stem in 16-bit storage (round 5, csrc/stem16.hip): conv 7x7 / 2 on a PACKED, zero-bordered 4-channel image instead of 16-channel
    # blocks with 13 zero channels (K = 224 instead of 784, no 16-channel copy of the input).  DBN_STEM16=0: the generic 16-bit loop.
    stem16 = os.environ.get('DBN_STEM16', '1') == '1'

    def _stem16_ok(self, conv, N, H, W):
        return (self.stem16 and self.at != 0 and (conv.k, conv.stride, conv.padding, conv.cin, conv.cout) == (7, 2, 3, 3, 64) and conv.bias is None
                and bool(self.L.dbn_stem16_eligible(self.at, N, H, W)))

    def _stem16_conv_bn(self, x, conv, bn, train):
        """x: the fp32 NCHW input.  Returns (y, scale, shift) like conv_bn; in training also leaves the packed [N,H,W,4] image in 'x4w'.
        Inference with the fused pool: (None, pooled, None)."""
        L, st = self.L, self.stream
        N, _, H, W = x.shape
        Hp, Wp = L.dbn_stem16_padded_h(H), L.dbn_stem16_padded_w(W)
        dt = ACT_DTYPES[self.at]
        xp = self.bufs.get('x4p')
        if xp is None or tuple(xp.shape) != (N, Hp, Wp, 4) or xp.dtype != dt or xp.device != self.flat.device:
            xp = torch.zeros((N, Hp, Wp, 4), device=self.flat.device, dtype=dt)  # (the zero border is written once, here)
            self.bufs['x4p'] = xp
        x4w = self.buf('x4w', N, H, W, 4) if train else None
        check(L.dbn_nchw3_to_padded4_t(self.at, x.data_ptr(), xp.data_ptr(), _p(x4w), N, H, W, st), 'nchw3_to_padded4')
        w = conv.weight
        key = ('backbone.conv1', 'stem16', self.kind)
        stamp = (w._version, self.param_epoch, w.data_ptr())
        ent = self.packs.get(key)
        if ent is None or ent[1] != stamp:
            panel = ent[0] if ent is not None else torch.empty(L.dbn_stem16_panel_bytes(), device=w.device, dtype=torch.uint8)
            check(L.dbn_stem16_pack(self.kind, w.data_ptr(), panel.data_ptr(), st), 'stem16_pack')
            self.packs[key] = (panel, stamp)
        panel = self.packs[key][0]
        Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        name = 'backbone.bn1'
        if not train and self.stem16_pool and bool(L.dbn_stem16_pool_eligible(self.at, N, H, W)):
            # inference (round 5): conv + BatchNorm + ReLU + MaxPool2d(3, 2, 1) in ONE launch — the 64-channel conv output never reaches memory
            sc, sh = self.fbuf(name + '/scale', 64), self.fbuf(name + '/shift', 64)
            check(L.dbn_bn_eval_coef(64, bn.weight.data_ptr(), bn.bias.data_ptr(), bn.running_mean.data_ptr(), bn.running_var.data_ptr(), bn.eps,
                                     sc.data_ptr(), sh.data_ptr(), st), 'bn eval ' + name)
            pool = self.buf('stem/pool', N, (Ho - 1) // 2 + 1, (Wo - 1) // 2 + 1, 64)
            if self.prof:
                self.prof.begin('stem7x7_pool_b16_kernel<%d>' % self.at, 2.0 * N * Ho * Wo * 64 * 3 * 49, float(2 * (N * Hp * Wp * 4 + pool.numel())),
                                'fwd backbone.conv1 + bn1 + relu + maxpool')
            check(L.dbn_stem16_conv_bn_relu_pool_t(self.at, xp.data_ptr(), panel.data_ptr(), sc.data_ptr(), sh.data_ptr(), pool.data_ptr(), N, H, W,
                                                   st), 'stem16 conv+bn+relu+pool')
            if self.prof:
                self.prof.end()
            return None, pool, None
        y = self.buf('stem/y', N, Ho, Wo, 64)
        if self.prof:
            self.prof.begin('stem7x7_b16_kernel<%d>' % self.at, 2.0 * N * Ho * Wo * 64 * 3 * 49, float(2 * (N * Hp * Wp * 4 + N * Ho * Wo * 64)),
                            'fwd backbone.conv1')
        if train and self.fuse_bn_stats:
            sc, sh = self.fbuf(name + '/scale', 64), self.fbuf(name + '/shift', 64)
            mu, rs = self.fbuf(name + '/mean', 64), self.fbuf(name + '/rstd', 64)
            ws = self.scratch('_conv_bn_ws', (3 * 64 + 1) * L.dbn_stem16_rows())
            check(L.dbn_stem16_conv_bn_t(self.at, xp.data_ptr(), panel.data_ptr(), y.data_ptr(), N, H, W, bn.weight.data_ptr(), bn.bias.data_ptr(),
                                         bn.eps, bn.momentum, bn.running_mean.data_ptr(), bn.running_var.data_ptr(), sc.data_ptr(),
                                         sh.data_ptr(), mu.data_ptr(), rs.data_ptr(), ws.data_ptr(), st), 'stem16 conv+bn')
            self.nbt_pending[name] = self.nbt_pending.get(name, 0) + 1
        else:
            check(L.dbn_stem16_conv_bn_t(self.at, xp.data_ptr(), panel.data_ptr(), y.data_ptr(), N, H, W, None, None, 0.0, 0.0, None, None,
                                         None, None, None, None, None, st), 'stem16 conv')
            sc = sh = None
        if self.prof:
            self.prof.end()
        if sc is None:
            sc, sh = self.bn_coef(name, bn, y, train)
        return y, sc, sh

    # ---- deformable conv2 (resnet.py:54-65,81-82,111-124,145-146): offsets conv -> bilinear im2col -> 1x1 GEMM
    def _offset_conv(self, name, oc):
        """conv2_offset has 18 output channels; the GEMM kernels want multiples of 64: zero-padded copy of its
        weight/bias (the offsets then live in channels 0..17 of a 64-channel map)."""
        w = oc.weight
        stamp = (w._version, self.param_epoch, w.data_ptr())
        ent = self.packs.get((name, 'pad64'))
        if ent is None or ent[1] != stamp:
            wp, bp = ent[0] if ent is not None else (torch.zeros(64, oc.cin, oc.k, oc.k, device=w.device),
                                                     torch.zeros(64, device=w.device))
            wp[:w.shape[0]].copy_(w.detach())
            bp[:w.shape[0]].copy_(oc.bias.detach())
            self.packs[(name, 'pad64')] = ((wp, bp), stamp)
        wp, bp = self.packs[(name, 'pad64')][0]
        return _VirtualConv(oc.cin, 64, oc.k, oc.stride, oc.padding, wp, bp), stamp[:2]

    def _cols_conv(self, name, conv):
        """The deformable conv's weight [O,C,3,3] as the 1x1 conv [O, 9C] over the sampled columns ((tap, channel) order)."""
        w = conv.weight
        stamp = (w._version, self.param_epoch, w.data_ptr())
        ent = self.packs.get((name, 'ohwi'))
        if ent is None or ent[1] != stamp:
            wp = ent[0] if ent is not None else device_empty((conv.cout, conv.k * conv.k * conv.cin, 1, 1), w.device)
            check(self.L.dbn_permute_weight(w.data_ptr(), wp.data_ptr(), conv.cout, conv.cin, conv.k * conv.k, 1, 1.0, self.stream),
                  'permute_weight')
            self.packs[(name, 'ohwi')] = (wp, stamp)
        wp = self.packs[(name, 'ohwi')][0]
        return _VirtualConv(conv.k * conv.k * conv.cin, conv.cout, 1, 1, 0, wp, None), stamp[:2]

    def _deform_conv_bn(self, name, x, blk, out_name, bn_name, bn, train):
        conv = blk.conv2
        voc, ver = self._offset_conv(name + '.conv2_offset', blk.conv2_offset)
        off = self.conv_fwd(name + '.conv2_offset', x, voc, name + '/offset', version=ver)
        N, H, W, C = x.shape
        Ho, Wo = off.shape[1], off.shape[2]
        cols = self.buf(name + '/cols', N, Ho, Wo, conv.k * conv.k * C)
        check(self.L.dbn_deform_im2col_t(self.at, x.data_ptr(), off.data_ptr(), cols.data_ptr(), N, H, W, C, Ho, Wo, conv.k, conv.k,
                                         conv.stride, conv.padding, 64, self.stream), 'deform_im2col')
        vconv, ver2 = self._cols_conv(name + '.conv2', conv)
        return self.conv_bn(name + '.conv2', cols, vconv, out_name, bn_name, bn, train, version=ver2)

    flush_before_pool = os.environ.get('DBN_FLUSH_BEFORE_POOL', '1') == '1'
    dcn_gather = os.environ.get('DBN_DCN_GATHER', '1') == '1'  # the sampling adjoint as a gather (0: round 3's fixed-point scatter)
    dcn_gather_max_offset = float(os.environ.get('DBN_DCN_GATHER_MAX_OFFSET', '16'))  # ... while the previous step's max |offset| of the layer is below this
    _dcn_table = _dcn_host = _dcn_event = None

    def _dcn_read_back(self):
        """Start of a backward pass: the per-layer offset maxima the previous pass left behind.  Deformable nets only (`_dcn_event` is None
        otherwise).  The wait is on the PREVIOUS backward pass's copy: it bounds the host's run-ahead to one step on those nets — the price
        of a deterministic choice of the adjoint's form (taking the value only "if it has already landed" would make the form, and with it
        the low bits of the gradients, depend on host timing: test_deformable_backbone_step_is_bit_reproducible).  Never under stream
        capture (an event wait inside a captured pass is illegal; a replayed graph keeps the forms it was captured with)."""
        if self._dcn_event is not None and not torch.cuda.is_current_stream_capturing():
            self._dcn_event.synchronize()
            self._dcn_event = None
            vals = self._dcn_host.view(torch.float32)
            for name, slot in self._dcn_slots.items():
                v = float(vals[slot]) if slot < 64 else float('inf')
                self._dcn_E[name] = v if v == v else float('inf')  # (0x7FC00000: a non-finite offset)

    def _dcn_send_back(self):
        """End of a backward pass: the maxima go to pinned host memory behind the pass (not under stream capture: a replayed graph keeps
        the forms it was captured with)."""
        if self._dcn_slots and self._dcn_table is not None and not torch.cuda.is_current_stream_capturing():
            self._dcn_host.copy_(self._dcn_table, non_blocking=True)
            self._dcn_event = torch.cuda.Event()
            self._dcn_event.record(torch.cuda.current_stream(self.flat.device))

    def _deform_conv_bwd(self, name, blk, dy, x, dx):
        """dy: gradient of the deformable conv's output; writes dx (gradient of its input x), the conv2 / conv2_offset
        weight gradients and the offset-bias gradient."""
        conv, oc = blk.conv2, blk.conv2_offset
        B, G = self.bufs, self.grad_views
        cols, off = B[name + '/cols'], B[name + '/offset']
        N, H, W, C = x.shape
        Ho, Wo = off.shape[1], off.shape[2]
        T = conv.k * conv.k
        vconv, ver2 = self._cols_conv(name + '.conv2', conv)
        t = self.fbuf(name + '/dw_cols', conv.cout, T * C, 1, 1)
        self.wgrad(name + '.conv2 (cols)', dy, cols, conv.cout, T * C, 1, 1, 0, t)
        check(self.L.dbn_permute_weight(t.data_ptr(), G[name + '.conv2.weight'].data_ptr(), conv.cout, C, T, 0, 1.0, self.stream),
              'permute_weight')
        dcols = self.buf(name + '/dcols', *cols.shape)
        self.conv_dgrad(name + '.conv2', dy, vconv, dcols, False, version=ver2)
        doff = self.buf(name + '/doffset', N, Ho, Wo, 64)
        # The adjoint of the sampling.  Round 5: a gather (a team per input pixel, fixed summation order, plain fp32) whose search window grows
        # with the largest learned offset — 2-4x faster than round 3's fixed-point scatter up to max |offset| ~ 20 pixels, slower beyond
        # (tools/dcn_probe.py).  Which one runs is decided per layer from the PREVIOUS step's maximum (read back after that step: no
        # synchronisation inside a step; both forms are deterministic, and so is the choice).
        slot = self._dcn_slots.setdefault(name, len(self._dcn_slots))
        if self._dcn_table is None:
            self._dcn_table = torch.zeros(64, dtype=torch.int32, device=x.device)
            self._dcn_host = torch.zeros(64, dtype=torch.int32).pin_memory()
        gather = self.dcn_gather and C <= 512 and slot < 64 and self._dcn_E.get(name, 0.0) <= self.dcn_gather_max_offset
        self.dcn_forms['gather' if gather else 'scatter'] += 1  # (observability: which adjoint the layers of this engine have taken so far)
        if gather:
            ws = self.scratch('_dcn_gather_ws', self.L.dbn_deform_col2im_gather_ws_bytes(N, Ho, Wo) // 4 + 1)
            check(self.L.dbn_deform_col2im_gather_t(self.at, dcols.data_ptr(), x.data_ptr(), off.data_ptr(), dx.data_ptr(), doff.data_ptr(),
                                                    0, ws.data_ptr(), N, H, W, C, Ho, Wo, conv.k, conv.k, conv.stride, conv.padding, 64,
                                                    self.stream), 'deform_col2im_gather')
            if slot < 64:
                self._dcn_table[slot:slot + 1].copy_(ws.view(torch.int32)[:1])  # (ws[0]: the maximum the gather took its window from)
        else:
            if slot < 64:
                check(self.L.dbn_deform_offset_absmax_t(self.at, off.data_ptr(), off.numel(), self._dcn_table.data_ptr() + 4 * slot,
                                                        self.stream), 'deform_offset_absmax')
            ws = self.scratch('_dcn_col2im_ws', self.L.dbn_deform_col2im_ws_bytes(N, H, W, C, Ho, Wo, conv.k, conv.k) // 4 + 1)
            check(self.L.dbn_deform_col2im_t(self.at, dcols.data_ptr(), x.data_ptr(), off.data_ptr(), dx.data_ptr(), doff.data_ptr(), 0,
                                             ws.data_ptr(), N, H, W, C, Ho, Wo, conv.k, conv.k, conv.stride, conv.padding, 64,
                                             self.stream), 'deform_col2im')
        voc, ver = self._offset_conv(name + '.conv2_offset', oc)
        tg = self.fbuf(name + '/dw_offset', 64, C, oc.k, oc.k)
        self.wgrad(name + '.conv2_offset', doff, x, 64, C, oc.k, oc.stride, oc.padding, tg)
        nreal = oc.weight.shape[0]
        G[name + '.conv2_offset.weight'].copy_(tg[:nreal])
        tb = self.fbuf(name + '/db_offset', 64)
        self.col_sum(doff, tb)
        G[name + '.conv2_offset.bias'].copy_(tb[:nreal])
        self.conv_dgrad(name + '.conv2_offset', doff, voc, dx, True, version=ver)

    def _conv2_bn(self, name, blk, z1, train):
        if getattr(blk, 'with_dcn', False):
            return self._deform_conv_bn(name, z1, blk, name + '/y2', name + '.bn2', blk.bn2, train)
        return self.conv_bn(name + '.conv2', z1, blk.conv2, name + '/y2', name + '.bn2', blk.bn2, train)

    def _conv2_bwd(self, name, blk, dy2, z1, dz1):
        if getattr(blk, 'with_dcn', False):
            self._deform_conv_bwd(name, blk, dy2, z1, dz1)
        else:
            self.conv_wgrad(name + '.conv2', dy2, z1, blk.conv2)
            self.conv_dgrad(name + '.conv2', dy2, blk.conv2, dz1, False, consumer=(name + '.bn1', self.bufs[name + '/y1'], None))

    def _block_fwd(self, name, blk, x, train):
        """BasicBlock (resnet.py:70-91) or Bottleneck (resnet.py:135-159)."""
        if not train and self.fold_eval_bn:
            out = self._block_fwd_eval(name, blk, x)
            if out is not None:
                return out
        y1, s1, h1 = self.conv_bn(name + '.conv1', x, blk.conv1, name + '/y1', name + '.bn1', blk.bn1, train)
        if not getattr(blk, 'with_dcn', False) and self.lazy_act(y1, (blk.conv2, ), train):
            self._drop_buf(name + '/z1')  # (never written: conv2 and its weight gradient read y1 through bn1's coefficients)
            y2, s2, h2 = self.conv_bn(name + '.conv2', y1, blk.conv2, name + '/y2', name + '.bn2', blk.bn2, train, x_act=(s1, h1))
        else:
            z1 = self.bn_apply(y1, s1, h1, name + '/z1')
            y2, s2, h2 = self._conv2_bn(name, blk, z1, train)
        if hasattr(blk, 'conv3'):
            z2 = self.bn_apply(y2, s2, h2, name + '/z2')
            ylast, s2, h2 = self.conv_bn(name + '.conv3', z2, blk.conv3, name + '/y3', name + '.bn3', blk.bn3, train)
        else:
            ylast = y2
        if blk.downsample is not None:
            yd, sd, hd = self.conv_bn(name + '.downsample.0', x, blk.downsample[0], name + '/yd', name + '.downsample.1',
                                      blk.downsample[1], train)
            out = self.bn_apply(ylast, s2, h2, name + '/out', relu=True, res=yd, rsc=sd, rsh=hd)
        else:
            out = self.bn_apply(ylast, s2, h2, name + '/out', relu=True, res=x)
        self.bufs[name + '/in'] = x
        return out

    def _block_fwd_eval(self, name, blk, x):
        """Inference form of a BasicBlock / Bottleneck (resnet.py:70-91,135-159 under model.eval()): every conv writes its activation
        itself — BatchNorm folded into the weights, ReLU and the residual addition in the epilogue.  None: a conv of the block cannot
        take that form (deformable conv2, a split-K launch): the caller runs the general chain."""
        convs = [(name + '.conv1', blk.conv1, blk.bn1), (name + '.conv2', blk.conv2, blk.bn2)]
        if hasattr(blk, 'conv3'):
            convs.append((name + '.conv3', blk.conv3, blk.bn3))
        if getattr(blk, 'with_dcn', False):
            return None
        cur = x
        for _, conv, _ in convs:  # is every launch eligible?  (the shapes follow from the geometry alone)
            if not self._fold_ok(cur, conv, False):
                return None
            N, H, W, _ = cur.shape
            cur = _Shape(N, (H + 2 * conv.padding - conv.k) // conv.stride + 1, (W + 2 * conv.padding - conv.k) // conv.stride + 1, conv.cout)
        if blk.downsample is not None and not self._fold_ok(x, blk.downsample[0], False):
            return None
        cur, res = x, x
        beside = blk.downsample is not None and self.eval_downsample_beside and self.overlap_wgrad
        if blk.downsample is not None:
            # the projection shortcut reads x like conv1 and is needed by the LAST conv only: on the second stream, beside conv1 (round 5: at
            # cfg5 the three 1x1 / stride-2 launches — a K of 4-16 k-steps, two ring stages per workgroup — were 0.28 + 0.22 + 0.21 ms of the
            # main stream)
            if beside:
                with self.side_stream():
                    res = self.conv_bn_act_eval(name + '.downsample.0', x, blk.downsample[0], name + '/yd', blk.downsample[1], relu=False)
            else:
                res = self.conv_bn_act_eval(name + '.downsample.0', x, blk.downsample[0], name + '/yd', blk.downsample[1], relu=False)
        for i, (cname, conv, bn) in enumerate(convs):
            last = i == len(convs) - 1
            if last and beside:
                self.join_side()
            cur = self.conv_bn_act_eval(cname, cur, conv, name + ('/out' if last else '/z%d' % (i + 1)), bn, relu=True, res=res if last else None)
        return cur

    eval_downsample_beside = os.environ.get('DBN_EVAL_DOWNSAMPLE_BESIDE', '1') == '1'

    # 16-bit storage: level 0 of the pyramid conv through the pixel-patch kernel.  0: off (default); 1: in train mode; 2: in inference too.
    # Measured: `tools/cfg_timing.py` (eager loop, host-paced) bf16 16 x 640^2 12.02 -> 11.5 ms, twice — but bench.py's step (resident inputs,
    # collector off, GPU-paced), interleaved on one box: 1655 / 1654 images/s without it, 1643 / 1638 with it; cfg5 fp16 inference 14.95 ->
    # 15.7 ms (the second pass over the 1.7 GB output — write, read, write — costs more than the faster level 0 saves).  Off.
    fpn_level0_patch = int(os.environ.get('DBN_FPN_LV0_PATCH', '0'))

    def _fpn_level0_patch_ok(self, z0, Co, train=False):
        N, H, W, Cg = z0.shape
        return (self.fpn_level0_patch >= (1 if train else 2) and self.at != 0 and self.ns == 1 and Cg % 32 == 0 and H % 8 == 0 and W % 16 == 0
                and Co % 128 == 0)

    def _fpn_conv_eval(self, name, conv, zs, out_name, bn):
        """Inference form of the FPN output conv + BatchNorm + ReLU (segmentation_body.py:55-61,75-76): the pyramid conv on the combined
        weights of the FOLDED filters, ReLU in its epilogue."""
        N, H, W, Cg = zs[0].shape
        Co = conv.cout
        vconv, ver = self._folded(name, conv, bn)
        fname = name + '#fold'
        wds, wver = self._fpn_combined_weights(fname, vconv, Cg, version=ver)
        z = self.buf(out_name, N, H, W, Co)
        wpk = [self.pack('%s#f%d' % (fname, g), wds[g], 1, 1 << g, version=wver) for g in range(4)]
        lv0 = _VirtualConv(Cg, Co, 3, 1, 1, wds[0], vconv.bias)
        first = int(self.fpn_level0_winograd and wds[0].shape[0] == Cg and self._winograd_ok(zs[0], lv0))
        if self._fpn_level0_patch_ok(zs[0], Co):  # 16-bit storage: level 0 through the pixel-patch kernel (see _fpn_conv_forward)
            if self.prof:
                self._prof_igemm(N * H * W, Co, 2.0 * N * H * W * Co * Cg * 9, 'fwd %s level 0' % name, 1, (N, H, W, Cg, H, W, 3, 1, 1))
            check(self.L.dbn_igemm_act_t(self.at, self.ns, zs[0].data_ptr(), wpk[0].data_ptr(), vconv.bias.data_ptr(), None, 0, z.data_ptr(), N, H, W,
                                         Cg, H, W, Co, 3, 3, 1, 1, 1, 0, self.stream), 'igemm fpn level 0')
            if self.prof:
                self.prof.end()
            first = 1
        elif first:
            up = self._winograd_panel(fname + '#lv0', wds[0], Cg, dgrad=1, version=wver)
            if self.prof:
                self.prof.begin('winograd_f32_kernel', 2.0 * N * H * W * Co * Cg * 4, 0.0, 'fwd %s level 0' % name)
            check(self.L.dbn_winograd_conv_act_f32(zs[0].data_ptr(), up.data_ptr(), vconv.bias.data_ptr(), None, 0, z.data_ptr(), N, H, W, Cg, Co,
                                                   self.stream), 'winograd ' + name)
            if self.prof:
                self.prof.end()
        if self.prof:
            flops = sum(2.0 * N * t.shape[1] * t.shape[2] * Cg * Co * ((1 << g) + 2)**2 for g, t in enumerate(zs) if g >= first)
            bn_tile = 256 if self.L.dbn_pyramid_wide_would_run(self.at, N, H, W, Cg, Co) else 128  # (round 6: the 128 x 256 tile on large 16-bit launches)
            waves = '1,4' if (self.at in (1, 2) and self.ns == 1) else '2,2'  # (16-bit storage: 1 x 4 waves, see _prof_igemm)
            self.prof.begin('igemm_f32_kernel<128,%d,%s,3,%d,%d,false,0,true>' % (bn_tile, waves, self.ns, self.at), flops, 0.0, 'fwd %s (pyramid)' % name)
        check(self.L.dbn_pyramid_conv_act_t(first, self.at, *[t.data_ptr() for t in zs], *[w_.data_ptr() for w_ in wpk], vconv.bias.data_ptr(), 1,
                                            z.data_ptr(), N, H, W, Cg, Co, self.ns, self.stream), 'pyramid_conv_act')
        if self.prof:
            self.prof.end()
        return z

    # ----------------------------------------------------------------- backward
    def backward(self, dpreds):
        """Consumes d(loss)/d(preds) [N,3,H,W]; fills the flat gradient buffer."""
        if self.saved_generation != self.generation:
            raise RuntimeError('backward() without a matching train-mode forward (activations were overwritten)')
        m, L, st = self.model, self.L, self.stream
        N, H, W, Hh, Wh, resample = self.saved_shape
        B = self.bufs
        self._bias_done = set()
        self._bnb_sums = {}
        self._dcn_read_back()
        # (a pass that raised between queueing and flushing must not leave launches holding the PREVIOUS batch's tensors behind)
        self._wgrad_fifo, self._reduce_pending = [], []
        self._slab_free = [None, None]  # (the previous pass's reductions were joined: no event of it is waited for again)
        out = self.saved_out  # head output before the (optional) final resample
        dpreds = dpreds.contiguous()
        assert dpreds.shape == (N, 3, H, W)
        if resample:
            dhead = self.fbuf('head/dout', *out.shape)
            check(L.dbn_bilinear_bwd(dpreds.data_ptr(), dhead.data_ptr(), N * 3, 2 * Hh, 2 * Wh, H, W, st), 'bilinear_bwd')
            dpreds = dhead
        head = m.segmentation_head
        b6, t6 = head.binarize[6], head.thresh[6]
        dz1b = self.buf('binarize/dz1', N, Hh, Wh, 64)
        dz1t = self.buf('thresh/dz1', N, Hh, Wh, 64)
        ws = self.scratch('_head_ws', L.dbn_head_tail_bwd_ws_floats())
        G = self.grad_views
        hb, ht = 'segmentation_head.binarize.4', 'segmentation_head.thresh.4'
        bn_sums = self.fbuf('head/bn4_sums', 4, 64)  # the kernel also reduces what the two BatchNorm backwards need
        # reads both 64-channel ConvT outputs, the maps and their gradients; writes both 64-channel gradients
        self._prof_hbm('head_tail_bwd_kernel + fold_head_grads_kernel + fold_partials_d_kernel', 4.0 * dz1b.element_size() * dz1b.numel() + 4.0 * (out.numel() + dpreds.numel()))
        check(L.dbn_head_tail_bwd_t(self.at, B['binarize/y1'].data_ptr(), B['thresh/y1'].data_ptr(), b6.weight.data_ptr(),
                                  t6.weight.data_ptr(), out.data_ptr(), dpreds.data_ptr(), B[hb + '/scale'].data_ptr(),
                                  B[hb + '/shift'].data_ptr(), B[ht + '/scale'].data_ptr(), B[ht + '/shift'].data_ptr(),
                                  B[hb + '/mean'].data_ptr(), B[hb + '/rstd'].data_ptr(), B[ht + '/mean'].data_ptr(),
                                  B[ht + '/rstd'].data_ptr(), bn_sums.data_ptr(), dz1b.data_ptr(), dz1t.data_ptr(),
                                  G['segmentation_head.binarize.6.weight'].data_ptr(),
                                  G['segmentation_head.binarize.6.bias'].data_ptr(),
                                  G['segmentation_head.thresh.6.weight'].data_ptr(),
                                  G['segmentation_head.thresh.6.bias'].data_ptr(), N, Hh, Wh, 3, float(head.k),
                                  self.grad_scale, ws.data_ptr(), st), 'head_tail_bwd')
        if self.prof:
            self.prof.end()
        f, f_act = B.get('fpn/z'), None
        if f is None:  # apply-on-load (see forward)
            f, f_act = B['fpn/y'], (B['segmentation_body.conv.1/scale'], B['segmentation_body.conv.1/shift'])
        df = self.buf('fpn/dz', *f.shape)
        fpn = m.segmentation_body
        pre = 'segmentation_body.'
        for i, (br, dz1) in enumerate((('binarize', dz1b), ('thresh', dz1t))):
            # (running the two branches' backward on two streams was tried: the join before the FPN backward makes the main
            # stream wait for every queued weight gradient and costs more than it gains)
            seq = getattr(head, br)
            hp = 'segmentation_head.%s.' % br
            dy1 = self.bn_backward(hp + '4', B[br + '/y1'], 'self', dz1, br + '/dy1', sums=bn_sums[2 * i:2 * i + 2],
                                   conv_bias=hp + '3.bias' if seq[3].bias is not None else None)
            dz0 = self.buf(br + '/dz0', *B[br + '/z0'].shape)
            self.convT_bwd(hp + '3', dy1, B[br + '/z0'], seq[3], dz0, consumer=(hp + '1', B[br + '/y0'], None))
            dy0 = self.bn_backward(hp + '1', B[br + '/y0'], 'self', dz0, br + '/dy0',
                                   conv_bias=hp + '0.bias' if seq[0].bias is not None else None)
            self.conv_wgrad(hp + '0', dy0, f, seq[0], x_act=f_act)
            # (the second branch's data gradient is the last writer of df: it carries the sums of the FPN output BatchNorm)
            self.conv_dgrad(hp + '0', dy0, seq[0], df, accumulate=(i > 0),
                            consumer=(pre + 'conv.1', B['fpn/y'], None) if i > 0 else None)
        fpn = m.segmentation_body
        pre = 'segmentation_body.'
        dfy = self.bn_backward(pre + 'conv.1', B['fpn/y'], 'self', df, 'fpn/dy',
                               conv_bias=pre + 'conv.0.bias' if fpn.conv[0].bias is not None else None)
        dP = {}
        levels = ('smooth_p2', 'smooth_p3', 'smooth_p4', 'reduce_conv_c5')
        zs = [B[nm + '/z'] for nm in levels]
        if self.fpn_exact:
            # conv over [p2 | up2(p3) | up4(p4) | up8(p5)]: per level a (f+2)x(f+2) stride-f conv with combined weights
            # (47 % of the MACs of the dense 256->256 3x3 data/weight gradients, no concat-gradient tensor)
            self._fpn_conv_backward(pre + 'conv.0', fpn.conv[0], dfy, levels, zs, dP)
        else:
            cat = B['cat']
            self.conv_wgrad(pre + 'conv.0', dfy, cat, fpn.conv[0])
            dcat = self.buf('dcat', *cat.shape)
            self.conv_dgrad(pre + 'conv.0', dfy, fpn.conv[0], dcat, False)
            for i, nm in enumerate(levels):
                d = self.buf(nm + '/dz', *zs[i].shape)
                self.up_bwd(dcat, d, 64 * i, False)
                dP[nm] = d

        def cbr_bwd(name, mod, xin, dz, dx, dx_acc, consumer=None):
            """consumer: lateral conv whose BatchNorm consumes dx next (dx = gradient of `up(..) + lateral`: same tensor)"""
            dy = self.bn_backward(pre + name + '.bn', B[name + '/y'], 'self', dz, name + '/dy',
                                  conv_bias=pre + name + '.conv.bias' if mod.conv.bias is not None else None)
            self.conv_wgrad(pre + name + '.conv', dy, xin, mod.conv)
            self.conv_dgrad(pre + name + '.conv', dy, mod.conv, dx, dx_acc,
                            consumer=(pre + consumer + '.bn', B[consumer + '/y'], None) if consumer else None)

        # gradient slots of the backbone features (written first by the FPN reduce convs)
        lastb = {i: len(getattr(m.backbone, 'layer%d' % i)) - 1 for i in (1, 2, 3, 4)}  # index of each stage's last block
        feat = {i: 'backbone.layer%d.%d/out' % (i, lastb[i]) for i in (1, 2, 3, 4)}
        dC = {k: self.buf('d' + k, *B[k].shape) for k in feat.values()}
        c2, c3, c4, c5 = (B[feat[i]] for i in (1, 2, 3, 4))
        dc2, dc3, dc4, dc5 = (dC[feat[i]] for i in (1, 2, 3, 4))
        # p2 = smooth_p2(p2pre), p2pre = up(p3) + r2
        dp2pre = self.buf('dp2pre', *B['p2pre'].shape)
        cbr_bwd('smooth_p2', fpn.smooth_p2, B['p2pre'], dP['smooth_p2'], dp2pre, False, consumer='reduce_conv_c2')
        self.up_bwd(dp2pre, dP['smooth_p3'], 0, True)
        cbr_bwd('reduce_conv_c2', fpn.reduce_conv_c2, c2, dp2pre, dc2, False)
        dp3pre = self.buf('dp3pre', *B['p3pre'].shape)
        cbr_bwd('smooth_p3', fpn.smooth_p3, B['p3pre'], dP['smooth_p3'], dp3pre, False, consumer='reduce_conv_c3')
        self.up_bwd(dp3pre, dP['smooth_p4'], 0, True)
        cbr_bwd('reduce_conv_c3', fpn.reduce_conv_c3, c3, dp3pre, dc3, False)
        dp4pre = self.buf('dp4pre', *B['p4pre'].shape)
        cbr_bwd('smooth_p4', fpn.smooth_p4, B['p4pre'], dP['smooth_p4'], dp4pre, False, consumer='reduce_conv_c4')
        self.up_bwd(dp4pre, dP['reduce_conv_c5'], 0, True)
        cbr_bwd('reduce_conv_c4', fpn.reduce_conv_c4, c4, dp4pre, dc4, False)
        cbr_bwd('reduce_conv_c5', fpn.reduce_conv_c5, c5, dP['reduce_conv_c5'], dc5, False)
        self.flush_wgrad_reduces()  # FPN + head (one grouped launch per gradient stage: train.GRAD_STAGES)
        if self.grad_ready_hook is not None:  # every FPN / head gradient kernel has been enqueued
            with self.side_stream():  # announced from the side stream (it has waited for the main one): main is not stalled
                self._side_waits_for_reductions()
                self.grad_ready_hook('segmentation')
        # backbone, deepest stage first; dC[...] already holds the FPN contribution
        bb = m.backbone
        dpool = self.buf('stem/dpool', *B['stem/pool'].shape)
        for li in (4, 3, 2, 1):
            layer = getattr(bb, 'layer%d' % li)
            for bi in range(lastb[li], -1, -1):
                name = 'backbone.layer%d.%d' % (li, bi)
                dout = self.bufs['d' + name + '/out']
                xin = B[name + '/in']
                if bi > 0:
                    dx, acc = self.buf('d' + 'backbone.layer%d.%d/out' % (li, bi - 1), *xin.shape), False
                elif li > 1:
                    dx, acc = dC[feat[li - 1]], True
                else:
                    dx, acc = dpool, False
                prev = 'backbone.layer%d.%d' % ((li, bi - 1) if bi > 0 else (li - 1, lastb[li - 1])) if (bi > 0 or li > 1) else None
                self._block_bwd(name, layer[bi], xin, dout, dx, acc, prev=prev)
            if li >= 3:
                self.flush_wgrad_reduces()
            if self.grad_ready_hook is not None and li >= 3:
                with self.side_stream():
                    self._side_waits_for_reductions()
                    self.grad_ready_hook('layer%d' % li)
        y0 = B['stem/y']
        if self.flush_before_pool:
            self._flush_wgrads()  # (late order: layer1's last weight gradient starts beside the max-pool backward, not behind it — round-5 trace: it had waited 218 us)
        if self.pool_argmax and self._pool_arg_generation == self.generation:
            # pool + ReLU + BatchNorm backward from the recorded argmax: sums over the pooled tensors, then ONE pass y -> dy
            dy0 = self.buf('stem/dy', *y0.shape)
            ws = self.scratch('_stem_pool_bn_ws', L.dbn_maxpool_bn_backward_ws_floats(N, y0.shape[1], y0.shape[2], 64))
            es = y0.element_size()
            self._prof_hbm('maxpool_bn_stats_kernel + maxpool_bn_bwd_apply_kernel',
                           (2 * y0.numel() + 3 * dpool.numel()) * es + 2 * dpool.numel())  # dpool, ypool, idx | y, dpool, idx -> dy
            check(L.dbn_maxpool_bn_backward_t(self.at, y0.data_ptr(), dpool.data_ptr(), B['stem/pool_idx'].data_ptr(), B['stem/pool_y'].data_ptr(),
                                              B['backbone.bn1/mean'].data_ptr(), B['backbone.bn1/rstd'].data_ptr(),
                                              self.views['backbone.bn1.weight'].data_ptr(), dy0.data_ptr(),
                                              self.grad_views['backbone.bn1.weight'].data_ptr(), self.grad_views['backbone.bn1.bias'].data_ptr(),
                                              N, y0.shape[1], y0.shape[2], 64, self.grad_scale, ws.data_ptr(), st), 'maxpool + bn backward')
            if self.prof:
                self.prof.end()
            self.conv_wgrad('backbone.conv1', dy0, B['x4w' if self.at != 0 else 'x4'], bb.conv1)
            self.flush_wgrad_reduces()
            self.join_side()
            self._dcn_send_back()
            self.saved_generation = -1
            self.backwards_since_clear += 1
            return
        dz = self.buf('stem/dz', *y0.shape)
        # the max-pool backward also emits the partial sums of the stem BatchNorm's backward (it has y and dz in registers)
        nparts = L.dbn_maxpool_bwd_parts(N, y0.shape[1], y0.shape[2], 64)
        parts = self.scratch('_stem_bn_parts', 2 * 64 * nparts)
        self._prof_hbm('bnrelu_maxpool_bwd_kernel', (2 * y0.numel() + 2 * dpool.numel()) * y0.element_size())  # y, pooled, dpool -> dz
        check(L.dbn_bnrelu_maxpool_bwd_t(self.at, y0.data_ptr(), B['backbone.bn1/scale'].data_ptr(), B['backbone.bn1/shift'].data_ptr(),
                                         B['stem/pool'].data_ptr(), dpool.data_ptr(), dz.data_ptr(), N, y0.shape[1], y0.shape[2], 64,
                                         B['backbone.bn1/mean'].data_ptr(), B['backbone.bn1/rstd'].data_ptr(), parts.data_ptr(), st),
              'maxpool bwd')
        if self.prof:
            self.prof.end()
        dy0 = self.bn_backward('backbone.bn1', y0, None, dz, 'stem/dy', sums=parts, sums_parts=nparts)
        self.conv_wgrad('backbone.conv1', dy0, B['x4w' if self.at != 0 else 'x4'], bb.conv1)
        self.flush_wgrad_reduces()
        self.join_side()
        self._dcn_send_back()
        self.saved_generation = -1
        self.backwards_since_clear += 1  # FusedAdam.step refuses gradients that a second backward pass overwrote

    backwards_since_clear = 0
    # the stem's max-pool with a recorded argmax (PyTorch's first-maximum rule; csrc/pointwise.hip: bnrelu_maxpool_fwd_arg_kernel).
    # DBN_POOL_ARGMAX=0: round 4's pair (gradient to every position that ties with the maximum, a gradient tensor at the conv's resolution)
    pool_argmax = os.environ.get('DBN_POOL_ARGMAX', '1') == '1'
    _pool_arg_generation = -1
    fpn_structured = True  # FPN output conv per upsample level (forward and backward) instead of over the concat
    fpn_one_launch = True  # forward: the four levels in one launch (dbn_pyramid_conv_f32) instead of four accumulating ones
    fpn_exact = False  # set by forward(): the levels are exact 1, 1/2, 1/4, 1/8 sizes, so the structured path applies

    def _fpn_combined_weights(self, name, conv, Cg, version=None):
        """Wd_g[ci][co][u][v] = sum of the 3x3 taps of W[co][64g+ci] that land on offset (u,v) of level g's (f+2)^2 footprint.
        version: stamp of a derived weight tensor (the eval-folded filters), as in pack()."""
        w = conv.weight
        Co = w.shape[0]
        if version is None:
            self._fpn_src = (name, conv, Cg)
        stamp = (w._version if version is None else version, self.param_epoch, w.data_ptr())
        ent = self.packs.get((name, 'combined'))
        if ent is None or ent[1] != stamp:
            wds = ent[0] if ent is not None else [device_empty((Cg, Co, (1 << g) + 2, (1 << g) + 2), w.device) for g in range(4)]
            for g in range(4):
                check(self.L.dbn_fpn_combine_weights(w.data_ptr(), Co, w.shape[1], g, Cg, wds[g].data_ptr(), self.stream),
                      'fpn_combine_weights')
            self.packs[(name, 'combined')] = (wds, stamp)
        return self.packs[(name, 'combined')][0], stamp[:2]

    def _fpn_conv_forward(self, name, conv, zs, out_name, bn_name, bn, train):
        N, H, W, Cg = zs[0].shape
        Co = conv.cout
        wds, wver = self._fpn_combined_weights(name, conv, Cg)
        y = self.buf(out_name, N, H, W, Co)
        fused = train and self.fuse_bn_stats
        sc = sh = None
        if self.fpn_one_launch and Co % 128 == 0 and Cg % 16 == 0:
            # all four levels in one launch: one accumulator per output tile, no read-modify-write of y
            wpk = [self.pack('%s#f%d' % (name, g), wds[g], 1, 1 << g, version=wver) for g in range(4)]
            # exact fp32: level 0 — a plain 3x3 conv of p2 with the first Cg input channels' filters, 53 % of the pyramid's FLOPs — goes
            # through the Winograd kernel (2.25x fewer matrix FLOPs) and the pyramid launch adds levels 1-3 onto it
            lv0 = _VirtualConv(Cg, Co, 3, 1, 1, wds[0], conv.bias)
            first = int(self.fpn_level0_winograd and not self._use_planes and wds[0].shape[0] == Cg and self._winograd_ok(zs[0], lv0))
            # 16-bit storage (round 5): level 0 — 53 % of the pyramid's FLOPs — as a mode-1 3x3 / stride-1 launch on the level-0 panel, which
            # takes the pixel-patch kernel (0.3-0.36 of the 16-bit matrix peak; the pyramid's generic loop: 0.2), levels 1-3 added onto it
            first16 = self._fpn_level0_patch_ok(zs[0], Co, train)
            if first16:
                if self.prof:
                    self._prof_igemm(N * H * W, Co, 2.0 * N * H * W * Co * Cg * 9, 'fwd %s level 0' % name, 1, (N, H, W, Cg, H, W, 3, 1, 1))
                self._igemm('igemm fpn level 0', zs[0].data_ptr(), wpk[0].data_ptr(), _p(conv.bias), y.data_ptr(), N, H, W, Cg, H, W, Co, 3, 3, 1, 1, 1,
                            0, 0)
                if self.prof:
                    self.prof.end()
                first = 1
            elif first:
                # (wds[0][ci][co][u][v] = W[co][ci][2-u][2-v]: its data-gradient panel is the forward conv of W's first Cg input channels)
                up = self._winograd_panel(name + '#lv0', wds[0], Cg, dgrad=1, version=wver)
                if self.prof:
                    self.prof.begin('winograd_f32_kernel', 2.0 * N * H * W * Co * Cg * 4, 4.0 * N * H * W * (Co + Cg), 'fwd %s level 0' % name)
                check(self.L.dbn_winograd_conv_bn_f32(zs[0].data_ptr(), up.data_ptr(), _p(conv.bias), y.data_ptr(), N, H, W, Cg, Co, None, None,
                                                      0.0, 0.0, None, None, None, None, None, None, None, self.stream), 'winograd ' + name)
                if self.prof:
                    self.prof.end()
            flops = sum(2.0 * N * z.shape[1] * z.shape[2] * Cg * Co * ((1 << g) + 2)**2 for g, z in enumerate(zs) if g >= first)
            if self.prof:
                bn_tile = 256 if (not self._use_planes and self.L.dbn_pyramid_wide_would_run(self.at, N, H, W, Cg, Co)) else 128
                waves = '1,4' if (not self._use_planes and self.at in (1, 2) and self.ns == 1) else '2,2'
                self.prof.begin('igemm_f32_kernel<128,%d,%s,3,%d,%d,false,0,true>' % (bn_tile, waves, self.ns, 3 if self._use_planes else self.at), flops, 0.0,
                                'fwd %s (pyramid)' % name)
            if fused:
                C = Co
                sc, sh = self.fbuf(bn_name + '/scale', C), self.fbuf(bn_name + '/shift', C)
                mu, rs = self.fbuf(bn_name + '/mean', C), self.fbuf(bn_name + '/rstd', C)
                ws = self.scratch('_conv_bn_ws', self.L.dbn_pyramid_conv_ws_floats(N, H, W, Co))
                bnargs = (bn.weight.data_ptr(), bn.bias.data_ptr(), bn.eps, bn.momentum, bn.running_mean.data_ptr(),
                          bn.running_var.data_ptr(), sc.data_ptr(), sh.data_ptr(), mu.data_ptr(), rs.data_ptr(), ws.data_ptr())
                self.nbt_pending[bn_name] = self.nbt_pending.get(bn_name, 0) + 1
            else:
                bnargs = (None, None, 0.0, 0.0) + (None, ) * 7
            pat, zp = self.at, [z.data_ptr() for z in zs]
            if self._use_planes and Cg % 16 == 0:
                pat, zp = 3, [self._planes(z).data_ptr() for z in zs]
            check(self.L.dbn_pyramid_conv_from_t(first, pat, *zp, *[w_.data_ptr() for w_ in wpk], _p(conv.bias), y.data_ptr(),
                                                   N, H, W, Cg, Co, 0, self.ns, *bnargs, self.stream), 'pyramid_conv')
            if self.prof:
                self.prof.end()
            if not fused:
                sc, sh = self.bn_coef(bn_name, bn, y, train)
            return y, sc, sh
        for g, z in enumerate(zs):
            f, k = 1 << g, (1 << g) + 2
            Hg, Wg = z.shape[1], z.shape[2]
            wpk = self.pack('%s#f%d' % (name, g), wds[g], 1, f, version=wver)
            if self.prof:
                self._prof_igemm(N * H * W, Co, 2.0 * N * Hg * Wg * Cg * Co * k * k, 'fwd %s level %d' % (name, g), 2 if g else 1)
            args = (z.data_ptr(), wpk.data_ptr(), _p(conv.bias) if g == 0 else None, y.data_ptr(), N, Hg, Wg, Cg, H, W, Co, k, k, f, 1, 1)
            if g == 3 and fused:
                sc, sh = self._conv_bn_call('fpn conv+bn level 3', bn_name, bn, y, args, 1, f, accumulate=1)
            else:
                self._igemm('igemm fpn fwd', *args, int(g > 0), 0)
            if self.prof:
                self.prof.end()
        if not fused:
            sc, sh = self.bn_coef(bn_name, bn, y, train)
        return y, sc, sh

    def _fpn_conv_backward(self, name, conv, dy, levels, zs, dP):
        N, H, W, Co = dy.shape
        Cg = zs[0].shape[3]
        wds, wver = self._fpn_combined_weights(name, conv, Cg)
        ts = []
        for g, nm in enumerate(levels):
            f, k = 1 << g, (1 << g) + 2
            z = zs[g]
            Hg, Wg = z.shape[1], z.shape[2]
            d = self.buf(nm + '/dz', *z.shape)
            wpk = self.pack('%s#g%d' % (name, g), wds[g], 0, f, version=wver)
            flops = 2.0 * N * Hg * Wg * Cg * Co * k * k
            args = (dy.data_ptr(), wpk.data_ptr(), None, d.data_ptr(), N, H, W, Co, Hg, Wg, Cg, k, k, f, 1, 0, 0, 0)
            # level 0 (p2) receives nothing else: its gradient is final here and carries the sums of smooth_p2's BatchNorm; the
            # coarser levels are completed by the nearest-upsample adjoints later
            bn_ = 'segmentation_body.%s.bn' % nm
            consumer = (bn_, self.bufs[nm + '/y'], None) if (g == 0 and self._bnb_eligible(args)) else None
            if (g == 0 and self.winograd and self.ns == 0 and self.at == 0 and dy.dtype == torch.float32
                    and bool(self.L.dbn_winograd_eligible(N, H, W, Co, Cg))):
                # level 0 is a plain 3x3 / stride-1 / pad-1 conv of dy with the (combined == original) weights [Cg][Co][3][3]
                up = self._winograd_panel('%s#g0' % name, wds[0], Co, dgrad=0, version=wver)
                self._winograd_dgrad('%s level 0' % name, dy, None, d, False, consumer, panel=up)
            else:
                if self.prof:
                    self._prof_igemm(N * Hg * Wg, Cg, flops, 'dgrad %s level %d' % (name, g), 0, epi=int(consumer is not None))
                self._igemm('igemm fpn dgrad', *args, consumer=consumer)
                if self.prof:
                    self.prof.end()
            dP[nm] = d
            t = self.fbuf('%s#t%d' % (name, g), Cg, Co, k, k)
            self._presplit(z, dy)
            with self.side_stream():
                self.wgrad('%s level %d' % (name, g), z, dy, Cg, Co, k, f, 1, t)
            ts.append(t)
        with self.side_stream():
            check(self.L.dbn_fpn_scatter_wgrad(ts[0].data_ptr(), ts[1].data_ptr(), ts[2].data_ptr(), ts[3].data_ptr(), Co, Cg,
                                               self.grad_views[name + '.weight'].data_ptr(), self.stream), 'fpn_scatter_wgrad')
            if conv.bias is not None and name + '.bias' not in self._bias_done:
                self.col_sum(dy, self.grad_views[name + '.bias'])

    def _block_bwd(self, name, blk, xin, dout, dx, dx_acc, prev=None):
        """prev: name of the block whose output is this block's input (None: the stem's pooled map) — its last BatchNorm
        consumes dx with the ReLU mask of its output; conv1's data gradient writes dx last and carries that BatchNorm-backward's
        sums in its epilogue."""
        B = self.bufs
        out = B[name + '/out']
        has_down = blk.downsample is not None
        last = '3' if hasattr(blk, 'conv3') else '2'  # Bottleneck: the residual joins after bn3
        ylast = B[name + '/y' + last]
        if has_down:
            dyl = self.bn_backward(name + '.bn' + last, ylast, out, dout, name + '/dy' + last)
        else:
            # identity shortcut: the ReLU-masked gradient goes straight to the block input
            dyl = self.bn_backward(name + '.bn' + last, ylast, out, dout, name + '/dy' + last, gout=dx, gout_acc=dx_acc)
            dx_acc = True
        if last == '3':
            z2 = B[name + '/z2']
            self.conv_wgrad(name + '.conv3', dyl, z2, blk.conv3)
            dz2 = self.buf(name + '/dz2', *z2.shape)
            self.conv_dgrad(name + '.conv3', dyl, blk.conv3, dz2, False, consumer=(name + '.bn2', B[name + '/y2'], None))
            dy2 = self.bn_backward(name + '.bn2', B[name + '/y2'], 'self', dz2, name + '/dy2')
        else:
            dy2 = dyl
        z1 = B.get(name + '/z1')
        if z1 is None:  # apply-on-load: conv2 read y1 through bn1's coefficients (see _block_fwd)
            y1 = B[name + '/y1']
            dz1 = self.buf(name + '/dz1', *y1.shape)
            self.conv_wgrad(name + '.conv2', dy2, y1, blk.conv2, x_act=(B[name + '.bn1/scale'], B[name + '.bn1/shift']))
            self.conv_dgrad(name + '.conv2', dy2, blk.conv2, dz1, False, consumer=(name + '.bn1', y1, None))
        else:
            dz1 = self.buf(name + '/dz1', *z1.shape)
            self._conv2_bwd(name, blk, dy2, z1, dz1)
        dy1 = self.bn_backward(name + '.bn1', B[name + '/y1'], 'self', dz1, name + '/dy1')
        self.conv_wgrad(name + '.conv1', dy1, xin, blk.conv1)
        if has_down:
            # the shortcut's gradient first: conv1's data gradient then is the last writer of dx
            dyd = self.bn_backward(name + '.downsample.1', B[name + '/yd'], out, dout, name + '/dyd')
            self.conv_wgrad(name + '.downsample.0', dyd, xin, blk.downsample[0])
            self.conv_dgrad(name + '.downsample.0', dyd, blk.downsample[0], dx, dx_acc)
            dx_acc = True
        consumer = None
        if prev is not None:
            plast = '3' if (prev + '/y3') in B else '2'
            consumer = (prev + '.bn' + plast, B[prev + '/y' + plast], B[prev + '/out'])
            if (prev + '/yd') in B:  # the projection shortcut's BatchNorm consumes the same gradient under the same mask
                consumer += ((prev + '.downsample.1', B[prev + '/yd']), )
        self.conv_dgrad(name + '.conv1', dy1, blk.conv1, dx, dx_acc, consumer=consumer)
