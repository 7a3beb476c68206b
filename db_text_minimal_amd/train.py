"""The per-step training loop of /root/reference/src/train.py:155-172, MI355X-native.

    preds = dbnet(batch['img'])                      # train.py:160
    _batch = stack(prob_map, supervision_mask, thresh_map, text_area_map)   # :163-166
    losses = criterion(preds, _batch)                # :167-168
    optimizer.zero_grad(); total.backward(); optimizer.step()               # :169-172

`DBTrainer.step` issues exactly that sequence as HIP kernels on one stream, with
no autograd graph, no host synchronisation and (for world_size > 1) ONE sum
all-reduce of the flat gradient buffer over RCCL/xGMI per step (north_star's single
collective; optionally the same sum as four contiguous buckets issued in the order the
backward pass completes them — `overlap_allreduce`); the 1/world average is folded
into the Adam kernel.  BatchNorm statistics
and the loss normalisers stay per GPU (standard data-parallel semantics).
"""
import os

import torch
import torch.distributed as dist

from . import _lib
from . import engine as engine_mod
from ._lib import check
from .optim import FusedAdam

GT_KEYS = ('prob_map', 'supervision_mask', 'thresh_map', 'text_area_map')  # train.py:163-166 order


def stack_gts(batch):
    return torch.stack([batch[k] for k in GT_KEYS])


def allreduce_flat_grads(flat_grad, world, group=None):
    """THE collective of a data-parallel step: one sum all-reduce over the flat fp32 gradient
    buffer (12 269 378 elements = 49 MB for ResNet18-FPN-DBHead).  Returns the factor the
    optimizer applies to turn the sum into the mean (folded into the Adam kernel)."""
    if world <= 1 and not (dist.is_available() and dist.is_initialized()):
        return 1.0
    dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=group)
    return 1.0 / world


def replica_divergence(flat, group=None):
    """((sum, sum of squares) of `flat` in fp64, max over both of |max over ranks - min over ranks|).  The second value is
    0.0 exactly when every replica holds the same buffer — data-parallel replicas must (same start, same averaged gradient,
    same Adam state).  Two tiny all-reduces; bench.py fails an N > 1 run on divergence."""
    f = flat.double()
    mine = torch.stack([f.sum(), (f * f).sum()])
    if not (dist.is_available() and dist.is_initialized()):
        return mine.tolist(), 0.0
    hi, lo = mine.clone(), mine.clone()
    dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=group)
    dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=group)
    return mine.tolist(), float((hi - lo).abs().max())


# backward completes the gradient buffer back to front: FPN+head first, then backbone stages 4..1, stem last
GRAD_STAGES = ('segmentation', 'layer4', 'layer3', 'rest')


def bucket_ranges(names, offsets, total):
    """[lo, hi) float ranges of the flat gradient buffer per GRAD_STAGES entry.  `names`/`offsets`: the live parameters in
    flat-buffer order (named_parameters order: backbone stem, layer1..4, FPN, head).  The ranges are contiguous and
    cover [0, total) exactly once."""
    def first(prefix):
        for n, off in zip(names, offsets):
            if n.startswith(prefix):
                return off
        raise KeyError(prefix)
    l3, l4, seg = first('backbone.layer3.'), first('backbone.layer4.'), first('segmentation_body.')
    assert 0 < l3 < l4 < seg < total
    return {'segmentation': (seg, total), 'layer4': (l4, seg), 'layer3': (l3, l4), 'rest': (0, l3)}


class BucketedAllReduce:
    """The data-parallel exchange of one step: the same sum as one all-reduce over the flat gradient buffer, issued per
    bucket as soon as the backward pass has enqueued that bucket's last gradient kernel (`ready`), asynchronously on the
    process group's stream, and joined before the optimizer (`finish`)."""

    def __init__(self, flat_grad, ranges, world, group=None):
        self.flat, self.ranges, self.world, self.group = flat_grad, ranges, world, group
        self.pending, self.done = [], set()

    def ready(self, stage):
        lo, hi = self.ranges[stage]
        assert stage not in self.done
        self.done.add(stage)
        self.pending.append(dist.all_reduce(self.flat[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def finish(self):
        """Issues whatever has not been announced (at least 'rest'), waits for all buckets; returns the mean factor."""
        for stage in GRAD_STAGES:
            if stage not in self.done:
                self.ready(stage)
        for w in self.pending:
            w.wait()
        self.pending, self.done = [], set()
        return 1.0 / self.world


class DBTrainer:
    def __init__(self, model, criterion, optimizer=None, process_group=None, lr=0.005, distributed=None, gc_freeze=None):
        """`distributed`: None = data parallel exactly when a process group is initialised (the default group or
        `process_group`); False = this trainer never touches a process group, whatever the process has initialised (a
        rank-local side computation beside a data-parallel job: bench.py's parity gate — a trainer that joined the default
        group there would pair its start-up broadcasts with the other ranks' gradient all-reduce).
        `gc_freeze`: see freeze_heap(); None reads DBN_GC_FREEZE (default off: a process-wide side effect is the caller's
        decision — fit() and bench.py take it explicitly)."""
        self.model = model
        self.criterion = criterion
        self.optimizer = optimizer if optimizer is not None else FusedAdam(model, lr=lr)
        self.pg = process_group
        have_group = dist.is_available() and dist.is_initialized()
        self.distributed = have_group if distributed is None else bool(distributed)
        if self.distributed and not have_group:
            raise RuntimeError('DBTrainer(distributed=True) needs an initialised process group (train.init_distributed)')
        self.world = dist.get_world_size(process_group) if self.distributed else 1
        self._gone = None
        # The data-parallel exchange: ONE sum all-reduce of the flat gradient buffer after the backward pass — the step of
        # BASELINE's north_star (reference train.py:169-172 is the sequence being parallelised).  overlap_allreduce = True
        # (DBN_OVERLAP_ALLREDUCE=1, bench.py --bucketed-allreduce) issues the same sum as four contiguous buckets under the
        # backward pass instead (bit-identical results, tests/test_dp_gloo.py, tests/test_rccl_gpu.py).
        self.overlap_allreduce = os.environ.get('DBN_OVERLAP_ALLREDUCE', '0') == '1'
        # Replicas start from rank 0's state.  The broadcast is the first collective, and the first collective creates the
        # RCCL communicator — which should happen AFTER the activation arena exists (init_distributed: creating it first
        # cost 2.4 ms/step at bs16 640^2).  So the sync is deferred to the first step(), which knows the batch shape: it
        # takes one dry forward+backward to allocate every buffer, restores the BatchNorm buffers, then broadcasts.
        self._need_sync = self.distributed  # (also with one forced rank: tests/dist_child.py)
        self.exchange_events = None  # set to [] to collect (issue, done) HIP event pairs around the exchange (bench.py)
        # hipGraph: forward + DBLoss + backward of the steady-state step as ONE graph launch (the arena is static, so the ~340
        # kernel launches of a step replay with their captured arguments; the two-stream fork / join is captured with them).
        # The gradient exchange and Adam stay outside the graph (bias corrections and the learning rate change every step).
        # Captured after `graph_warmup` eager steps on the same batch shape; bit-identical to the eager step (tested).
        # OFF by default: on ROCm 7.2 / MI355X the replay is SLOWER than the eager launches (f32 33.4 vs 32.0 ms, bf16 10.76 vs
        # 10.07 ms per step at bs16 640^2): the host already enqueues far ahead of the GPU, so there is no launch latency to
        # remove, and the graph executor leaves larger gaps between dependent nodes than back-to-back stream launches do.
        self.use_graph = os.environ.get('DBN_STEP_GRAPH', '0') == '1'
        self.graph_warmup = 2
        self._graph = None
        # The cyclic garbage collector and the step (round 5: the root cause of the "bimodal" 16-bit step of DESIGN section 9 row 3 /
        # section 12.4 — tools/bimodal_probe.py).  A step enqueues ~300 launches in ~4 ms of host time and creates a few thousand short-lived
        # Python objects; every few steps that trips a generation-2 collection, which walks the WHOLE heap of the process (torch, the
        # model's module tree, the engine's tables: ~90-100 ms) while the GPU runs dry — one such stall inside twenty 10 ms bf16 steps
        # is the difference between 1670 and 1280 images/s, two are 1100.  Nothing in a step creates reference cycles.  After the
        # third step — every buffer, panel and table exists by then — freeze_heap() collects once and FREEZES the heap (gc.freeze():
        # existing objects leave the collector's generations), so later passes scan only what was created since: < 1 ms.
        # That is a PROCESS-WIDE side effect (the caller's data loaders and loggers are frozen with everything else), so it is
        # opt-in: DBTrainer(gc_freeze=True) / DBN_GC_FREEZE=1 do it at the fourth step, fit() asks for it and undoes it on return
        # (unfreeze_heap()), a caller with its own loop calls freeze_heap() when its set-up is complete.
        self.gc_freeze = (os.environ.get('DBN_GC_FREEZE', '0') == '1') if gc_freeze is None else bool(gc_freeze)
        self._froze = False
        self._steps = 0

    def freeze_heap(self):
        """gc.collect() + gc.freeze(): every object alive in this PROCESS now leaves the cyclic collector's generations, so the
        generation-2 passes a step's few thousand temporaries trip every few steps stop walking torch, the module tree and
        the engine's tables (~90 ms each with the GPU running dry: 9.5 vs 12.4 ms per bf16 step, tools/bimodal_probe.py).
        Frozen objects are still freed by reference counting; cyclic garbage among them is not reclaimed until
        unfreeze_heap().  Call it once the steady state is reached (after the first steps of a batch shape)."""
        import gc
        gc.collect()
        gc.freeze()
        self._froze = True

    def unfreeze_heap(self):
        """Undo freeze_heap(): the frozen objects return to the oldest generation (gc.unfreeze())."""
        if self._froze:
            import gc
            gc.unfreeze()
            self._froze = False

    def sync_from_rank0(self):
        """Data-parallel replicas must start from identical state: rank 0's flat parameter buffer, its BatchNorm buffers and
        (when it exists) the Adam state are broadcast once (what DistributedDataParallel does at construction).  During
        training the parameters stay identical by construction (same averaged gradient, same Adam state); BatchNorm running
        statistics are NOT synchronised afterwards — each rank tracks its own shard, and checkpoints carry rank 0's
        (SURVEY.md §8e)."""
        eng = self.model.engine
        eng.ensure_flat()
        eng.flush_counters()
        src = dist.get_global_rank(self.pg, 0) if self.pg is not None else 0
        dist.broadcast(eng.flat, src=src, group=self.pg)
        for b in self.model.buffers():
            dist.broadcast(b, src=src, group=self.pg)
        opt = self.optimizer
        if isinstance(opt, FusedAdam):
            opt._ensure_state()
            cnt = torch.tensor([opt.step_count], device=eng.flat.device, dtype=torch.int64)
            dist.broadcast(cnt, src=src, group=self.pg)
            opt.step_count = int(cnt.item())
            dist.broadcast(opt.exp_avg, src=src, group=self.pg)
            dist.broadcast(opt.exp_avg_sq, src=src, group=self.pg)
        eng.mark_params_dirty()
        self._need_sync = False

    def _warm_arena_then_sync(self, img, gts):
        """One dry forward + loss + backward so that every activation / gradient / scratch buffer of this batch shape exists,
        with the BatchNorm buffers put back afterwards (the dry pass must leave no trace), THEN the start-up broadcast."""
        model, eng = self.model, self.model.engine
        eng.ensure_flat()
        eng.flush_counters()
        saved = [b.detach().clone() for b in model.buffers()]
        preds = eng.forward(img, train=True)
        _, dpreds = self._loss(preds, gts)
        eng.backward(dpreds)
        if isinstance(self.optimizer, FusedAdam):
            self.optimizer._ensure_state()
        eng.nbt_pending = {}
        with torch.no_grad():
            for b, s in zip(model.buffers(), saved):
                b.copy_(s)
        self.sync_from_rank0()

    def param_checksum(self):
        """See replica_divergence(): ((sum, sum of squares) of this rank's parameters, spread over the ranks)."""
        if not self.distributed:
            f = self.model.engine.flat.double()
            return [float(f.sum()), float((f * f).sum())], 0.0
        return replica_divergence(self.model.engine.flat, self.pg)

    def _loss(self, preds, gts):
        """dbn_db_loss_fwd + _bwd with d(total)=1; returns (losses[5], dpreds)."""
        L = _lib.lib()
        c = self.criterion
        N, C, H, W = preds.shape
        dev = preds.device
        st = torch.cuda.current_stream(dev).cuda_stream
        reduction = getattr(c, 'reduction', 'mean')
        per_pixel = reduction == 'none'
        frac = bool(getattr(c, 'fractional_maps', False)) and not per_pixel  # literal top-k form of the scalar-BCE reductions
        guard = getattr(c, '_guard', None)
        if guard is not None:
            guard.check()  # (a count that has landed from an earlier step; never waits)
        if self._gone is None or self._gone.device != dev:
            self._gone = torch.tensor([0., 0., 0., 0., 1.], device=dev)
            self._coef = torch.zeros(8, device=dev)
            self._ws = None
        need = (L.dbn_db_loss_ohem_ws_bytes(N, H, W) if (per_pixel or frac) else L.dbn_db_loss_ws_bytes()) // 4 + 1
        if self._ws is None or self._ws.numel() < need:
            self._ws = engine_mod.device_empty(need, dev)
        losses = engine_mod.device_empty(5, dev)
        fwd = L.dbn_db_loss_ohem_fwd if per_pixel else (L.dbn_db_loss_sum_fwd if reduction == 'sum' else L.dbn_db_loss_fwd)
        eng = self.model.engine
        if eng.prof:  # reads the 3 maps and the 4 targets once
            eng.prof.begin('db_loss_fwd_kernel', 0.0, 4.0 * (preds.numel() + gts.numel()))
        if frac:
            check(L.dbn_db_loss_frac_fwd(preds.data_ptr(), gts.data_ptr(), N, H, W, C, c.alpha, c.beta, float(c.negative_ratio), c.eps,
                                         1 if reduction == 'sum' else 0, losses.data_ptr(), self._coef.data_ptr(), self._ws.data_ptr(), st),
                  'db_loss_frac_fwd')
        else:
            check(fwd(preds.data_ptr(), gts.data_ptr(), N, H, W, C, c.alpha, c.beta, float(c.negative_ratio), c.eps, losses.data_ptr(),
                      self._coef.data_ptr(), self._ws.data_ptr(), st), 'db_loss_fwd')
            if guard is not None and not per_pixel:
                guard.watch(self._coef)
        if eng.prof:
            eng.prof.end()
            eng.prof.begin('db_loss_bwd_kernel', 0.0, 4.0 * (2 * preds.numel() + gts.numel()))  # + writes the 3 map gradients
        dpreds = self.model.engine.fbuf('dpreds', N, C, H, W)
        if per_pixel:
            check(L.dbn_db_loss_ohem_bwd(preds.data_ptr(), gts.data_ptr(), self._coef.data_ptr(), self._gone.data_ptr(),
                                         self._ws.data_ptr(), c.alpha, c.beta, N, H, W, C, dpreds.data_ptr(), st), 'db_loss_bwd')
        else:
            check(L.dbn_db_loss_bwd(preds.data_ptr(), gts.data_ptr(), self._coef.data_ptr(), self._gone.data_ptr(), c.alpha, c.beta,
                                    N, H, W, C, dpreds.data_ptr(), st), 'db_loss_bwd')
        if eng.prof:
            eng.prof.end()
        return losses, dpreds

    def step(self, img, gts=None, resident=False):
        """One training iteration.  img [N,3,H,W], gts [4,N,H,W] (or the reference's batch dict
        as `img`, with gts=None).  Returns (preds, losses[5]) as device tensors; no host sync.
        `resident=True` (hipGraph step only) is the caller's promise that `img` / `gts` are the very tensors of the previous
        call, unmodified (bench.py's synthetic batch): the copy into the graph's static inputs is skipped.  Nothing is
        inferred from object identity — a new batch is always copied."""
        if isinstance(img, dict):
            batch = img
            img, gts = batch['img'], stack_gts(batch)
        model, eng = self.model, self.model.engine
        if not model.training:
            raise RuntimeError('DBTrainer.step requires model.train()')
        self._steps += 1
        if self.gc_freeze and self._steps == 4:
            self.freeze_heap()
        gts = gts.contiguous().float()
        if self._need_sync:
            self._warm_arena_then_sync(img, gts)
        distributed = self.distributed
        if self.use_graph and eng.prof is None and not (distributed and self.overlap_allreduce):
            self.optimizer.zero_grad()  # (before the backward pass, as on the eager path: the replay below counts as ONE pass)
            got = self._graph_step(img, gts, resident)
            if got is not None:
                preds, losses = got
                ev = self._exchange_event()
                scale = allreduce_flat_grads(eng.flat_grad, self.world, self.pg) if distributed else 1.0
                self._exchange_event(ev)
                self.optimizer.step(grad_scale=scale)
                return preds, losses
        preds = eng.forward(img, train=True)
        assert preds.size(1) == 3  # train.py:161
        losses, dpreds = self._loss(preds, gts)
        self.optimizer.zero_grad()
        if distributed and self.overlap_allreduce:
            names = [n for n, _ in eng.live_params]
            ex = BucketedAllReduce(eng.flat_grad, bucket_ranges(names, eng.offsets, eng.flat_grad.numel()), self.world, self.pg)
            eng.grad_ready_hook = ex.ready
            try:
                eng.backward(dpreds)
            finally:
                eng.grad_ready_hook = None
            ev = self._exchange_event()  # backward fully enqueued: what the exchange still takes from here on is exposed
            scale = ex.finish()
            self._exchange_event(ev)
        else:
            eng.backward(dpreds)
            ev = self._exchange_event()
            scale = allreduce_flat_grads(eng.flat_grad, self.world, self.pg) if distributed else 1.0
            self._exchange_event(ev)
        self.optimizer.step(grad_scale=scale)
        return preds, losses

    def _graph_step(self, img, gts, resident=False):
        """forward + loss + backward through a captured hipGraph; None while still warming up (the caller takes the eager path).
        The graph reads the batch from its own static buffers: every batch is copied in first (183 MB at bs16 640^2, ~0.1 ms)
        unless the caller declares it `resident` (see step()).  Round 3 inferred "same batch" from (id, _version, data_ptr) of the
        arguments: a freed temporary's object slot and allocator block are reused by the next batch's tensor, so that key
        collided and the graph replayed on the PREVIOUS batch (advisor finding) — identity is no longer consulted.
        `preds` is the graph's static output tensor (overwritten by the next step, like the engine's arena in the eager step);
        the five losses are returned as a fresh 5-float copy, so a caller may keep summing them across steps (fit())."""
        eng = self.model.engine
        img = img.contiguous().float()
        key = (tuple(img.shape), tuple(gts.shape), eng.math_mode, eng.overlap_wgrad, eng.fuse_bn_bwd_sums)
        G = self._graph
        if G is None or G['key'] != key:
            self._graph = G = {'key': key, 'seen': 0, 'graph': None}
        if G['graph'] is None:
            G['seen'] += 1
            if G['seen'] <= self.graph_warmup:  # eager steps first: every buffer, weight panel and side stream exists afterwards
                return None
            dev = img.device
            G['img'], G['gts'] = img.clone(), gts.clone()  # the graph's static inputs
            G['src'] = None
            torch.cuda.synchronize(dev)
            nbt0, gen0 = dict(eng.nbt_pending), eng.generation
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                preds = eng.forward(G['img'], train=True)
                losses, dpreds = self._loss(preds, G['gts'])
                eng.backward(dpreds)
            # capture ran the host side of one step without executing it: undo its bookkeeping, replay() redoes it per launch
            G['nbt_delta'] = {k: v - nbt0.get(k, 0) for k, v in eng.nbt_pending.items() if v != nbt0.get(k, 0)}
            eng.nbt_pending, eng.generation = nbt0, gen0
            eng.backwards_since_clear -= 1  # (the captured backward() counted itself without running: the replay below is the pass)
            G['graph'], G['preds'], G['losses'] = g, preds, losses
        if not (resident and G['src']):
            G['img'].copy_(img)
            G['gts'].copy_(gts)
            G['src'] = True
        G['graph'].replay()
        eng.generation += 1
        for k, v in G['nbt_delta'].items():
            eng.nbt_pending[k] = eng.nbt_pending.get(k, 0) + v
        eng.saved_generation = -1  # (the graph contains the backward pass: the saved activations are consumed)
        eng.backwards_since_clear += 1
        return G['preds'], G['losses'].clone()

    def _exchange_event(self, start=None):
        """HIP events on the main stream before / after the gradient exchange (when `exchange_events` is a list): the elapsed
        time between them is the part of the collective(s) the step waits for — everything not hidden under backward kernels."""
        if self.exchange_events is None:
            return None
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        if start is not None:
            self.exchange_events.append((start, e))
        return e


def init_distributed(backend=None):
    """One process per GPU, launched by torch.distributed.run; backend nccl == RCCL on ROCm (the default).
    `backend='gloo'` (or DBN_DIST_BACKEND=gloo) moves the same device tensors through gloo instead — for exercising the
    N > 1 control path (start-up broadcasts, gradient all-reduce, checksums) with several ranks on ONE GPU, which RCCL
    refuses (tests/dist_child2.py); DBN_DIST_ONE_DEVICE=1 then maps every local rank to device 0.
    Returns (rank, local device index, world)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if os.environ.get('DBN_DIST_ONE_DEVICE', '0') == '1':
        local = 0
    backend = backend or os.environ.get('DBN_DIST_BACKEND', 'nccl')
    force = os.environ.get('DBN_FORCE_DIST', '0') == '1'  # exercise the RCCL path even with one rank
    if (world > 1 or force) and not dist.is_initialized():
        torch.cuda.set_device(local)
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29511')
        # No device_id: the RCCL communicator is then created lazily by the first collective, i.e. after the model's arena
        # exists.  Creating it eagerly BEFORE the activations are allocated costs 2.4 ms/step (6 %) at bs16 640^2 on MI355X
        # (tools/dist_probe2.py: 37.6 vs 35.2 ms) — RCCL's buffers come first and the arena lands in a slower placement.
        dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, local, world


# ----------------------------------------------------------------------------------------------------------------------
# Epoch-level loop: reference src/train.py:146-318 without its Hydra / TensorBoard / OpenCV-metric plumbing.
# ----------------------------------------------------------------------------------------------------------------------
def _mean_over_ranks(values, group, device):
    """Rank-uniform floats: the mean over the ranks of each value (one small all-reduce).  Every decision of the epoch loop
    that leads to a collective (checkpoint barrier) or changes the optimizer (plateau scheduler) must be taken from these —
    the per-rank losses differ (each rank sees its own shard and tracks its own BatchNorm running statistics), and ranks
    that disagree about saving would enter different collectives."""
    if not (dist.is_available() and dist.is_initialized()):
        return [float(v) for v in values]
    t = torch.tensor([float(v) for v in values], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return (t / dist.get_world_size(group)).tolist()


def _collective_device(model):
    """Device for the small control all-reduces: the model's (RCCL needs device tensors, gloo takes host ones)."""
    backend = dist.get_backend() if (dist.is_available() and dist.is_initialized()) else None
    if backend == 'nccl':
        return next(model.parameters()).device
    return torch.device('cpu')


def evaluate(model, criterion, loader, thresh=0.3, device=None, pixel_metric=True):
    """train.py:228-262: eval-mode forward under no_grad, DBLoss on the 2-channel output (single value), the pixel metric
    on device.  Returns (mean test loss as a float, score dict of the RunningScore over the whole loader)."""
    was_training = model.training
    model.eval()
    running = None
    if pixel_metric:
        from .text_metrics import RunningScore
        running = RunningScore(2)
    total, n = None, 0
    with torch.no_grad():
        for batch in loader:
            if device is not None:
                batch = {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in batch.items()}
            preds = model(batch['img'])
            assert preds.size(1) == 2  # train.py:241
            loss = criterion(preds, stack_gts(batch))
            total = loss if total is None else total + loss
            n += 1
            if running is not None:
                running.update_device(preds[:, 0, :, :], batch['prob_map'], batch['supervision_mask'], thresh)  # no host sync
    model.train(was_training)
    score = running.get_scores()[0] if (n and running is not None) else {}
    return (float(total) / max(n, 1) if total is not None else float('nan')), score


def fit(model, criterion, optimizer, train_loader, test_loader=None, epochs=1, scheduler=None, lrs_mode=None, thresh=0.3,
        best_cp_path=None, last_cp_path=None, device=None, log=None, process_group=None, trainer=None, pixel_metric=True):
    """The reference's training driver (train.py:146-318): per epoch a pass over `train_loader` through DBTrainer.step
    (lr scheduler stepped per iteration when lrs_mode == 'poly', train.py:172-173), the running pixel metric
    (train.py:175-181, on device), then evaluate() on `test_loader`, the reference's best-checkpoint rule
    (`test_loss <= best_test_loss and train_loss <= best_train_loss`, train.py:301-305; `train_loss` is the epoch SUM as
    there), ReduceLROnPlateau-style schedulers stepped with the test loss when lrs_mode == 'reduce' (train.py:307-308), and
    the final state_dict at `last_cp_path` (train.py:316).  Box-level P/R/HMean (train.py:277-299) needs the host OpenCV
    post-processing and is left to the caller.  Returns a list of per-epoch dicts.

    Data parallel: each rank iterates its own shard of the loaders.  The epoch's train-loss sum and the test loss are
    AVERAGED OVER THE RANKS before they are compared, recorded or given to the scheduler, so every rank takes the same
    save / no-save decision (the checkpoint barrier is reached by all or none) and the same learning-rate schedule.
    `trainer`: a ready DBTrainer (default: built here); `pixel_metric=False` skips the device pixel metric."""
    own_trainer = trainer is None
    if own_trainer:
        # (the heap freeze of DBTrainer.freeze_heap is taken for the duration of this call and undone on return)
        trainer = DBTrainer(model, criterion, optimizer, process_group=process_group,
                            gc_freeze=os.environ.get('DBN_GC_FREEZE', '1') == '1')
    distributed = dist.is_available() and dist.is_initialized()
    is_writer = (not distributed) or dist.get_rank(process_group) == 0
    ctl_dev = _collective_device(model) if distributed else None
    RunningScore = None
    if pixel_metric:
        from .text_metrics import RunningScore

    def save(path):
        """Checkpoints are written by rank 0 only (every rank writing the same path would race); the others wait, so the
        file is complete when any rank returns.  The BatchNorm buffers in it are rank 0's.  Called at rank-uniform points
        only (the decision values are all-reduced first)."""
        if is_writer:
            torch.save(model.state_dict(), path)
        if distributed:
            dist.barrier(group=process_group)

    best_test, best_train = float('inf'), float('inf')
    history = []
    steps = 0
    for epoch in range(epochs):
        model.train()
        running = RunningScore(2) if RunningScore is not None else None
        train_sum = None
        for batch in train_loader:
            if device is not None:
                batch = {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in batch.items()}
            steps += 1
            preds, losses = trainer.step(batch, None)
            if lrs_mode == 'poly' and scheduler is not None:
                scheduler.step()
            if running is not None:
                running.update_device(preds[:, 0, :, :], batch['prob_map'], batch['supervision_mask'], thresh)
            train_sum = losses[4].clone() if train_sum is None else train_sum + losses[4]  # device-side sums: no per-step host sync
        score = running.get_scores()[0] if (train_sum is not None and running is not None) else {}
        train_loss = float(train_sum) if train_sum is not None else float('nan')
        test_loss, test_score = float('nan'), {}
        if test_loader is not None:
            test_loss, test_score = evaluate(model, criterion, test_loader, thresh=thresh, device=device, pixel_metric=pixel_metric)
        local = (train_loss, test_loss)
        train_loss, test_loss = _mean_over_ranks(local, process_group, ctl_dev)  # rank-uniform from here on
        rec = {'epoch': epoch + 1, 'global_steps': steps, 'lr': optimizer.param_groups[0]['lr'], 'train_loss_sum': train_loss,
               'train_loss': train_loss / max(len(train_loader), 1), 'train_score': score}
        if distributed:
            rec['rank_local'] = {'train_loss_sum': local[0], 'test_loss': local[1]}
        if test_loader is not None:
            rec.update(test_loss=test_loss, test_score=test_score)
            if test_loss <= best_test and train_loss <= best_train:
                best_test, best_train = test_loss, train_loss
                if best_cp_path:
                    save(best_cp_path)
                rec['saved_best'] = True
            if lrs_mode == 'reduce' and scheduler is not None:
                scheduler.step(test_loss)
        if log is not None:
            log(rec)
        history.append(rec)
    if last_cp_path:
        save(last_cp_path)
    if own_trainer:
        trainer.unfreeze_heap()
    return history
