"""CPU, world_size 2 over gloo: the data-parallel exchange of the per-step loop.  Each rank
computes the gradients of ITS shard with the CPU oracle, lays them out with the product's flat
layout and runs the product's one-collective-per-step function; the result must be the mean of
the per-shard gradients, pinned against the reference-generated golden."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from db_text_minimal_amd.engine import DEAD_PREFIXES, flat_layout
from db_text_minimal_amd.train import allreduce_flat_grads
from oracle import dbnet_oracle as O


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, size, seed, out_path):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.set_num_threads(4)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        img, gts = O.synthetic_batch(world, size, seed=seed + 100)
        _, losses, grads = O.loss_and_grads(O.new_state(seed), img[rank:rank + 1], gts[:, rank:rank + 1])
        keys = [k for k in O.trainable_keys() if not k.startswith(DEAD_PREFIXES)]
        offs, total = flat_layout([grads[k].numel() for k in keys])
        flat = torch.zeros(total)
        for k, off in zip(keys, offs):
            flat[off:off + grads[k].numel()] = grads[k].reshape(-1)
        scale = allreduce_flat_grads(flat, world)  # one all_reduce call
        flat *= scale
        if rank == 0:
            np.savez(out_path, flat=flat.numpy(), offs=np.array(offs), keys=np.array(keys), loss=losses[4])
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_rank_gradient_allreduce_matches_reference_golden(tmp_path, golden_dir):
    z = np.load(os.path.join(golden_dir, 'dp_2x1x128.npz'))
    _, size, seed, _ = (int(v) for v in z['meta'])
    out = str(tmp_path / 'dp.npz')
    mp.spawn(_worker, args=(2, _free_port(), size, seed, out), nprocs=2, join=True)
    r = np.load(out)
    assert abs(float(r['loss']) - float(z['loss_rank0'])) < 1e-5
    flat, offs, keys = r['flat'], r['offs'], list(r['keys'])
    assert len(keys) == 111 and flat.size >= 12269378  # live parameter tensors / elements
    for k in ('backbone.conv1.weight', 'backbone.layer4.1.conv2.weight', 'segmentation_body.conv.0.weight',
              'segmentation_head.thresh.6.weight', 'backbone.layer2.0.downsample.1.bias'):
        st = z['grad/' + k + '/stats']
        i = keys.index(k)
        n = int(np.prod([s for s in O.new_state(seed)[k].shape])) if k in O.new_state(0) else 0
        a = flat[offs[i]:offs[i] + n].astype(np.float64)
        assert abs(np.sqrt((a * a).sum()) - st[2]) <= 1e-5 * st[2], k
        assert abs(a.sum() - st[0]) <= 1e-4 * st[1] + 1e-12, k


def test_single_rank_is_a_noop():
    g = torch.arange(8.)
    assert allreduce_flat_grads(g, 1) == 1.0 and torch.equal(g, torch.arange(8.))


def _bucket_worker(rank, world, port, out_path):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from db_text_minimal_amd.train import GRAD_STAGES, BucketedAllReduce, bucket_ranges
        keys = [k for k in O.trainable_keys() if not k.startswith(DEAD_PREFIXES)]
        shapes = {k: s for k, s, _ in O.state_spec()}
        offs, total = flat_layout([int(np.prod(shapes[k])) for k in keys])
        g = torch.Generator().manual_seed(100 + rank)
        flat = torch.randn(total, generator=g)
        single = flat.clone()
        ranges = bucket_ranges(keys, offs, total)
        ex = BucketedAllReduce(flat, ranges, world)
        for stage in GRAD_STAGES[:3]:  # the order backward announces them; finish() issues the rest
            ex.ready(stage)
        scale = ex.finish()
        assert allreduce_flat_grads(single, world) == scale == 0.5
        if rank == 0:
            np.savez(out_path, same=bool(torch.equal(flat, single)), ranges=np.array([ranges[s] for s in GRAD_STAGES]), total=total)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_bucketed_allreduce_equals_the_single_collective(tmp_path):
    """The overlapped exchange (four contiguous buckets, async, in backward-completion order) sums exactly what the one
    flat all-reduce sums; the buckets tile the gradient buffer without gap or overlap."""
    out = str(tmp_path / 'bucket.npz')
    mp.spawn(_bucket_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    r = np.load(out)
    assert bool(r['same'])
    rg = sorted((int(a), int(b)) for a, b in r['ranges'])
    assert rg[0][0] == 0 and rg[-1][1] == int(r['total']) and all(a[1] == b[0] for a, b in zip(rg[:-1], rg[1:]))
    # layer4 alone is two thirds of the buffer: it travels under the backward of layers 3..1
    sizes = {tuple(x): x[1] - x[0] for x in rg}
    assert max(sizes.values()) > 0.6 * int(r['total'])


# ---- epoch loop under data parallelism: rank-uniform decisions (train.fit) ------------------------------------------------
class _StubModel(torch.nn.Module):
    """A stand-in with the module contract fit()/evaluate() use (the real model needs an MI355X): eval output [N,2,H,W]."""

    def __init__(self):
        super().__init__()
        self.w = torch.nn.Parameter(torch.zeros(1))

    def forward(self, x):
        return torch.stack([x[:, 0], x[:, 1]], 1) * 0 + self.w


class _StubTrainer:
    """step() returns a loss that depends on the rank's shard, like a real data-parallel step does."""

    def __init__(self, rank, train_losses):
        self.rank, self.train_losses, self.i = rank, train_losses, 0

    def step(self, batch, gts):
        # every real step contains the gradient all-reduce: a rank that skipped ahead would pair it with a barrier
        t = torch.ones(1)
        dist.all_reduce(t)
        v = self.train_losses[self.i]
        self.i += 1
        return batch['img'][:, :3], torch.tensor([0., 0., 0., 0., v])


def _fit_worker(rank, world, port, out_dir):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from db_text_minimal_amd.train import fit, replica_divergence
        model = _StubModel()
        opt = torch.optim.SGD(model.parameters(), lr=1.0)
        sched = torch.optim.lr_scheduler.ReduceLROnPlateau(opt, factor=0.5, patience=0)
        batch = {'img': torch.zeros(1, 3, 8, 8), 'prob_map': torch.zeros(1, 8, 8), 'supervision_mask': torch.ones(1, 8, 8),
                 'thresh_map': torch.zeros(1, 8, 8), 'text_area_map': torch.zeros(1, 8, 8)}
        # per-epoch losses chosen so that the RANK-LOCAL best-checkpoint rule disagrees between the ranks in epochs 2 and 3
        # (rank 0 alone would save in epoch 2, rank 1 alone in epoch 3) while the rank-mean rule saves in epochs 1 and 3
        train = {0: [4.0, 3.0, 3.5], 1: [4.0, 5.5, 3.0]}[rank]
        test = {0: [2.0, 1.5, 1.8], 1: [2.0, 2.7, 1.0]}[rank]
        ep = {'i': 0}

        def criterion(preds, gts):
            return torch.tensor(test[ep['i']])

        def log(rec):
            ep['i'] += 1

        hist = fit(model, criterion, opt, [batch], test_loader=[batch], epochs=3, scheduler=sched, lrs_mode='reduce',
                   best_cp_path=os.path.join(out_dir, 'best.pth'), last_cp_path=os.path.join(out_dir, 'last.pth'), log=log,
                   trainer=_StubTrainer(rank, train), pixel_metric=False)
        _, spread_same = replica_divergence(torch.arange(5.))
        _, spread_diff = replica_divergence(torch.arange(5.) + rank)
        torch.save({'saved': [bool(h.get('saved_best')) for h in hist], 'test': [h['test_loss'] for h in hist],
                    'train': [h['train_loss_sum'] for h in hist], 'lr': opt.param_groups[0]['lr'],
                    'local': [h['rank_local'] for h in hist], 'spread': (spread_same, spread_diff)},
                   os.path.join(out_dir, 'r%d.pt' % rank))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_fit_takes_rank_uniform_decisions_when_shard_losses_differ(tmp_path):
    """the best-checkpoint barrier and the plateau scheduler must see rank-uniform losses — with
    rank-local ones the ranks pair a barrier with the next epoch's gradient all-reduce (hang) and their learning rates drift."""
    mp.spawn(_fit_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)  # a hang would hit the timeout
    r0, r1 = (torch.load(str(tmp_path / ('r%d.pt' % r))) for r in (0, 1))
    assert r0['saved'] == r1['saved'] == [True, False, True]
    assert r0['test'] == r1['test'] == [2.0, pytest.approx(2.1), pytest.approx(1.4)]
    assert r0['train'] == r1['train'] and r0['lr'] == r1['lr'] == 0.5  # one plateau (epoch 2) on the mean test loss
    assert r0['local'] != r1['local']  # the shards really disagreed
    assert r0['spread'] == (0.0, pytest.approx(25.0)) and (tmp_path / 'best.pth').exists() and (tmp_path / 'last.pth').exists()
