"""Child process of tests/test_dist2_gpu.py (NOT a test module): the REAL DBTrainer with world_size == 2 on a device.

Two ranks share the one visible GPU (RCCL refuses two ranks per device, so the same device tensors go through gloo:
DBN_DIST_BACKEND=gloo, DBN_DIST_ONE_DEVICE=1 — train.init_distributed).  What runs is the trainer's whole N > 1 control path
— the deferred start-up broadcasts (DBTrainer._warm_arena_then_sync / sync_from_rank0), the flat-gradient all-reduce with the
1/world mean folded into Adam, param_checksum — on the shards of the reference-generated golden tests/golden/dp_2x1x128.npz
(two shards of 1x3x128x128, the reference's per-shard gradients averaged).  Started by tests/conftest.py as

    python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P tests/dist_child2.py OUT.json

before anything in the pytest process touches the GPU.  Rank 0 writes a JSON verdict to OUT.json."""
import importlib.util
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(out_path):
    import numpy as np
    import torch
    import torch.distributed as dist
    from db_text_minimal_amd import DBLoss, DBTextModel, DBTrainer, FusedAdam
    from db_text_minimal_amd.train import init_distributed

    verdict = {'ok': False}
    rank = int(os.environ.get('RANK', '0'))
    try:
        gdir = os.path.join(ROOT, 'tests', 'golden')
        spec = importlib.util.spec_from_file_location('fixture_inputs', os.path.join(gdir, 'fixture_inputs.py'))
        fx = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(fx)
        z = np.load(os.path.join(gdir, 'dp_2x1x128.npz'))
        _, size, seed, _ = (int(v) for v in z['meta'])

        rank, local, world = init_distributed()
        assert world == 2 and dist.is_initialized() and local == 0, (rank, local, world)
        verdict['backend'] = dist.get_backend()
        dev = torch.device('cuda', local)
        img, gts = fx.synthetic_batch(2, size, seed=seed + 100)
        img, gts = img[rank:rank + 1].to(dev), gts[:, rank:rank + 1].contiguous().to(dev)

        # (c) rank 1 starts from DIFFERENT weights (seed + 1) and different BatchNorm buffers: the start-up broadcast inside the
        # first step must make it adopt rank 0's before anything is computed from them
        model = DBTextModel()
        model.load_state_dict(fx.procedural_fill({k: v.clone() for k, v in model.state_dict().items()}, seed + rank))
        model = model.to(dev).train()
        tr = DBTrainer(model, DBLoss(), FusedAdam(model, lr=0.005))
        assert tr.world == 2 and tr.distributed and tr._need_sync
        eng = model.engine
        eng.ensure_flat()
        _, spread0 = tr.param_checksum()
        verdict['param_spread_before_sync'] = spread0  # > 0: the two ranks really differ

        preds, losses = tr.step(img, gts)
        torch.cuda.synchronize()
        assert not tr._need_sync

        # (a) the post-all-reduce gradient (the SUM sits in the flat buffer; 1/world is folded into Adam) against the
        # reference's mean of the per-shard gradients
        grad_report = {}
        bad = []
        for f in z.files:
            if not (f.startswith('grad/') and f.endswith('/stats')):
                continue
            k = f[len('grad/'):-len('/stats')]
            if k.endswith('.bias') and ('conv.bias' in k or k.endswith(('.0.bias', '.3.bias'))):
                continue  # conv bias ahead of train-mode BN: analytically zero, the reference value is round-off noise
            st = z[f]
            g = (eng.grad_views[k].detach().double().cpu().reshape(-1)) / world
            scale = max(float(st[4]), -float(st[3]), 1e-30)
            key = 'grad/' + k + ('/full' if 'grad/' + k + '/full' in z.files else '/sample')
            ref = torch.from_numpy(z[key]).double().reshape(-1)
            got = g if key.endswith('full') else g[torch.from_numpy(fx.sample_idx(g.numel(), ref.numel()))]
            err = float((got - ref).abs().max()) / scale
            cos = float((got @ ref) / (got.norm() * ref.norm())) if float(ref.norm()) > 0 and float(got.norm()) > 0 else 1.0
            l2 = abs(float(g.pow(2).sum().sqrt()) - float(st[2])) / max(float(st[2]), 1e-30)
            # the tolerances of tests/test_model_gpu.py::check_grad_summary (ReLU-flip level)
            if not (err <= 5e-2 and cos >= 0.999 and l2 <= 2e-2 and bool(torch.isfinite(g).all())):
                bad.append([k, err, cos, l2])
            grad_report[k] = [err, cos, l2]
        verdict['grads_checked'] = len(grad_report)
        verdict['grads_bad'] = bad
        verdict['grad_worst'] = {'sample_err_over_scale': max(v[0] for v in grad_report.values()),
                                 'cos_min': min(v[1] for v in grad_report.values()), 'l2_rel': max(v[2] for v in grad_report.values())}
        verdict['loss_rank'] = float(losses[4])
        verdict['loss_ref'] = float(z['loss_rank%d' % rank])
        loss_ok = abs(verdict['loss_rank'] - verdict['loss_ref']) <= 1e-4 * max(1.0, abs(verdict['loss_ref']))

        # (b) three steps in all: the replicas hold identical parameters (same start, same averaged gradient, same Adam state)
        _, spread1 = tr.param_checksum()
        for _ in range(2):
            tr.step(img, gts)
        torch.cuda.synchronize()
        (psum, psq), spread3 = tr.param_checksum()
        verdict.update(param_spread_after_step1=spread1, param_spread_after_step3=spread3, param_sum=psum)
        # ... and so does the bucketed exchange under the backward pass
        tr.overlap_allreduce = True
        tr.step(img, gts)
        torch.cuda.synchronize()
        _, spread4 = tr.param_checksum()
        verdict['param_spread_bucketed'] = spread4
        # the rank-local trainer beside a live process group (bench.py's parity gate): no collective, no hang
        m2 = DBTextModel()
        m2.load_state_dict(fx.procedural_fill({k: v.clone() for k, v in m2.state_dict().items()}, seed))
        m2 = m2.to(dev).train()
        if rank == 0:
            t2 = DBTrainer(m2, DBLoss(), FusedAdam(m2, lr=0.005), distributed=False)
            assert t2.world == 1 and not t2._need_sync
            t2.step(img, gts)
            torch.cuda.synchronize()
        ok_t = torch.tensor([1.0 if (not bad and loss_ok) else 0.0], device=dev)
        dist.all_reduce(ok_t, op=dist.ReduceOp.MIN)
        verdict['all_ranks_ok'] = bool(ok_t.item() == 1.0)
        verdict['ok'] = bool(verdict['all_ranks_ok'] and spread0 > 0.0 and spread1 == 0.0 and spread3 == 0.0 and spread4 == 0.0
                             and verdict['grads_checked'] >= 60)
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:  # the parent test reports it
        import traceback
        verdict['error'] = 'rank %d: %s\n%s' % (rank, e, traceback.format_exc())
    if rank == 0 or 'error' in verdict:
        with open(out_path if rank == 0 else out_path + '.rank%d' % rank, 'w') as f:
            json.dump(verdict, f)


if __name__ == '__main__':
    main(sys.argv[1])
