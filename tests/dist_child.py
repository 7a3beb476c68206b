"""Child process of tests/test_rccl_gpu.py (NOT a test module): two DBTrainer.step()s through backend='nccl' (= RCCL) with
one rank per process, in every combination of the bucketed / single all-reduce and the one- / two-stream backward, compared
bit for bit with the same steps taken before the process group existed.  Started by tests/conftest.py as a fresh process

    python -m torch.distributed.run --nproc-per-node 1 --master-addr 127.0.0.1 --master-port P tests/dist_child.py OUT.json

before anything in the pytest process touches the GPU.  Writes a JSON verdict to OUT.json."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(out_path):
    import torch
    import torch.distributed as dist
    from db_text_minimal_amd import DBLoss, DBTextModel, DBTrainer, FusedAdam
    from db_text_minimal_amd.train import init_distributed
    from oracle import dbnet_oracle as O

    seed, n, size, steps = 9, 2, 96, 2
    img, gts = O.synthetic_batch(n, size, seed=seed)
    verdict = {'ok': False, 'cases': []}

    def run(overlap_allreduce, overlap_wgrad):
        model = DBTextModel()
        model.load_state_dict(O.new_state(seed))
        model = model.to('cuda').train()
        model.engine.overlap_wgrad = overlap_wgrad
        tr = DBTrainer(model, DBLoss(), FusedAdam(model, lr=0.005))
        tr.overlap_allreduce = overlap_allreduce
        for _ in range(steps):
            preds, losses = tr.step(img.to('cuda'), gts.to('cuda'))
        torch.cuda.synchronize()
        return tr, (model.engine.flat_grad.clone(), model.engine.flat.clone(), preds.clone(), losses.clone())

    try:
        assert not dist.is_initialized()
        tr0, ref = run(True, True)  # no process group: the plain single-GPU step
        assert tr0.world == 1
        os.environ['DBN_FORCE_DIST'] = '1'
        rank, local, world = init_distributed()
        assert dist.is_initialized() and dist.get_backend() == 'nccl' and world == 1
        for ar in (True, False):
            for wg in (True, False):
                tr, got = run(ar, wg)
                same = all(torch.equal(a, b) for a, b in zip(ref, got))
                verdict['cases'].append({'overlap_allreduce': ar, 'overlap_wgrad': wg, 'bit_identical': bool(same),
                                         'max_abs_grad_diff': float((ref[0] - got[0]).abs().max())})
        # the N > 1 diagnostics of bench.py: replicas hold identical parameters; the exposed part of the exchange is timed
        _, spread = tr.param_checksum()
        tr.exchange_events = []
        tr.step(img.to('cuda'), gts.to('cuda'))
        torch.cuda.synchronize()
        verdict['param_spread'] = spread
        verdict['exchange_ms'] = [a.elapsed_time(b) for a, b in tr.exchange_events]
        verdict['diagnostics_ok'] = bool(spread == 0.0 and len(verdict['exchange_ms']) == 1 and verdict['exchange_ms'][0] >= 0.0
                                         and not tr._need_sync)
        # a real collective went through RCCL: sum over one rank of a device tensor, on the trainer's path
        t = torch.arange(8, device='cuda', dtype=torch.float32)
        dist.all_reduce(t)
        verdict['allreduce_identity'] = bool(torch.equal(t.cpu(), torch.arange(8, dtype=torch.float32)))
        verdict['ok'] = (all(c['bit_identical'] for c in verdict['cases']) and verdict['allreduce_identity']
                         and verdict['diagnostics_ok'])
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:  # the parent test reports it
        import traceback
        verdict['error'] = '%s\n%s' % (e, traceback.format_exc())
    with open(out_path, 'w') as f:
        json.dump(verdict, f)


if __name__ == '__main__':
    main(sys.argv[1])
