"""-m gpu: world_size == 2 on a device, before an 8-GPU node runs it for the first time.

The reference is single-device (/root/reference/src/train.py:96-98); the data-parallel control path is this build's own:
DBTrainer's deferred start-up broadcasts, the flat-gradient all-reduce, the replica checksum, bench.py's rank-0-only parity
gate beside a live process group.  One GPU is visible here and RCCL refuses two ranks per device, so the two ranks move the
same DEVICE tensors through gloo (train.init_distributed: DBN_DIST_BACKEND=gloo, DBN_DIST_ONE_DEVICE=1).  The children are
started by tests/conftest.py before this process touches the GPU; the tests read their verdicts.  Tolerances are the golden
tests' (two processes share the GPU: no bit-level claims)."""
import json
import os

import pytest

pytestmark = pytest.mark.gpu


def _collect(child, what):
    if not child:
        pytest.skip('the %s child was not started (tests/conftest.py: needs a visible GPU)' % what)
    log = open(child['log']).read()[-4000:]
    assert not child.get('timed_out'), 'the %s child did not finish in time and was terminated:\n%s' % (what, log)
    rc = child['proc'].wait(timeout=600)
    if 'gloo' in log and ('does not support' in log or 'No backend type associated with device type cuda' in log):
        pytest.skip('this torch build\'s gloo cannot move HIP tensors: ' + log[-400:])
    return rc, log


@pytest.mark.timeout(900)
def test_real_trainer_with_two_ranks_on_one_device(dist2_child):
    """(a) gradients after the all-reduce == the reference's mean of the per-shard gradients (tests/golden/dp_2x1x128.npz);
    (b) parameter checksum spread 0 after 1 and 3 steps and after a bucketed step; (c) rank 1, started from different weights,
    adopts rank 0's through sync_from_rank0; a distributed=False trainer on rank 0 alone does not hang the group."""
    rc, log = _collect(dist2_child, 'two-rank trainer')
    for r in (1, ):
        p = dist2_child['out'] + '.rank%d' % r
        if os.path.exists(p):
            pytest.fail('rank %d: %s' % (r, json.load(open(p)).get('error')))
    assert rc == 0, 'torch.distributed.run failed (rc %d):\n%s' % (rc, log)
    assert os.path.exists(dist2_child['out']), log
    v = json.load(open(dist2_child['out']))
    print(json.dumps(v, indent=1))
    assert 'error' not in v, v.get('error')
    assert v['param_spread_before_sync'] > 0.0
    assert v['param_spread_after_step1'] == 0.0 and v['param_spread_after_step3'] == 0.0 and v['param_spread_bucketed'] == 0.0
    assert not v['grads_bad'], v['grads_bad']
    assert v['ok'], v


@pytest.mark.timeout(900)
def test_bench_two_ranks_with_the_parity_gate_on(bench2_child):
    """bench.py launched as the driver launches it for N = 2, parity gate enabled: rank 0's gate must be a rank-local
    computation (advisor, round 5: a gate trainer on the default group paired its start-up broadcasts with the other ranks'
    all-reduce).  The line must come out with parity.ok, n_gpus 2 and replicas that did not diverge."""
    rc, log = _collect(bench2_child, 'two-rank bench')
    assert rc == 0, 'bench.py --gpus 2 failed (rc %d):\n%s' % (rc, log)
    lines = [ln for ln in open(bench2_child['log']).read().splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1, log
    d = json.loads(lines[0])
    print({k: d[k] for k in ('value', 'n_gpus', 'ms_per_step', 'parity', 'data_parallel')})
    assert d['n_gpus'] == 2 and d['config']['global_batch'] == 32 and d['value'] > 0
    assert d['parity'] is not None and d['parity']['ok']
    assert d['data_parallel']['world_seen'] == 2 and d['data_parallel']['param_checksum']['max_spread_over_ranks'] == 0.0
