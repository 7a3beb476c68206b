"""-m gpu: the data-parallel step through backend='nccl' (RCCL) on a real GPU — the stream ordering of the overlapped
exchange (engine.grad_ready_hook announced from the side stream, BucketedAllReduce.finish joined on the main stream,
train.py:125-153).  tests/test_dp_gloo.py covers the arithmetic of the exchange with world_size 2 on CPU; this covers the
RCCL/stream side: with one rank the all-reduce is the identity, so gradients, parameters, maps and losses after two steps
must be BIT-IDENTICAL to the same steps without a process group, in all four overlap modes.  A missed wait would show as
a difference (partly written gradients reduced, or Adam reading gradients before the collective finished)."""
import json
import os

import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.timeout(900)
def test_nccl_step_is_bit_identical_to_the_single_process_step(rccl_child):
    if not rccl_child:
        pytest.skip('the RCCL child process was not started (tests/conftest.py: needs a visible GPU)')
    assert not rccl_child.get('timed_out'), ('the RCCL child did not finish within 900 s and was terminated (tests/conftest.py):\n'
                                             + open(rccl_child['log']).read()[-3000:])
    rc = rccl_child['proc'].wait(timeout=800)
    log = open(rccl_child['log']).read()[-3000:]
    assert rc == 0, 'torch.distributed.run failed (rc %d):\n%s' % (rc, log)
    assert os.path.exists(rccl_child['out']), log
    v = json.load(open(rccl_child['out']))
    print(json.dumps(v, indent=1))
    assert 'error' not in v, v.get('error')
    assert len(v['cases']) == 4 and v['ok'], v
