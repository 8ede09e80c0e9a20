"""GPU: the data-parallel train step of the FULL product network on two ranks (e).  The ranks
are started by tests/conftest.py at session start -- before this process initialises the GPU -- and share the one card
of the test box over gloo (RCCL needs one device per rank); tools/ddp_rehearsal.py holds the assertions: bit-identical
replicas after 3 train-mode steps with ClipAdamW on DDP bucket views, `dummy_tensor` out of the reduction, one batch-dice
all-reduce each way per step, and the same parameters as a single process stepping on the global batch."""
import os

import pytest

from conftest import REHEARSAL_LOG

pytestmark = pytest.mark.gpu


def test_two_rank_train_step_of_the_full_network():
    assert os.path.exists(REHEARSAL_LOG), "the session-start hook did not run the rehearsal (run with -m gpu)"
    text = open(REHEARSAL_LOG).read()
    assert text.startswith("rc=0"), text[-3000:]
    assert "replicas bit-identical after 3 steps: True" in text
    assert "batch-dice all-reduces: 3 forward + 3 backward" in text
    assert "PASS" in text
