"""No kernel of the training step reads memory that was never written (scripts/poison_check.py): every buffer the
package takes uninitialised from torch - including the library's scratch - is filled with NaN / all-ones first, and
three seeded iterations must still equal the clean run BIT FOR BIT, in the split-precision and the fp32 arithmetic.
Runs in a subprocess (the check replaces torch.empty & co. for the whole process)."""
import os
import subprocess
import sys

import pytest

from conftest import REPO

pytestmark = pytest.mark.gpu


def test_poisoned_buffers_leave_the_step_bit_identical(cuda):
    env = dict(os.environ, GANGEALING_SYNTHETIC='1')
    res = subprocess.run([sys.executable, os.path.join(REPO, 'scripts', 'poison_check.py'), 'small', 'cluster'], env=env,
                         cwd=REPO, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0 and 'POISON CHECK PASSED' in res.stdout, (res.stdout[-3000:], res.stderr[-2000:])
    assert res.stdout.count('IDENTICAL to the clean run') == 4
