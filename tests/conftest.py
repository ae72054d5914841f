import json
import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, 'tests', 'golden')
# the tests build trainers without perceptual-loss weight files (none are reachable offline): opt in to the seeded
# random trunk that GangealingTrainer otherwise refuses (gangealing_amd/train_step.py: allow_random_loss)
os.environ.setdefault('GANGEALING_SYNTHETIC', '1')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def load_golden(name):
    """-> list of dict cases; 'meta' decoded from JSON."""
    z = np.load(os.path.join(GOLDEN, name + '.npz'))
    cases = {}
    for key in z.files:
        case, field = key.split('/', 1)
        v = z[key]
        if field == 'meta':
            v = json.loads(bytes(v.tolist()).decode())
        cases.setdefault(case, {})[field] = v
    return [cases[k] for k in sorted(cases)]


def has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(scope='session')
def cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    return torch.device('cuda:0')


# ---- measured-parity report: GPU tests record the errors they observed (not just pass / fail); written at session end
# ---- to gpurun_out/parity_report.json (copied to profiles/parity_rNN.json for the round's record)
PARITY = {}


def record_parity(test, mode, tensor, got, ref, extra=None):
    """Store max |got - ref|, the reference scale and the relative figure; returns the max abs error."""
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    err = float(np.abs(got - ref).max()) if got.size else 0.0
    scale = float(np.abs(ref).max()) if ref.size else 0.0
    entry = dict(max_abs_err=err, ref_max_abs=scale, rel_to_max=(err / scale if scale > 0 else 0.0))
    if extra:
        entry.update(extra)
    PARITY.setdefault(test, {}).setdefault(mode, {})[tensor] = entry
    return err


def pytest_sessionfinish(session, exitstatus):
    if not PARITY:
        return
    out_dir = os.path.join(REPO, 'gpurun_out')
    try:
        os.makedirs(out_dir, exist_ok=True)
        with open(os.path.join(out_dir, 'parity_report.json'), 'w') as f:
            json.dump(PARITY, f, indent=1, sort_keys=True)
    except OSError:
        pass
