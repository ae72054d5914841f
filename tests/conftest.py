import json
import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, 'tests', 'golden')


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
