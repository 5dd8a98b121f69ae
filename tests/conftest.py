import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name + '.npz'))
    return load


def has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_sessionfinish(session, exitstatus):
    """the measured forward / rollout errors of this session (tests/test_gpu_model.py: MEASURED) next to the test log"""
    measured = {}
    for name, mod in list(sys.modules.items()):          # (the module may be imported under two names: as a test and by its siblings)
        if name.endswith('test_gpu_model'):
            for k, v in (getattr(mod, 'MEASURED', None) or {}).items():
                measured.setdefault(k, []).extend(v)
    if not measured:
        return
    import json
    out = os.path.join(ROOT, 'gpurun_out')
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, 'forward_errors.json'), 'w') as f:
            json.dump({k: [float('%.3e' % v) for v in vs] for k, vs in sorted(measured.items())}, f, indent=1)
    except OSError:
        pass
