import glob
import json
import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN_DIR = os.path.join(REPO, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    config.addinivalue_line('markers', 'slow: minutes of host time and ~10 GB of host memory (skipped unless HPK_SLOW=1)')


@pytest.fixture(scope='session', autouse=True)
def _torch_before_libhpk(request):
    """GPU sessions: PyTorch bundles its own HIP runtime; when libhpk's (system ROCm) initialises first, torch no
    longer finds the device.  Touch the GPU through torch once before any hpk context exists."""
    expr = request.config.getoption('-m') or ''
    if 'gpu' in expr and 'not gpu' not in expr:
        try:
            import torch
            if torch.cuda.is_available():
                torch.zeros(1, device='cuda')
        except ImportError:
            pass
    yield


class Golden(object):
    """One fixture produced by oracle/gen_golden.py from the real reference."""

    def __init__(self, path):
        self.z = np.load(path, allow_pickle=False)
        self.meta = json.loads(str(self.z['meta']))
        self.name = self.meta['name']
        self.mode = self.meta['mode']
        self.params = self.meta['params']

    def __getitem__(self, k):
        return self.z[k]

    def __contains__(self, k):
        return k in self.z.files

    @property
    def mw(self):
        return min(self.params['ww']) if self.mode == 'hiccups' else self.params['ww']


def golden_names(mode=None):
    out = []
    for p in sorted(glob.glob(os.path.join(GOLDEN_DIR, '*.npz'))):
        name = os.path.basename(p)[:-4]
        if name.startswith('ref_'):          # reference-pinned outputs above fixture size: tests/refbig.py
            continue
        if mode is None or name.startswith(mode):
            out.append(name)
    return out


def load_golden(name):
    return Golden(os.path.join(GOLDEN_DIR, name + '.npz'))


@pytest.fixture
def golden():
    return load_golden
