import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    import torch
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    out = {}
    for k in z.files:
        a = z[k]
        out[k] = a if a.dtype.kind in "US" else torch.from_numpy(a)
    return out


@pytest.fixture
def golden():
    return load_golden


def rel_err(a, b):
    """max |a-b| / max(|b|max, tiny): the 'relative fp32' measure used for parity bars."""
    import torch
    a, b = a.detach().double().flatten(), b.detach().double().flatten()
    return float((a - b).abs().max() / b.abs().max().clamp(min=1e-30))
