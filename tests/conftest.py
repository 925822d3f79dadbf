import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """`gpu`-marked tests need a HIP device: skipped (not failed) on a box without one, whatever -m says."""
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="needs a HIP GPU (torch.cuda.is_available() is False)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def load_golden(name):
    """npz -> dict of torch tensors / numpy scalars; '__bf16' / '__f16' keys are bit patterns."""
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    out = {}
    for k in z.files:
        v = z[k]
        if k.endswith("__bf16"):
            out[k[:-6]] = torch.from_numpy(v.view(np.int16).copy()).view(torch.bfloat16)
        elif k.endswith("__f16"):
            out[k[:-5]] = torch.from_numpy(v.view(np.int16).copy()).view(torch.float16)
        elif v.dtype.kind in "US" or v.ndim == 0:
            out[k] = v if v.ndim else v.item()
        else:
            out[k] = torch.from_numpy(v.copy())
    return out


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = load_golden(name)
        return cache[name]
    return get


def has_gpu():
    return torch.cuda.is_available()
