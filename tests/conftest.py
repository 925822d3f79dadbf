import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")

# The built libraries are git-ignored: a fresh checkout has none.  Build them (no-ops when up to date; hipcc cross-compiles without a
# GPU, ~50 s from scratch) BEFORE any test module imports regione_amd.torch_ops, which picks the op registration - C++ when
# libregione_torch.so exists - at import time.  A box without the toolchain keeps whatever is there (the GPU box gets prebuilt files).
try:
    from regione_amd import build as _build
    _build.build_lib(verbose=False)
    _build.build_torch_binding(verbose=False)
except Exception as _e:          # noqa: BLE001 - reported, not fatal: test_capi_symbols fails loudly if the library is really missing
    sys.stderr.write(f"[conftest] could not (re)build the HIP libraries: {_e}\n")


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


@pytest.fixture(autouse=True)
def _reset_plan_overrides():
    """Launch-plan knobs forced by a test (tests/plan_helpers.py, rgn_plan_override) never leak into the next one."""
    yield
    try:
        from regione_amd import _lib
        if _lib._lib is not None:
            _lib._lib.rgn_plan_override(None, 0)
    except Exception:          # noqa: BLE001 - a missing library is reported by the tests that need it
        pass


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
