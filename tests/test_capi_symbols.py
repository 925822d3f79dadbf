"""CPU-side checks of the drop-in boundary: the C-ABI library builds, loads, and exports every
symbol include/regione_hip.h declares (no compute calls - there is no GPU in the build container)."""
import ctypes
import os
import re

import pytest

from regione_amd import _lib, build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "regione_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rgn_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_expected_surface():
    syms = declared_symbols()
    for s in ("rgn_arp_partition", "rgn_gather_rows", "rgn_scatter_rows", "rgn_euler_step", "rgn_avd_apply",
              "rgn_gemm_bf16", "rgn_attention", "rgn_qk_norm_rope_store", "rgn_ln_modulate"):
        assert s in syms


def test_library_builds_and_exports_every_declared_symbol():
    path = build.build_lib()
    assert os.path.exists(path)
    h = ctypes.CDLL(path)
    for s in declared_symbols():
        assert hasattr(h, s), f"{s} declared in include/regione_hip.h but not exported"
    assert set(declared_symbols()) == set(_lib.SIGNATURES), "ctypes table out of sync with the header"
    assert h.rgn_version() >= 100


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.RegionEHipError, match="no CPU fallback"):
        _lib.lib()


def test_ops_refuse_cpu_tensors():
    import torch
    from regione_amd import ops
    with pytest.raises(_lib.RegionEHipError):
        ops.gather_rows(torch.zeros(1, 4, 64), torch.zeros(1, 2, dtype=torch.int64))
