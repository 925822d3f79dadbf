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


def test_argument_validation_returns_codes_and_messages_without_touching_the_gpu():
    """Every entry point validates before it launches: bad shapes / alignment / null pointers give a negative
    RGN_E_* code and a message through rgn_last_error() (the Python wrappers turn it into RegionEHipError)."""
    h = _lib.lib()
    P = 0x10000                                          # a plausible, 16-byte aligned, never dereferenced address

    def msg():
        return h.rgn_last_error().decode()
    # GEMM: K must be a multiple of 64, strides multiples of 8, pointers 16-byte aligned, gate/resid for epilogue 2
    assert h.rgn_gemm_bf16(P, 96, P, 96, None, P, 64, 8, 64, 96, 0, 0, None, None, None, None, 0, None) < 0 and "multiple of 64" in msg()
    assert h.rgn_gemm_bf16(P, 68, P, 64, None, P, 64, 8, 64, 64, 0, 0, None, None, None, None, 0, None) < 0 and "strides" in msg()
    assert h.rgn_gemm_bf16(P + 2, 64, P, 64, None, P, 64, 8, 64, 64, 0, 0, None, None, None, None, 0, None) < 0 and "aligned" in msg()
    assert h.rgn_gemm_bf16(P, 64, P, 64, None, P, 64, 8, 64, 64, 2, 0, None, None, None, None, 0, None) < 0 and "gate" in msg()
    assert h.rgn_gemm_bf16(None, 64, P, 64, None, P, 64, 8, 64, 64, 0, 0, None, None, None, None, 0, None) < 0
    assert h.rgn_gemm_bf16(P, 64, P, 64, None, P, 64, 0, 64, 64, 0, 0, None, None, None, None, 0, None) == 0      # M = 0: nothing to do
    # fused QKV epilogue: descriptor required, column blocks 256-aligned
    assert h.rgn_gemm_bf16_qkv(P, 64, P, 64, None, P, 768, 8, 768, 64, 768, None, None, 0, None) < 0 and "rgn_qkv_epilogue" in msg()
    e = _lib.QkvEpilogue(P, P, P, P, P, P, None, P, P, 0, 64, 0, 128, 512, 2, 1e-6)      # v_col = 128: not 256-aligned
    assert h.rgn_gemm_bf16_qkv(P, 64, P, 64, None, P, 768, 8, 768, 64, 768, e, None, 0, None) < 0 and "256-aligned" in msg()
    # attention / row kernels
    assert h.rgn_attention(None, 0, P, P, 64, P, 0, 8, 8, 2, 0.1, None, 0, None) < 0
    assert h.rgn_qk_norm_rope_store(P, 12, 0, 0, 0, 8, 2, 0, None, None, P, P, 1e-6, P, P, P, P, None, P, P, 64, None) < 0
    assert h.rgn_gather_rows(None, P, P, 4, 128, None) < 0
    assert h.rgn_gemv_bf16(P, 64, P, None, P, 64, 9, 64, 64, 0, None) < 0                 # batch > 4
    # round-5 entries
    assert h.rgn_add_bf16(None, P, P, 8, None) < 0 and h.rgn_add_bf16(P, P, P, 0, None) == 0
    assert h.rgn_sel_rows(None, 4, 8, P, None) < 0 and h.rgn_sel_rows(P, -1, 8, P, None) < 0 and h.rgn_sel_rows(None, 0, 0, P, None) == 0
    assert h.rgn_fill_zero(None, 16, None) < 0 and h.rgn_fill_zero(None, 0, None) == 0


def test_abi_stamp_and_plan_override_hook():
    """The library reports the header's ABI version and the struct sizes the ctypes mirrors were written for (a binding built against
    another header refuses to run, regione_amd/torch_ops.py); the launch-plan hook knows exactly the documented knobs."""
    import ctypes as C
    h = _lib.lib()
    text = open(os.path.join(ROOT, "include", "regione_hip.h")).read()
    assert h.rgn_version() == int(re.search(r"#define RGN_ABI_VERSION (\d+)", text).group(1))
    assert h.rgn_abi_struct_bytes() == C.sizeof(_lib.QkvEpilogue) * 1000 + C.sizeof(_lib.GemmProblem)
    for k in _lib.PLAN_KEYS:
        assert h.rgn_plan_override(k.encode(), 1) == 0 and h.rgn_plan_override(k.encode(), -1) == 0
    assert h.rgn_plan_override(b"RGN_GEMM_VARIANT", 1) < 0 and b"unknown key" in h.rgn_last_error()
    assert h.rgn_plan_override(None, 0) == 0
    with _lib.plan_override(gemm_pieces=1):
        pass
    # a scoped override restores what it found (advisor round 5): nested blocks keep the outer value
    def get(k):
        v = C.c_int(-7)
        assert h.rgn_plan_override_get(k.encode(), C.byref(v)) == 0
        return v.value
    with _lib.plan_override(gemm_pieces=3, attn_split=0):
        with _lib.plan_override(gemm_pieces=5):
            assert get("gemm_pieces") == 5 and get("attn_split") == 0
        assert get("gemm_pieces") == 3 and get("attn_split") == 0
    assert get("gemm_pieces") == -1 and get("attn_split") == -1
    assert h.rgn_plan_override_get(b"no_such_knob", C.byref(C.c_int())) < 0
    with pytest.raises(_lib.RegionEHipError):
        with _lib.plan_override(no_such_knob=1):
            pass


def test_a_binding_built_against_another_header_is_refused(tmp_path):
    """regione_amd/torch_ops.py compares the C++ binding's compiled-in ABI stamp with the HIP library's before any op can run."""
    import subprocess, sys, textwrap
    if not os.path.exists(os.path.join(ROOT, "regione_amd", "lib", "libregione_torch.so")):
        pytest.skip("C++ binding not built")
    code = textwrap.dedent("""
        import ctypes, sys
        sys.path.insert(0, %r)
        from regione_amd import _lib
        h = _lib.lib()
        real = h.rgn_version
        class Fake:                                   # the HIP library as a NEWER build would answer
            def __getattr__(self, n): return getattr(h, n)
            def rgn_version(self): return real() + 1
        _lib._lib = Fake()
        try:
            import regione_amd.torch_ops
        except _lib.RegionEHipError as e:
            print("REFUSED", "rebuild" in str(e))
        else:
            print("LOADED")
    """ % ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, RGN_TORCH_OPS="cpp"))
    assert "REFUSED True" in r.stdout, r.stdout + r.stderr


def test_the_in_tree_library_is_the_build_of_the_tree_sources(monkeypatch):
    """build_lib() stamps the library with the hash of the kernel sources it compiled (the hash the committed counter files carry);
    staleness is decided on content, and _lib.lib() refuses an in-tree library whose stamp names other sources - a stale .so would
    make every parity and bench statement about this tree a statement about something else."""
    from regione_amd import _lib, build
    assert build.built_from() == build.csrc_hash(), "conftest builds the library: the stamp must name the current sources"
    assert not build._stale()
    monkeypatch.setattr(build, "built_from", lambda: "0123456789abcdef")
    assert build._stale()
    monkeypatch.setattr(_lib, "_lib", None)
    with pytest.raises(_lib.RegionEHipError, match="rebuild"):
        _lib.lib()
    monkeypatch.undo()
    assert _lib.lib() is not None
