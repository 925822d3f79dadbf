"""ctypes binding of libregione_hip.so (the C ABI declared in include/regione_hip.h).

The product path has NO CPU fallback: if the shared library is missing or a symbol cannot be
resolved this module raises at import/first use - loudly - instead of degrading to eager torch.
"""
from __future__ import annotations

import ctypes as C
import os

# torch ships its own libamdhip64; it must be in the process BEFORE libregione_hip.so is dlopen'ed so
# that both resolve to the same HIP runtime (loading ours first binds /opt/rocm's copy and the two
# runtimes then disagree about the device: "no ROCm-capable device is detected").
import torch  # noqa: F401

from . import build as _build
from .build import LIB_PATH

# developer A/B switch: load another build of the SAME C ABI (never a fallback - it must exist)
LIB_PATH = os.environ.get("RGN_LIB", LIB_PATH)

_c_void_p, _c_int, _c_float = C.c_void_p, C.c_int, C.c_float



class QkvEpilogue(C.Structure):
    """struct rgn_qkv_epilogue (include/regione_hip.h)."""
    _fields_ = [("wq", _c_void_p), ("wk", _c_void_p), ("cos_q", _c_void_p), ("sin_q", _c_void_p), ("cos_k", _c_void_p),
                ("sin_k", _c_void_p), ("kv_rows", _c_void_p), ("k_slab", _c_void_p), ("vt_slab", _c_void_p),
                ("row_base", _c_int), ("skv_pad", _c_int), ("k_col", _c_int), ("v_col", _c_int), ("q_col", _c_int),
                ("heads", _c_int), ("eps", _c_float), ("fp16_roundtrip", _c_int)]


_qkv_p = C.POINTER(QkvEpilogue)


class GemmProblem(C.Structure):
    """struct rgn_gemm_problem (include/regione_hip.h): one problem of rgn_gemm_group."""
    _fields_ = [("A", _c_void_p), ("W", _c_void_p), ("wscale", _c_void_p), ("bias", _c_void_p), ("C", _c_void_p),
                ("gate", _c_void_p), ("resid", _c_void_p), ("qkv", _qkv_p), ("lda", _c_int), ("ldc", _c_int), ("M", _c_int)]


_prob_p = C.POINTER(GemmProblem)

# name -> argtypes (restype is always int unless listed in _RESTYPE)
SIGNATURES = {
    "rgn_version": [],
    "rgn_abi_struct_bytes": [],
    "rgn_plan_override": [C.c_char_p, _c_int],
    "rgn_plan_override_get": [C.c_char_p, C.POINTER(C.c_int)],
    "rgn_gemm_last_plan": [],
    "rgn_attention_last_plan": [],
    "rgn_attention_plan_query": [_c_int, _c_int, _c_int, C.c_size_t],
    "rgn_gemm_plan_query": [C.POINTER(C.c_int), _c_int, _c_int, _c_int, _c_int, _c_int, C.c_size_t],
    "rgn_last_error": [],
    "rgn_device_info": [C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_size_t)],
    "rgn_arp_partition": [_c_void_p, _c_int, _c_void_p, _c_int, _c_void_p, _c_int, _c_float, _c_float,
                          _c_int, _c_int, _c_int, _c_int, _c_int, _c_void_p, _c_void_p, _c_void_p, _c_void_p,
                          _c_void_p, _c_void_p, _c_void_p],
    "rgn_morph_compact": [_c_void_p, _c_int, _c_int, _c_int, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p],
    "rgn_gather_rows": [_c_void_p, _c_void_p, _c_void_p, _c_int, _c_int, _c_void_p],
    "rgn_scatter_rows": [_c_void_p, _c_void_p, _c_void_p, _c_int, _c_int, _c_void_p],
    "rgn_euler_step": [_c_void_p, _c_int, _c_void_p, _c_int, _c_void_p, _c_void_p, _c_float, _c_float, _c_int,
                       _c_int, _c_void_p],
    "rgn_avd_apply": [_c_void_p, _c_int, _c_void_p, _c_float, _c_int, _c_void_p, _c_int, _c_int, _c_void_p],
    "rgn_cfg_combine": [_c_void_p, _c_void_p, _c_void_p, _c_int, _c_float, _c_int, _c_float, _c_int, _c_int, _c_void_p],
    "rgn_gemm_bf16": [_c_void_p, _c_int, _c_void_p, _c_int, _c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_int,
                      _c_int, _c_int, _c_void_p, _c_void_p, _c_void_p, _c_void_p, C.c_size_t, _c_void_p],
    "rgn_gemm_workspace_bytes": [],
    "rgn_gemm_group": [_prob_p, _c_int, _c_int, _c_int, _c_int, _c_int, _c_void_p, C.c_size_t, _c_void_p],
    "rgn_gemm_bf16_qkv": [_c_void_p, _c_int, _c_void_p, _c_int, _c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_int, _c_int,
                          _qkv_p, _c_void_p, C.c_size_t, _c_void_p],
    "rgn_gemm_bf16_qkv_pair": [_c_void_p, _c_int, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_int, _qkv_p,
                               _c_void_p, _c_int, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_int, _qkv_p,
                               _c_int, _c_int, _c_void_p, C.c_size_t, _c_void_p],
    "rgn_gemm_bf16_pair": [_c_void_p, _c_int, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_int, _c_void_p, _c_void_p,
                           _c_void_p, _c_int, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_int, _c_void_p, _c_void_p,
                           _c_int, _c_int, _c_int, _c_int, _c_void_p, C.c_size_t, _c_void_p],
    # fp8 (e4m3fn) weights + per-output-channel fp32 scales: the bf16 signatures with a scale pointer behind each W
    "rgn_gemm_w8": [_c_void_p, _c_int, _c_void_p, _c_int, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_int,
                    _c_int, _c_int, _c_void_p, _c_void_p, _c_void_p, _c_void_p, C.c_size_t, _c_void_p],
    "rgn_gemm_w8_pair": [_c_void_p, _c_int, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_int, _c_void_p, _c_void_p,
                         _c_void_p, _c_int, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_int, _c_void_p, _c_void_p,
                         _c_int, _c_int, _c_int, _c_int, _c_void_p, C.c_size_t, _c_void_p],
    "rgn_gemm_w8_qkv": [_c_void_p, _c_int, _c_void_p, _c_int, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_int,
                        _c_int, _qkv_p, _c_void_p, C.c_size_t, _c_void_p],
    "rgn_gemm_w8_qkv_pair": [_c_void_p, _c_int, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_int, _qkv_p,
                             _c_void_p, _c_int, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_int, _qkv_p,
                             _c_int, _c_int, _c_void_p, C.c_size_t, _c_void_p],
    "rgn_gemv_bf16": [_c_void_p, _c_int, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_int, _c_int,
                      _c_void_p],
    "rgn_rms_norm_rows": [_c_void_p, _c_int, _c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_float, _c_void_p],
    "rgn_silu_bf16": [_c_void_p, _c_void_p, C.c_size_t, _c_void_p],
    "rgn_add_bf16": [_c_void_p, _c_void_p, _c_void_p, C.c_size_t, _c_void_p],
    "rgn_sel_rows": [_c_void_p, _c_int, _c_int, _c_void_p, _c_void_p],
    "rgn_fill_zero": [_c_void_p, C.c_size_t, _c_void_p],
    "rgn_ln_modulate": [_c_void_p, _c_int, _c_void_p, _c_int, _c_int, _c_int, _c_float, _c_int, _c_void_p,
                        _c_void_p, _c_void_p, _c_void_p, _c_void_p],
    "rgn_ln_modulate_segs": [_c_void_p, _c_int, _c_void_p, _c_int, _c_int, _c_int, _c_float, _c_int, C.POINTER(_c_int),
                             C.POINTER(_c_void_p), C.POINTER(_c_void_p), _c_void_p],
    "rgn_qk_norm_rope_store": [_c_void_p, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_void_p,
                               _c_void_p, _c_void_p, _c_void_p, _c_float, _c_void_p, _c_void_p, _c_void_p,
                               _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_void_p],
    "rgn_attention": [_c_void_p, _c_int, _c_void_p, _c_void_p, _c_int, _c_void_p, _c_int, _c_int, _c_int, _c_int,
                      _c_float, _c_void_p, C.c_size_t, _c_void_p],
    "rgn_attention_bounded": [_c_void_p, _c_int, _c_void_p, _c_void_p, _c_int, _c_void_p, _c_int, _c_int, _c_int, _c_int,
                              _c_float, _c_float, _c_void_p, C.c_size_t, _c_void_p],
    "rgn_attention_workspace_bytes": [_c_int, _c_int],
    # f4: VAE decoder (implicit-GEMM convolutions + row kernels, csrc/vae.hip)
    "rgn_conv_bf16": [_c_void_p, _c_int, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int,
                      _c_void_p, C.POINTER(C.c_int), _c_void_p],
    "rgn_conv_s2_bf16": [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_int, _c_int, _c_void_p, _c_void_p],
    "rgn_conv_up2_bf16": [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_int, _c_int, _c_void_p, _c_void_p,
                          C.POINTER(C.c_int), _c_void_p],
    "rgn_groupnorm_workspace_bytes": [],
    "rgn_groupnorm_partial_bytes": [],
    "rgn_groupnorm_silu": [_c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_void_p, _c_void_p, _c_float, _c_int, _c_void_p, _c_int, _c_void_p],
    "rgn_upsample2x": [_c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_void_p],
    "rgn_softmax_rows": [_c_void_p, _c_int, _c_int, _c_int, _c_float, _c_void_p],
    "rgn_nchw_to_padded": [_c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_int, _c_void_p],
    "rgn_padded_to_nchw": [_c_void_p, _c_int, _c_void_p, _c_int, _c_int, _c_int, _c_void_p],
}
_RESTYPE = {"rgn_last_error": C.c_char_p, "rgn_abi_struct_bytes": C.c_size_t, "rgn_attention_workspace_bytes": C.c_size_t,
            "rgn_gemm_workspace_bytes": C.c_size_t, "rgn_groupnorm_workspace_bytes": C.c_size_t,
            "rgn_groupnorm_partial_bytes": C.c_size_t}

_lib = None


class RegionEHipError(RuntimeError):
    pass


def lib():
    """Load (once) and return the ctypes handle; raise if the native library is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RegionEHipError(
            f"{LIB_PATH} not found: the HIP extension is not built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc); there is no CPU fallback.")
    if os.path.abspath(LIB_PATH) == os.path.abspath(_build.LIB_PATH) and os.path.isdir(_build.CSRC):
        # the in-tree library beside its sources: it must be the build of THESE sources (a stale .so would make every measurement and
        # parity claim about the tree a claim about something else)
        built, now = _build.built_from(), _build.csrc_hash()
        if built and built != now:
            raise RegionEHipError(f"{LIB_PATH} was built from kernel sources {built}, the tree holds {now}: rebuild with "
                                  "`python -m regione_amd.build` (or `__graft_entry__.build()`)")
    h = C.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(h, name)           # AttributeError if the symbol is missing: loud by design
        fn.argtypes = argtypes
        fn.restype = _RESTYPE.get(name, C.c_int)
    # the ctypes mirrors of the structs passed by pointer must be the library's layout (a library built from another header)
    want = C.sizeof(QkvEpilogue) * 1000 + C.sizeof(GemmProblem)
    if h.rgn_abi_struct_bytes() != want:
        raise RegionEHipError(f"{LIB_PATH}: struct layout {h.rgn_abi_struct_bytes()} != the Python binding's {want} "
                              f"(library ABI version {h.rgn_version()}): rebuild with `python -m regione_amd.build --force`")
    _lib = h
    return h


import contextlib as _contextlib

PLAN_KEYS = ("gemm_pieces", "gemm_geometry", "gemm_asm", "gemm_quarter", "attn_waves", "attn_split", "attn_streamk", "attn_asm")


@_contextlib.contextmanager
def plan_override(**knobs):
    """Force launch-plan knobs for the duration of a `with` block (rgn_plan_override, include/regione_hip.h) - tests and sweep
    tools only; every knob returns to the value it HAD on exit (nested blocks and an RGN_PLAN_OVERRIDE preset survive).

        with _lib.plan_override(gemm_pieces=3, gemm_geometry=256): ops.gemm(...)"""
    h = lib()
    before = {}
    try:
        for k, v in knobs.items():
            old = C.c_int(-1)
            check(h.rgn_plan_override_get(k.encode(), C.byref(old)), f"rgn_plan_override_get({k})")
            check(h.rgn_plan_override(k.encode(), int(v)), f"rgn_plan_override({k})")
            before[k] = old.value
        yield
    finally:
        for k, old in before.items():
            h.rgn_plan_override(k.encode(), old)


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = lib().rgn_last_error().decode(errors="replace")
        raise RegionEHipError(f"{what or 'libregione_hip'} failed (rc={rc}): {msg}")
