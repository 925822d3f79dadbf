"""Tensor-level wrappers over the C ABI (include/regione_hip.h).

PyTorch is plumbing here: it owns device memory and the HIP stream; every arithmetic op of the hot
path runs in libregione_hip.so.  Wrappers pass raw device pointers + the current stream; they never
synchronise except where a host-visible count is explicitly requested.
"""
from __future__ import annotations

import math
from typing import Optional, Tuple

import torch

from . import _lib

F32, BF16 = 0, 1
EPI_BIAS, EPI_GELU, EPI_GATE_RESID = 0, 1, 2


def _dt(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    raise TypeError(f"unsupported dtype {t.dtype} (fp32 / bf16 only)")


def _p(t: Optional[torch.Tensor]) -> Optional[int]:
    if t is None:
        return None
    if not t.is_cuda:
        raise _lib.RegionEHipError("regione_amd ops need CUDA/HIP tensors: there is no CPU fallback")
    return t.data_ptr()


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _rows(t: torch.Tensor) -> torch.Tensor:
    """[1,L,D] or [L,D] -> contiguous [L,D] view."""
    if t.dim() == 3:
        assert t.shape[0] == 1, "region ops are per image (reference is batch-1, quirk A-5)"
        t = t[0]
    assert t.is_contiguous()
    return t


def padded(n: int, m: int = 64) -> int:
    return (n + m - 1) // m * m


# ---------------------------------------------------------------------------------------------
# region ops
# ---------------------------------------------------------------------------------------------
def arp_partition(sample: torch.Tensor, model_output: Optional[torch.Tensor], cond: torch.Tensor,
                  dt_final: float, threshold: float, h_tok: int, w_tok: int, erosion_dilation: bool = True,
                  want_sim: bool = False):
    """token_selector + one-step estimate.  Returns (edited_ids i64[1,K], unedited_ids i64[1,L-K],
    mask u8[L], raw u8[L], sim f32[L] or None).  ONE 4-byte D2H read (K_e) - the only host sync the
    partition needs; every later launch of the image is shape-static."""
    s, c = _rows(sample), _rows(cond)
    mo = _rows(model_output) if model_output is not None else None
    L, D = s.shape
    dev = s.device
    if c.shape != s.shape or (mo is not None and mo.shape != s.shape) or h_tok * w_tok != L:
        raise _lib.RegionEHipError(f"arp_partition: sample {tuple(s.shape)}, model_output {None if mo is None else tuple(mo.shape)}, cond "
                                   f"{tuple(c.shape)} must all be [L, D] with L = h_tok * w_tok = {h_tok * w_tok}")
    e = torch.empty(L, dtype=torch.int64, device=dev)
    u = torch.empty(L, dtype=torch.int64, device=dev)
    raw = torch.empty(L, dtype=torch.uint8, device=dev)
    mask = torch.empty(L, dtype=torch.uint8, device=dev)
    sim = torch.empty(L, dtype=torch.float32, device=dev) if want_sim else None
    cnt = torch.empty(1, dtype=torch.int32, device=dev)
    rc = _lib.lib().rgn_arp_partition(_p(s), _dt(s), _p(mo), _dt(mo) if mo is not None else F32, _p(c), _dt(c),
                                      float(dt_final), float(threshold), L, D, h_tok, w_tok, int(erosion_dilation),
                                      _p(e), _p(u), _p(raw), _p(mask), _p(sim), _p(cnt), _stream())
    _lib.check(rc, "rgn_arp_partition")
    k = int(cnt.item())
    return e[:k].unsqueeze(0), u[: L - k].unsqueeze(0), mask, raw, sim


def morph_compact(raw_mask: torch.Tensor, erosion_dilation: bool = True):
    h, w = raw_mask.shape
    L = h * w
    dev = raw_mask.device
    e = torch.empty(L, dtype=torch.int64, device=dev)
    u = torch.empty(L, dtype=torch.int64, device=dev)
    mask = torch.empty(L, dtype=torch.uint8, device=dev)
    cnt = torch.empty(1, dtype=torch.int32, device=dev)
    rc = _lib.lib().rgn_morph_compact(_p(raw_mask.contiguous()), h, w, int(erosion_dilation), _p(e), _p(u), _p(mask),
                                      _p(cnt), _stream())
    _lib.check(rc, "rgn_morph_compact")
    k = int(cnt.item())
    return e[:k], u[: L - k], mask.view(h, w)


def gather_rows(x: torch.Tensor, ids: torch.Tensor) -> torch.Tensor:
    """ids_gather for [1,L,D] x, [1,K] ids -> [1,K,D]."""
    src = _rows(x)
    idv = ids.reshape(-1).contiguous()
    K = idv.numel()
    out = torch.empty((K, src.shape[1]), dtype=src.dtype, device=src.device)
    rc = _lib.lib().rgn_gather_rows(_p(src), _p(idv), _p(out), K, src.shape[1] * src.element_size(), _stream())
    _lib.check(rc, "rgn_gather_rows")
    return out.unsqueeze(0) if x.dim() == 3 else out


def scatter_rows_(src: torch.Tensor, ids: torch.Tensor, dst: torch.Tensor) -> torch.Tensor:
    """ids_scatter: dst[ids[k]] = src[k] in place; returns dst."""
    s, d = _rows(src), _rows(dst)
    idv = ids.reshape(-1).contiguous()
    if idv.dtype != torch.int64 or idv.numel() > s.shape[0] or s.shape[1] != d.shape[1] or s.dtype != d.dtype:
        raise _lib.RegionEHipError(f"scatter_rows_: {idv.numel()} {idv.dtype} ids for {s.shape[0]} source rows of width {s.shape[1]} -> {d.shape[1]}")
    rc = _lib.lib().rgn_scatter_rows(_p(s), _p(idv), _p(d), idv.numel(), s.shape[1] * s.element_size(), _stream())
    _lib.check(rc, "rgn_scatter_rows")
    return dst


def euler_step(sample: torch.Tensor, v: torch.Tensor, dt: float, mask: Optional[torch.Tensor] = None,
               dt_direct: float = 0.0) -> torch.Tensor:
    s, vv = _rows(sample), _rows(v)
    if mask is not None and (mask.dtype != torch.uint8 or not mask.is_contiguous() or mask.numel() != s.shape[0]):
        raise _lib.RegionEHipError("euler_step: mask must be uint8 [L] (the third result of arp_partition)")
    out = torch.empty_like(vv)
    rc = _lib.lib().rgn_euler_step(_p(s), _dt(s), _p(vv), _dt(vv), _p(out), _p(mask), float(dt), float(dt_direct),
                                   s.shape[0], s.shape[1], _stream())
    _lib.check(rc, "rgn_euler_step")
    return out.unsqueeze(0) if v.dim() == 3 else out


def avd_apply(cache: torch.Tensor, ratio: float, ids: Optional[torch.Tensor] = None,
              round_ratio: bool = False) -> torch.Tensor:
    c = _rows(cache)
    if ids is not None and ids.dtype != torch.int64:
        raise _lib.RegionEHipError("avd_apply: ids must be int64")        # the kernel reads 8-byte indices
    idv = ids.reshape(-1).contiguous() if ids is not None else None
    K = idv.numel() if idv is not None else c.shape[0]
    out = torch.empty((K, c.shape[1]), dtype=c.dtype, device=c.device)
    rc = _lib.lib().rgn_avd_apply(_p(c), _dt(c), _p(idv), float(ratio), int(round_ratio), _p(out), K, c.shape[1],
                                  _stream())
    _lib.check(rc, "rgn_avd_apply")
    return out.unsqueeze(0) if cache.dim() == 3 else out


CFG_PLAIN, CFG_STEP1X_RESCALE, CFG_QWEN_NORM = 0, 1, 2


def cfg_combine(pos: torch.Tensor, neg: torch.Tensor, scale: float, mode: int = CFG_PLAIN, power: float = 0.4) -> torch.Tensor:
    p, n = _rows(pos), _rows(neg)
    out = torch.empty_like(p)
    rc = _lib.lib().rgn_cfg_combine(_p(p), _p(n), _p(out), _dt(p), float(scale), mode, float(power), p.shape[0],
                                    p.shape[1], _stream())
    _lib.check(rc, "rgn_cfg_combine")
    return out.unsqueeze(0) if pos.dim() == 3 else out


# ---------------------------------------------------------------------------------------------
# MMDiT block kernels
# ---------------------------------------------------------------------------------------------
_gemm_ws = {}
MAX_WS_LANES = 4          # scratch lanes kept per kind (device x stream), least recently used dropped


def _shared_lane(device, kind: int):
    """With the C++ registration active its lane table is the one provider (torch.ops.regione_mi.workspace): the block-body GEMMs
    launched from here share the buffer the fused projections use on the same stream.  None -> the Python table below."""
    import sys
    to = sys.modules.get("regione_amd.torch_ops")
    if to is None or getattr(to, "REGISTRATION", "py") != "cpp":
        return None
    probes = _shared_lane.__dict__.setdefault("probes", {})
    like = probes.get(str(device))
    if like is None:
        like = probes[str(device)] = torch.empty(0, dtype=torch.float32, device=device)
    return torch.ops.regione_mi.workspace(like, kind)


def _ws_lane(table: dict, device, make):
    """One scratch buffer per (device, stream), bounded LRU: two forwards running on two streams must not share split-K /
    KV-split partials, and a host calling from many short-lived streams must not pin 256 + 128 MiB per stream forever.  A
    dropped lane's memory goes back to the caching allocator, which keeps it off limits until the work enqueued on its
    stream has finished."""
    key = (str(device), torch.cuda.current_stream(device).cuda_stream)
    buf = table.pop(key, None)
    if buf is None:
        mine = [k for k in table if k[0] == key[0]]           # the bound is per device
        while len(mine) >= MAX_WS_LANES:
            table.pop(mine.pop(0))
        buf = make()
    table[key] = buf                     # most recently used last
    return buf


def gemm_workspace(device) -> torch.Tensor:
    """fp32 scratch for the round-aware GEMM schedule (one per device and stream, see _ws_lane)."""
    shared = _shared_lane(device, 0)
    if shared is not None:
        return shared
    return _ws_lane(_gemm_ws, device,
                    lambda: torch.empty(_lib.lib().rgn_gemm_workspace_bytes() // 4, dtype=torch.float32, device=device))


FP8 = torch.float8_e4m3fn


def quantize_w8(W: torch.Tensor):
    """bf16 weight [N, K] -> (fp8 e4m3fn [N, K], fp32 scale [N]): per-output-channel absmax / 448 (OCP e4m3fn max).  The
    scale travels ON the quantised tensor (`._rgn_scale`), so call sites keep passing one weight object."""
    w = W.float()
    scale = (w.abs().amax(dim=1).clamp_min(1e-12) / 448.0)
    q = (w / scale[:, None]).to(FP8).contiguous()
    q._rgn_scale = scale.contiguous()
    return q


def wrows(W: torch.Tensor, lo: Optional[int] = None, hi: Optional[int] = None) -> torch.Tensor:
    """W[lo:hi] (output channels) of a GEMM weight; for a quantised weight the per-channel scales are sliced with it (a plain
    slice of the tensor would lose the `_rgn_scale` attribute)."""
    v = W[lo:hi]
    s = getattr(W, "_rgn_scale", None)
    if s is not None:
        v._rgn_scale = s[lo:hi]
    return v


def _wscale(W: torch.Tensor) -> Optional[torch.Tensor]:
    if W.dtype != FP8:
        return None
    s = getattr(W, "_rgn_scale", None)
    if s is None:
        raise _lib.RegionEHipError("fp8 weight without its per-channel scale (use ops.quantize_w8; slices lose the attribute)")
    return s


def gemm(A: torch.Tensor, W: torch.Tensor, bias: Optional[torch.Tensor], out: torch.Tensor, *, epilogue: int = EPI_BIAS,
         gelu_from_col: int = 0, gate: Optional[torch.Tensor] = None, resid: Optional[torch.Tensor] = None,
         out_rows: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[r] = epilogue(A @ W^T + bias).  A [M,K] (row stride may exceed K), W [N,K] (bf16, or fp8 from quantize_w8), out [*,N] view."""
    assert A.dtype == out.dtype == torch.bfloat16 and W.dtype in (torch.bfloat16, FP8)
    M, K = A.shape
    N = W.shape[0]
    assert A.stride(1) == 1 and W.stride(1) == 1 and out.stride(1) == 1 and W.shape[1] == K and out.shape[1] == N
    if resid is not None:
        assert resid.stride(0) == out.stride(0) and resid.stride(1) == 1
    ws = gemm_workspace(A.device)
    sc = _wscale(W)
    if sc is not None:
        rc = _lib.lib().rgn_gemm_w8(_p(A), A.stride(0), _p(W), W.stride(0), _p(sc), _p(bias), _p(out), out.stride(0), M, N, K,
                                    epilogue, gelu_from_col, _p(gate), _p(resid), _p(out_rows), _p(ws), ws.numel() * 4, _stream())
        _lib.check(rc, "rgn_gemm_w8")
        return out
    rc = _lib.lib().rgn_gemm_bf16(_p(A), A.stride(0), _p(W), W.stride(0), _p(bias), _p(out), out.stride(0), M, N, K,
                                  epilogue, gelu_from_col, _p(gate), _p(resid), _p(out_rows), _p(ws), ws.numel() * 4,
                                  _stream())
    _lib.check(rc, "rgn_gemm_bf16")
    return out


def gemm_pair(A0, W0, b0, out0, A1, W1, b1, out1, *, epilogue: int = EPI_BIAS, gelu_from_col: int = 0,
              gate0=None, resid0=None, gate1=None, resid1=None):
    """Two GEMMs with equal (N, K) and epilogue in one launch: outI = epilogue(AI @ WI^T + bI)."""
    N, K = W0.shape
    assert W1.shape == (N, K) and W0.is_contiguous() and W1.is_contiguous()
    assert A0.shape[1] == K and A1.shape[1] == K and out0.shape[1] == N and out1.shape[1] == N
    for t in (A0, A1, out0, out1):
        assert t.stride(1) == 1 and t.dtype == torch.bfloat16
    ws = gemm_workspace(A0.device)
    s0, s1 = _wscale(W0), _wscale(W1)
    if s0 is not None or s1 is not None:
        rc = _lib.lib().rgn_gemm_w8_pair(_p(A0), A0.stride(0), _p(W0), _p(s0), _p(b0), _p(out0), out0.stride(0), A0.shape[0],
                                         _p(gate0), _p(resid0), _p(A1), A1.stride(0), _p(W1), _p(s1), _p(b1), _p(out1),
                                         out1.stride(0), A1.shape[0], _p(gate1), _p(resid1), N, K, epilogue, gelu_from_col,
                                         _p(ws), ws.numel() * 4, _stream())
        _lib.check(rc, "rgn_gemm_w8_pair")
        return
    rc = _lib.lib().rgn_gemm_bf16_pair(_p(A0), A0.stride(0), _p(W0), _p(b0), _p(out0), out0.stride(0), A0.shape[0],
                                       _p(gate0), _p(resid0), _p(A1), A1.stride(0), _p(W1), _p(b1), _p(out1),
                                       out1.stride(0), A1.shape[0], _p(gate1), _p(resid1), N, K, epilogue,
                                       gelu_from_col, _p(ws), ws.numel() * 4, _stream())
    _lib.check(rc, "rgn_gemm_bf16_pair")


def _check_bias(bias, N: int, dev, what: str):
    """The kernels read `bias` as bf16 [N] (8-byte vectors): the C ABI cannot know its length."""
    if bias is not None and (bias.dtype != torch.bfloat16 or bias.dim() != 1 or bias.numel() != N or not bias.is_contiguous() or bias.device != dev):
        raise _lib.RegionEHipError(f"{what}: bias must be bf16 [{N}], contiguous, on {dev} (got {bias.dtype} {tuple(bias.shape)} on {bias.device})")


def qkv_epilogue(*, wq, wk, rope_q, rope_k, k_slab, vt_slab, H: int, k_col: int, v_col: int, q_col: int,
                 kv_rows: Optional[torch.Tensor] = None, row_base: int = 0, eps: float = 1e-6, fp16_roundtrip: bool = False,
                 rows: Optional[int] = None):
    """Descriptor of the fused Q/K/V epilogue (struct rgn_qkv_epilogue); keeps its tensors alive.  `rows` = the row count M of
    the problem the descriptor belongs to: with it the extents the kernel will read are checked here (the same checks as
    csrc/torch_binding.cpp:epi) - rotary rows [row_base, row_base + M), kv_rows[row_base + m], identity cache rows inside the slab."""
    skv_pad = k_slab.shape[0]
    assert vt_slab.shape == (H * 128, skv_pad) and k_slab.shape[1] == H * 128 and k_slab.is_contiguous() and vt_slab.is_contiguous()
    for t in (rope_q[0], rope_q[1], rope_k[0], rope_k[1]):
        assert t.dtype == torch.float32 and t.shape[1] == 128 and t.is_contiguous()
    assert rope_q[0].shape == rope_q[1].shape and rope_k[0].shape == rope_k[1].shape, "cos / sin of one rotary table differ in shape"
    for w in (wq, wk):
        assert w.dtype == torch.bfloat16 and w.numel() == 128 and w.is_contiguous(), "per-head RMSNorm weights: bf16 [128]"
    # the kernel reads 8-byte indices: an int32 tensor would be read past its end and scatter K / V to garbage cache rows
    assert kv_rows is None or (kv_rows.dtype == torch.int64 and kv_rows.dim() == 1 and kv_rows.is_contiguous())
    dev = k_slab.device
    for t in (wq, wk, rope_q[0], rope_q[1], rope_k[0], rope_k[1], vt_slab) + ((kv_rows,) if kv_rows is not None else ()):
        if t.device != dev:
            raise _lib.RegionEHipError(f"fused Q/K/V epilogue: tensors on {t.device} and {dev} (one device per problem)")
    if rows is not None:
        need = row_base + int(rows)
        if row_base < 0 or rope_q[0].shape[0] < need:
            raise _lib.RegionEHipError(f"rotary table of the queries has {rope_q[0].shape[0]} rows, the problem needs {need}")
        if kv_rows is not None:
            if kv_rows.numel() < need:
                raise _lib.RegionEHipError(f"kv_rows has {kv_rows.numel()} entries, the problem needs {need}")
            # the VALUES of kv_rows are device data, unchecked (no host sync per launch): the caller guarantees max(kv_rows) <
            # min(rope_k rows, skv_pad); checked is the necessary condition (distinct scatter targets need that many rows)
            if rope_k[0].shape[0] < need or skv_pad < need:
                raise _lib.RegionEHipError(f"key rotary table ({rope_k[0].shape[0]} rows) / slab ({skv_pad} rows) cannot hold the {need} "
                                           "distinct cache rows kv_rows names")
        elif rope_k[0].shape[0] < need or need > skv_pad:
            raise _lib.RegionEHipError(f"identity cache rows [{row_base}, {need}) exceed the key rotary table / the slab's {skv_pad} rows")
    e = _lib.QkvEpilogue(_p(wq), _p(wk), _p(rope_q[0]), _p(rope_q[1]), _p(rope_k[0]), _p(rope_k[1]), _p(kv_rows),
                         _p(k_slab), _p(vt_slab), row_base, skv_pad, k_col, v_col, q_col, H, eps, int(fp16_roundtrip))
    e._keep = (wq, wk, rope_q, rope_k, kv_rows, k_slab, vt_slab)
    return e


def gemm_qkv(A: torch.Tensor, W: torch.Tensor, bias: Optional[torch.Tensor], out: torch.Tensor, epi, *,
             gelu_from_col: Optional[int] = None) -> torch.Tensor:
    """QKV (+ fused MLP half) projection whose epilogue normalises / rotates Q and K, places K and V^T in the
    cache slabs and leaves Q (and GELU(mlp)) in `out`: rgn_gemm_bf16 + rgn_qk_norm_rope_store in one launch."""
    assert A.dtype == out.dtype == torch.bfloat16 and W.dtype in (torch.bfloat16, FP8)
    M, K = A.shape
    N = W.shape[0]
    assert A.stride(1) == 1 and W.stride(1) == 1 and out.stride(1) == 1 and W.shape[1] == K and out.shape[1] == N
    _check_bias(bias, N, A.device, "gemm_qkv")
    ws = gemm_workspace(A.device)
    sc = _wscale(W)
    if sc is not None:
        rc = _lib.lib().rgn_gemm_w8_qkv(_p(A), A.stride(0), _p(W), W.stride(0), _p(sc), _p(bias), _p(out), out.stride(0), M, N, K,
                                        N if gelu_from_col is None else gelu_from_col, epi, _p(ws), ws.numel() * 4, _stream())
        _lib.check(rc, "rgn_gemm_w8_qkv")
        return out
    rc = _lib.lib().rgn_gemm_bf16_qkv(_p(A), A.stride(0), _p(W), W.stride(0), _p(bias), _p(out), out.stride(0), M, N, K,
                                      N if gelu_from_col is None else gelu_from_col, epi, _p(ws), ws.numel() * 4, _stream())
    _lib.check(rc, "rgn_gemm_bf16_qkv")
    return out


def gemm_qkv_pair(A0, W0, b0, out0, epi0, A1, W1, b1, out1, epi1):
    """Both streams of a double block (image + text QKV projections) in one launch, fused Q/K/V epilogue."""
    N, K = W0.shape
    assert W1.shape == (N, K) and W0.is_contiguous() and W1.is_contiguous()
    assert A0.shape[1] == K and A1.shape[1] == K and out0.shape[1] == N and out1.shape[1] == N
    for t in (A0, A1, out0, out1):
        assert t.stride(1) == 1 and t.dtype == torch.bfloat16
    _check_bias(b0, N, A0.device, "gemm_qkv_pair")
    _check_bias(b1, N, A0.device, "gemm_qkv_pair")
    ws = gemm_workspace(A0.device)
    s0, s1 = _wscale(W0), _wscale(W1)
    if s0 is not None or s1 is not None:
        rc = _lib.lib().rgn_gemm_w8_qkv_pair(_p(A0), A0.stride(0), _p(W0), _p(s0), _p(b0), _p(out0), out0.stride(0), A0.shape[0],
                                             epi0, _p(A1), A1.stride(0), _p(W1), _p(s1), _p(b1), _p(out1), out1.stride(0),
                                             A1.shape[0], epi1, N, K, _p(ws), ws.numel() * 4, _stream())
        _lib.check(rc, "rgn_gemm_w8_qkv_pair")
        return
    rc = _lib.lib().rgn_gemm_bf16_qkv_pair(_p(A0), A0.stride(0), _p(W0), _p(b0), _p(out0), out0.stride(0), A0.shape[0], epi0,
                                           _p(A1), A1.stride(0), _p(W1), _p(b1), _p(out1), out1.stride(0), A1.shape[0], epi1,
                                           N, K, _p(ws), ws.numel() * 4, _stream())
    _lib.check(rc, "rgn_gemm_bf16_qkv_pair")


class Problem:
    """One problem of `gemm_group`: out = epilogue(A @ W^T + bias) with its own gate / residual or Q/K/V epilogue."""
    __slots__ = ("A", "W", "bias", "out", "gate", "resid", "epi")

    def __init__(self, A, W, bias, out, gate=None, resid=None, epi=None):
        self.A, self.W, self.bias, self.out, self.gate, self.resid, self.epi = A, W, bias, out, gate, resid, epi


def gemm_group(problems, *, epilogue: int = EPI_BIAS, gelu_from_col: int = 0):
    """Up to four GEMMs with equal (N, K), epilogue and weight format in ONE launch (rgn_gemm_group): the text / image
    streams of a double block x the cond / uncond CFG branches.  Problems may share W."""
    import ctypes as C
    probs = [p for p in problems if p.A.shape[0] > 0]
    if not probs:
        return
    assert len(probs) <= 4
    N, K = probs[0].W.shape
    arr = (_lib.GemmProblem * len(probs))()
    for i, p in enumerate(probs):
        assert p.W.shape == (N, K) and p.W.is_contiguous() and p.A.shape[1] == K and p.out.shape == (p.A.shape[0], N)
        for t in (p.A, p.out):
            assert t.stride(1) == 1 and t.dtype == torch.bfloat16
        if p.resid is not None:
            assert p.resid.stride(0) == p.out.stride(0) and p.resid.stride(1) == 1
        _check_bias(p.bias, N, p.A.device, "gemm_group")
        g = arr[i]
        g.A, g.W, g.wscale, g.bias, g.C = _p(p.A), _p(p.W), _p(_wscale(p.W)), _p(p.bias), _p(p.out)
        g.gate, g.resid = _p(p.gate), _p(p.resid)
        g.qkv = C.pointer(p.epi) if p.epi is not None else None
        g.lda, g.ldc, g.M = p.A.stride(0), p.out.stride(0), p.A.shape[0]
    ws = gemm_workspace(probs[0].A.device)
    rc = _lib.lib().rgn_gemm_group(arr, len(probs), N, K, epilogue, gelu_from_col, _p(ws), ws.numel() * 4, _stream())
    _lib.check(rc, "rgn_gemm_group")


def gemv(x: torch.Tensor, W: torch.Tensor, bias: Optional[torch.Tensor], silu_input: bool = False,
         out: Optional[torch.Tensor] = None) -> torch.Tensor:
    B, K = x.shape
    N = W.shape[0]
    assert W.is_contiguous() and x.stride(1) == 1
    if out is None:
        out = torch.empty((B, N), dtype=torch.bfloat16, device=x.device)
    rc = _lib.lib().rgn_gemv_bf16(_p(x), x.stride(0), _p(W), _p(bias), _p(out), out.stride(0), B, N, K,
                                  int(silu_input), _stream())
    _lib.check(rc, "rgn_gemv_bf16")
    return out


def rms_norm_rows(x: torch.Tensor, w: torch.Tensor, eps: float = 1e-6) -> torch.Tensor:
    out = torch.empty_like(x)
    rc = _lib.lib().rgn_rms_norm_rows(_p(x), x.stride(0), _p(w), _p(out), out.stride(0), x.shape[0], x.shape[1], eps,
                                      _stream())
    _lib.check(rc, "rgn_rms_norm_rows")
    return out


def silu(x: torch.Tensor) -> torch.Tensor:
    assert x.dtype == torch.bfloat16 and x.is_contiguous()
    y = torch.empty_like(x)
    _lib.check(_lib.lib().rgn_silu_bf16(_p(x), _p(y), x.numel(), _stream()), "rgn_silu_bf16")
    return y


def add_bf16(a: torch.Tensor, b: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """bf16 a + b (one rounding, like torch) into `out` (default: a new tensor; may alias an input)."""
    assert a.dtype == b.dtype == torch.bfloat16 and a.shape == b.shape and a.is_contiguous() and b.is_contiguous()
    if out is None:
        out = torch.empty_like(a)
    assert out.dtype == torch.bfloat16 and out.numel() == a.numel() and out.is_contiguous()
    _lib.check(_lib.lib().rgn_add_bf16(_p(a), _p(b), _p(out), a.numel(), _stream()), "rgn_add_bf16")
    return out


def sel_rows(edited_ids: torch.Tensor, T: int) -> torch.Tensor:
    """int64 [T + K]: [0..T) then T + edited_ids - the cache rows of a region step (inplace.py:732-733)."""
    idv = edited_ids.reshape(-1)
    assert idv.dtype == torch.int64 and idv.is_contiguous()
    out = torch.empty(T + idv.numel(), dtype=torch.int64, device=idv.device)
    _lib.check(_lib.lib().rgn_sel_rows(_p(idv), idv.numel(), int(T), _p(out), _stream()), "rgn_sel_rows")
    return out


def zeros(shape, dtype=torch.bfloat16, device="cuda") -> torch.Tensor:
    """torch.empty + hipMemsetAsync on the current stream (no fill kernel)."""
    t = torch.empty(shape, dtype=dtype, device=device)
    _lib.check(_lib.lib().rgn_fill_zero(_p(t), t.numel() * t.element_size(), _stream()), "rgn_fill_zero")
    return t


def cat_rows(parts, dim: int = 1) -> torch.Tensor:
    """torch.cat of contiguous pieces along `dim` when every piece is one contiguous run of the result (dim 0, or dim 1 of
    batch-1 tensors): device-to-device copies (hipMemcpyAsync), no concat kernel."""
    parts = list(parts)
    ref = parts[0]
    assert all(p.is_contiguous() and p.dtype == ref.dtype for p in parts) and (dim == 0 or all(p.shape[0] == 1 for p in parts))
    shape = list(ref.shape)
    shape[dim] = sum(p.shape[dim] for p in parts)
    out = torch.empty(shape, dtype=ref.dtype, device=ref.device)
    o = 0
    for p in parts:
        n = p.shape[dim]
        out.narrow(dim, o, n).copy_(p)
        o += n
    return out


def ln_modulate(x: torch.Tensor, out: torch.Tensor, shift1, scale1, split_row: int = 0, shift0=None, scale0=None,
                eps: float = 1e-6) -> torch.Tensor:
    M, d = x.shape
    rc = _lib.lib().rgn_ln_modulate(_p(x), x.stride(0), _p(out), out.stride(0), M, d, eps, split_row, _p(shift0),
                                    _p(scale0), _p(shift1), _p(scale1), _stream())
    _lib.check(rc, "rgn_ln_modulate")
    return out


def ln_modulate_segs(x: torch.Tensor, out: torch.Tensor, segs, eps: float = 1e-6) -> torch.Tensor:
    """LN * (1 + scale) + shift with up to four row segments: `segs` = [(end_row, shift, scale), ...], ascending, the last
    end_row == rows of x."""
    import ctypes as C
    M, d = x.shape
    n = len(segs)
    assert 1 <= n <= 4 and segs[-1][0] == M
    ends = (C.c_int * n)(*[int(s[0]) for s in segs])
    sh = (C.c_void_p * n)(*[_p(s[1]) for s in segs])
    sc = (C.c_void_p * n)(*[_p(s[2]) for s in segs])
    rc = _lib.lib().rgn_ln_modulate_segs(_p(x), x.stride(0), _p(out), out.stride(0), M, d, eps, n, ends, sh, sc, _stream())
    _lib.check(rc, "rgn_ln_modulate_segs")
    return out


def qk_norm_rope_store(qkv: torch.Tensor, k_col: int, v_col: int, q_col: int, H: int, wq1, wk1, rope_q, rope_k,
                       k_slab: torch.Tensor, vt_slab: torch.Tensor, kv_rows: Optional[torch.Tensor] = None,
                       split_row: int = 0, wq0=None, wk0=None, eps: float = 1e-6):
    M = qkv.shape[0]
    skv_pad = k_slab.shape[0]
    assert vt_slab.shape == (H * 128, skv_pad) and k_slab.shape[1] == H * 128
    assert kv_rows is None or (kv_rows.dtype == torch.int64 and kv_rows.dim() == 1 and kv_rows.is_contiguous() and kv_rows.numel() >= M)
    rc = _lib.lib().rgn_qk_norm_rope_store(_p(qkv), qkv.stride(0), k_col, v_col, q_col, M, H, split_row, _p(wq0),
                                           _p(wk0), _p(wq1), _p(wk1), eps, _p(rope_q[0]), _p(rope_q[1]),
                                           _p(rope_k[0]), _p(rope_k[1]), _p(kv_rows), _p(k_slab), _p(vt_slab), skv_pad,
                                           _stream())
    _lib.check(rc, "rgn_qk_norm_rope_store")


_attn_ws = {}


def attention_workspace(device) -> torch.Tensor:
    """fp32 scratch for the round-aware attention schedule (allocated once per device and stream)."""
    shared = _shared_lane(device, 1)
    if shared is not None:
        return shared
    return _ws_lane(_attn_ws, device,
                    lambda: torch.empty(_lib.lib().rgn_attention_workspace_bytes(0, 0) // 4, dtype=torch.float32, device=device))


def attention(q: torch.Tensor, k_slab: torch.Tensor, vt_slab: torch.Tensor, out: torch.Tensor, skv: int, H: int,
              scale: Optional[float] = None, workspace: Optional[torch.Tensor] = None, score_bound: float = 0.0) -> torch.Tensor:
    """q / out: [Sq, H*128] views (row stride free, may alias).  `score_bound` > 0: the caller's guarantee that every
    |q . k| * scale is at most that (rgn_attention_bounded: no running max in the softmax)."""
    Sq = q.shape[0]
    if scale is None:
        scale = 1.0 / math.sqrt(128.0)
    if workspace is None:
        workspace = attention_workspace(q.device)
    rc = _lib.lib().rgn_attention_bounded(_p(q), q.stride(0), _p(k_slab), _p(vt_slab), k_slab.shape[0], _p(out), out.stride(0),
                                          Sq, skv, H, float(scale), float(score_bound), _p(workspace), workspace.numel() * 4,
                                          _stream())
    _lib.check(rc, "rgn_attention_bounded")
    return out


def device_info() -> Tuple[int, int, int]:
    import ctypes as C
    cu, clk, mem = C.c_int(), C.c_int(), C.c_size_t()
    _lib.check(_lib.lib().rgn_device_info(C.byref(cu), C.byref(clk), C.byref(mem)), "rgn_device_info")
    return cu.value, clk.value, mem.value
