"""-m gpu: seeded random-shape sweeps of the MMDiT kernels through the C ABI - the fixed-shape tests of test_gpu_kernels.py cover
the shapes the four model families launch; these cover what lies between them (ragged M / N, row strides, every epilogue, every
planner outcome: whole rounds, split-K remainders, quarter tiles, 128 / 256 tile geometries, grouped problems; attention with
ragged query counts, KV lengths that are / are not whole tiles, KV splits and stream-K remainders).  References: fp64 matmul /
softmax on the GPU (torch), tolerance = one bf16 rounding of the result + fp32 accumulation noise, stated per check."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def bf(x):
    return x.to(torch.bfloat16)


def _check(out, ref, what, atol=1e-2, ulp=2 ** -8, extra=None):
    """|out - ref| <= ulp * |ref| + atol (+ extra) elementwise (ref in fp64: one bf16 rounding + accumulation noise of O(sqrt(K))
    products; `extra`: an elementwise allowance the caller derives, e.g. the gate-amplified rounding of the linear result)"""
    err = (out.double() - ref).abs()
    tol = ulp * ref.abs() + atol
    if extra is not None:
        tol = tol + extra
    bad = err > tol
    assert not bool(bad.any()), f"{what}: {int(bad.sum())} elements off, worst {float((err / tol).max()):.2f} x tolerance"


def _rand_gemm_case(rng, big):
    K = 64 * int(rng.integers(1, 9 if not big else 49))
    M = int(rng.integers(1, 700 if not big else 2600))
    N = int(rng.integers(1, 80 if not big else 420)) * 8
    return M, N, K


@pytest.mark.parametrize("seed", range(16))
def test_gemm_random_shapes_strides_and_epilogues(seed):
    import numpy as np
    from regione_amd import ops
    rng = np.random.default_rng(1000 + seed)
    g = torch.Generator(device="cuda").manual_seed(seed)
    for case in range(14):
        M, N, K = _rand_gemm_case(rng, big=(case % 3 == 2))
        lda = K + 8 * int(rng.integers(0, 5))
        ldc = N + 8 * int(rng.integers(0, 5))
        Abuf = bf(torch.randn(M, lda, generator=g, device="cuda"))
        A = Abuf[:, lda - K:] if lda - K and (lda - K) % 8 == 0 else Abuf[:, :K]
        W = bf(torch.randn(N, K, generator=g, device="cuda") * 0.05)
        b = bf(torch.randn(N, generator=g, device="cuda")) if rng.integers(0, 4) else None
        lin = A.double() @ W.double().T + (b.double() if b is not None else 0.0)
        epi = int(rng.integers(0, 4))
        Cbuf = torch.full((M, ldc), 7.0, dtype=torch.bfloat16, device="cuda")
        C = Cbuf[:, :N]
        tag = f"seed {seed} case {case}: M={M} N={N} K={K} lda={A.stride(0)} ldc={ldc} epi={epi} bias={b is not None}"
        if epi == 0:
            ops.gemm(A, W, b, C)
            _check(C, lin, tag)
        elif epi == 1:
            col = 8 * int(rng.integers(0, N // 8 + 1))
            ops.gemm(A, W, b, C, epilogue=ops.EPI_GELU, gelu_from_col=col)
            ref = bf(lin).double()                                   # the epilogue rounds the linear result to bf16 first
            ref[:, col:] = F.gelu(ref[:, col:], approximate="tanh")
            _check(C, ref, tag + f" gelu_from={col}", atol=2e-2, ulp=2 ** -7)
        elif epi == 2:
            gate = bf(torch.randn(N, generator=g, device="cuda"))
            Cbuf[:, :N] = bf(torch.randn(M, N, generator=g, device="cuda"))
            r0 = C.clone()
            ops.gemm(A, W, b, C, epilogue=ops.EPI_GATE_RESID, gate=gate, resid=C)
            ref = r0.double() + bf(gate.double().unsqueeze(0) * bf(lin).double()).double()
            _check(C, ref, tag, atol=3e-2, ulp=2 ** -7)
        else:
            rows = int(rng.integers(M, M + 300))
            cache = bf(torch.randn(rows, ldc, generator=g, device="cuda"))
            c0 = cache.clone()
            idx = torch.randperm(rows, generator=g, device="cuda")[:M].sort().values
            ops.gemm(A, W, b, cache[:, :N], out_rows=idx)
            _check(cache[idx][:, :N], lin, tag + " scatter")
            keep = torch.ones(rows, dtype=torch.bool, device="cuda")
            keep[idx] = False
            assert torch.equal(cache[keep], c0[keep]) and torch.equal(cache[:, N:], c0[:, N:]), tag + ": wrote outside its rows / columns"
        if epi != 3:
            assert bool((Cbuf[:, N:] == 7.0).all()), tag + ": wrote past column N"


@pytest.mark.parametrize("split", ["0", None])
@pytest.mark.parametrize("seed", range(6))
def test_gemm_random_groups_match_separate_launches(seed, split, monkeypatch):
    """rgn_gemm_group: 2 - 4 problems (random row counts incl. tiny ones, shared or own weights) against one launch per problem:
    bit-identical without the split-K schedule (gemm_pieces = 1: tile-local arithmetic does not depend on the grouping); with it
    the merged launch may cut a remainder into a different number of K pieces than the single launch does (fp32 summation order),
    so the two agree to one bf16 rounding and both sit within tolerance of the fp64 reference."""
    import numpy as np
    from regione_amd import ops
    if split is not None:
        ops._lib.lib().rgn_plan_override(b"gemm_pieces", 1)          # reset after the test (conftest)
    rng = np.random.default_rng(2000 + seed)
    g = torch.Generator(device="cuda").manual_seed(100 + seed)
    for case in range(6):
        K = 64 * int(rng.integers(2, 49))
        N = 8 * int(rng.integers(8, 400))
        n = int(rng.integers(2, 5))
        Ws = [bf(torch.randn(N, K, generator=g, device="cuda") * 0.05) for _ in range(2)]
        probs, singles = [], []
        epi = int(rng.integers(0, 3))
        for i in range(n):
            M = int(rng.integers(1, 1800))
            A = bf(torch.randn(M, K, generator=g, device="cuda"))
            W = Ws[int(rng.integers(0, 2))]
            b = bf(torch.randn(N, generator=g, device="cuda"))
            gate = bf(torch.randn(N, generator=g, device="cuda")) if epi == 2 else None
            r = bf(torch.randn(M, N, generator=g, device="cuda"))
            o_g, o_s = r.clone(), r.clone()
            probs.append(ops.Problem(A, W, b, o_g, gate=gate, resid=o_g if epi == 2 else None))
            singles.append((A, W, b, o_s, gate))
        kw = dict(epilogue=(ops.EPI_BIAS, ops.EPI_GELU, ops.EPI_GATE_RESID)[epi], gelu_from_col=8 * int(rng.integers(0, N // 8)))
        ops.gemm_group(probs, **kw)
        for (A, W, b, o_s, gate), p in zip(singles, probs):
            ops.gemm(A, W, b, o_s, gate=gate, resid=o_s if epi == 2 else None, **kw)
            tag = f"seed {seed} case {case}: group vs single launch (M={A.shape[0]} N={N} K={K} epi={epi})"
            if split == "0":
                assert torch.equal(p.out, o_s), tag
            else:
                # the two launches may sum K in a different order (another piece count / geometry): bf16(lin) can differ by ONE
                # rounding; the gated-residual epilogue multiplies that by the gate and rounds the product to bf16 again before adding
                # the residual - two roundings of size 2^-8 |gate * lin|
                extra = None
                if epi == 2:
                    lin = (A.double() @ W.double().T + b.double()).abs()
                    extra = gate.double().abs()[None, :] * lin * 2.0 ** -7
                _check(p.out, o_s.double(), tag, atol=2e-2, ulp=2 ** -7, extra=extra)


@pytest.mark.parametrize("seed", range(6))
def test_gemm_fp8_weights_random_shapes(seed):
    """fp8 (e4m3fn) weights + per-channel scale: == the bf16 kernel on the de-quantised weights up to the scale's rounding point
    (scale applied to the fp32 accumulator vs folded into W), within one bf16 ulp of the fp64 reference."""
    import numpy as np
    from regione_amd import ops
    rng = np.random.default_rng(3000 + seed)
    g = torch.Generator(device="cuda").manual_seed(200 + seed)
    for case in range(8):
        M, N, K = _rand_gemm_case(rng, big=(case % 2 == 1))
        A = bf(torch.randn(M, K, generator=g, device="cuda"))
        W8 = ops.quantize_w8(bf(torch.randn(N, K, generator=g, device="cuda") * 0.05))
        sc = ops._wscale(W8)
        b = bf(torch.randn(N, generator=g, device="cuda"))
        out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
        ops.gemm(A, W8, b, out)
        ref = (A.double() @ W8.float().double().T) * sc.double().unsqueeze(0) + b.double()
        _check(out, ref, f"seed {seed} case {case}: fp8 M={M} N={N} K={K}")


@pytest.mark.parametrize("seed", range(10))
def test_attention_random_shapes(seed):
    import numpy as np
    from regione_amd import ops
    rng = np.random.default_rng(4000 + seed)
    g = torch.Generator(device="cuda").manual_seed(300 + seed)
    for case in range(7):
        H = int(rng.integers(1, 25))
        Sq = int(rng.integers(1, 3000 if H <= 8 else 1300))
        whole = bool(rng.integers(0, 2))
        Skv = 64 * int(rng.integers(1, 140)) if whole else int(rng.integers(1, 6000))
        D = H * 128
        q = bf(torch.randn(Sq, D, generator=g, device="cuda"))
        k = bf(torch.randn(Skv, D, generator=g, device="cuda"))
        v = bf(torch.randn(Skv, D, generator=g, device="cuda"))
        skv_pad = ops.padded(Skv)
        # K / V^T slabs through the product's own store kernel (unit norm weights, zero rotation: K passes through the per-head
        # RMSNorm only), garbage behind Skv: the slabs are reused between edits and rows past Skv must not leak into the result
        kc = torch.full((skv_pad, D), 50.0, dtype=torch.bfloat16, device="cuda")
        vt = torch.full((D, skv_pad), 50.0, dtype=torch.bfloat16, device="cuda")
        qkv = torch.cat([k, v, k], dim=1).contiguous()        # [K | V | (unused Q columns)] rows of the KV sequence
        ones = torch.ones(128, dtype=torch.bfloat16, device="cuda")
        cos, sin = torch.ones(Skv, 128, device="cuda"), torch.zeros(Skv, 128, device="cuda")
        ops.qk_norm_rope_store(qkv, 0, D, 2 * D, H, ones, ones, (cos, sin), (cos, sin), kc, vt)
        k = kc[:Skv].clone()                              # the (normalised) K rows the kernel reads
        out = torch.empty(Sq, D, dtype=torch.bfloat16, device="cuda")
        scale = 1.0 / math.sqrt(128)
        ops.attention(q, kc, vt, out, Skv, H, scale=scale)
        p = torch.softmax((q.double().view(Sq, H, 128).transpose(0, 1) @ k.double().view(Skv, H, 128).permute(1, 2, 0)) * scale, dim=-1)
        ref = (p @ v.double().view(Skv, H, 128).transpose(0, 1)).transpose(0, 1).reshape(Sq, D)
        err = (out.double() - ref).abs()
        # P is rounded to bf16 before the PV product (2^-9 relative per term, averaging down over Skv terms) + one bf16 rounding of O
        tol = 2 ** -7 * ref.abs() + 6e-3
        assert bool((err <= tol).all()), f"seed {seed} case {case}: Sq={Sq} Skv={Skv} H={H}: worst {float((err / tol).max()):.2f} x tolerance"
        assert bool(torch.isfinite(out.float()).all())


@pytest.mark.parametrize("seed", range(8))
def test_gemm_fused_qkv_epilogue_random_shapes_bit_identical_to_separate_kernels(seed):
    """rgn_gemm_bf16_qkv == rgn_gemm_bf16 + rgn_qk_norm_rope_store bit for bit at random row counts, head counts, K, MLP widths,
    identity / gathered cache rows and joint-sequence offsets (any residue mod 16: the V^T store has an aligned fast path)."""
    import numpy as np
    from regione_amd import ops
    rng = np.random.default_rng(5000 + seed)
    gen = torch.Generator(device="cuda").manual_seed(400 + seed)
    for case in range(5):
        M, H, K = int(rng.integers(1, 900)), 2 * int(rng.integers(1, 4)), 64 * int(rng.integers(1, 13))   # column blocks are 256-aligned: even head counts
        mlp = 8 * int(rng.integers(0, 300)) * int(rng.integers(0, 2))
        gather, row_base = bool(rng.integers(0, 2)), int(rng.integers(0, 40)) * int(rng.integers(0, 2))
        D = H * 128
        N = 3 * D + mlp
        skv = row_base + (M if not gather else M + int(rng.integers(1, 2 * M + 2)))
        A = bf(torch.randn(M, K, generator=gen, device="cuda"))
        W = bf(0.05 * torch.randn(N, K, generator=gen, device="cuda"))
        b = bf(0.1 * torch.randn(N, generator=gen, device="cuda"))
        wq = bf(1 + 0.1 * torch.randn(128, generator=gen, device="cuda"))
        wk = bf(1 + 0.1 * torch.randn(128, generator=gen, device="cuda"))
        ang = torch.rand(skv, 64, generator=gen, device="cuda") * 6.28
        rope = (torch.repeat_interleave(torch.cos(ang), 2, dim=1).contiguous(), torch.repeat_interleave(torch.sin(ang), 2, dim=1).contiguous())
        kv_rows = None
        if gather:
            pick = torch.randperm(skv - row_base, generator=gen, device="cuda")[:M].sort().values
            kv_rows = torch.cat([torch.arange(row_base, device="cuda"), row_base + pick])
        rope_q = rope if kv_rows is None else (rope[0][kv_rows].contiguous(), rope[1][kv_rows].contiguous())
        skv_pad = ops.padded(skv)
        tag = f"seed {seed} case {case}: M={M} H={H} K={K} mlp={mlp} gather={gather} row_base={row_base} skv={skv}"

        def slabs():
            return (torch.zeros(skv_pad, D, dtype=torch.bfloat16, device="cuda"), torch.zeros(D, skv_pad, dtype=torch.bfloat16, device="cuda"))
        wide = torch.zeros(row_base + M, N, dtype=torch.bfloat16, device="cuda")
        ops.gemm(A, W, b, wide[row_base:], epilogue=ops.EPI_GELU, gelu_from_col=3 * D)
        k0, v0 = slabs()
        ops.qk_norm_rope_store(wide, 0, D, 2 * D, H, wq, wk, rope_q, rope, k0, v0, kv_rows)
        out = torch.zeros(row_base + M, N, dtype=torch.bfloat16, device="cuda")
        k1, v1 = slabs()
        epi = ops.qkv_epilogue(wq=wq, wk=wk, rope_q=rope_q, rope_k=rope, k_slab=k1, vt_slab=v1, H=H, k_col=0, v_col=D, q_col=2 * D,
                               kv_rows=kv_rows, row_base=row_base)
        ops.gemm_qkv(A, W, b, out[row_base:], epi, gelu_from_col=3 * D)
        assert torch.equal(out[row_base:, 2 * D:], wide[row_base:, 2 * D:]), tag + ": Q / GELU(mlp)"
        rows = kv_rows[row_base:] if kv_rows is not None else torch.arange(row_base, row_base + M, device="cuda")
        assert torch.equal(k1[rows], k0[rows]), tag + ": K slab"
        pos = (rows & ~12) | ((rows & 4) << 1) | ((rows & 8) >> 1)
        assert torch.equal(v1[:, pos], v0[:, pos]), tag + ": V^T slab"
        k1[rows] = 0
        v1[:, pos] = 0
        assert not k1.any() and not v1.any(), tag + ": wrote cache rows of another problem"
        assert not out[:, :2 * D].any() and not out[:row_base].any(), tag + ": K / V columns written to C"


@pytest.mark.parametrize("seed", range(4))
def test_ln_modulate_segments_random(seed):
    """rgn_ln_modulate_segs: 1 - 4 row segments with their own shift / scale vectors, any width that is a multiple of 8, against
    the eager bf16 sequence `LayerNorm(x) * (1 + scale) + shift` of the oracle (<= 2 bf16 ulp at the largest magnitude, almost
    always bit-identical - the bar of test_ln_modulate)."""
    import numpy as np
    from oracle import regione_oracle as O
    from regione_amd import ops
    rng = np.random.default_rng(6000 + seed)
    gen = torch.Generator().manual_seed(500 + seed)
    for case in range(6):
        d = 8 * int(rng.integers(1, 600))
        n = int(rng.integers(1, 5))
        lens = [int(rng.integers(1, 500)) for _ in range(n)]
        M = sum(lens)
        x = bf(torch.randn(M, d, generator=gen) * float(rng.uniform(0.2, 3.0)) + float(rng.uniform(-1, 1)))
        segs, ref, r0 = [], torch.empty(M, d, dtype=torch.bfloat16), 0
        for ln in lens:
            sh, sc = bf(torch.randn(1, d, generator=gen) * 0.5), bf(torch.randn(1, d, generator=gen) * 0.5)
            segs.append((r0 + ln, sh.cuda(), sc.cuda()))
            ref[r0:r0 + ln] = O.layer_norm(x[r0:r0 + ln]) * (1 + sc) + sh
            r0 += ln
        out = torch.empty(M, d, dtype=torch.bfloat16, device="cuda")
        ops.ln_modulate_segs(x.cuda(), out, segs)
        diff = (out.cpu().float() - ref.float()).abs()
        tag = f"seed {seed} case {case}: d={d} segments={lens}"
        assert float(diff.max()) <= 2 ** -6 * float(ref.float().abs().max()), tag
        assert float((diff > 0).float().mean()) < 0.03, tag
