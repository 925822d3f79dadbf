"""-m gpu: MMDiT block kernels (through the C ABI) vs torch-CPU fp32 references of the same op.
Floating point: tolerances are stated per test; bf16 outputs are compared after the same rounding
points as the eager bf16 sequence, so most results agree to <= 1 bf16 ulp."""
import math
import os

import pytest
import torch
import torch.nn.functional as F

from oracle import regione_oracle as O
import plan_helpers as P

pytestmark = pytest.mark.gpu


def bf(x):
    return x.to(torch.bfloat16)


def rel_err(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.mark.parametrize("variant", ["128", "256c", "256"])
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 384, 512), (300, 256, 192), (1, 128, 64), (1000, 64, 3072),
                                   (777, 200, 128), (8704, 3072, 3072)])
def test_gemm_bias(M, N, K, variant):
    from regione_amd import ops
    # 128 = 128x128 tiles, 256c = 256x256 tiles (8 waves, hipcc-scheduled loop), 256 = 256x256 tiles with the hand-scheduled
    # 4-wave K loop wherever it applies (K >= 128)
    P.geometry(variant)
    g = torch.Generator().manual_seed(M * 7 + N)
    # asymmetric operands so a transposed C-write cannot pass (guide rule 16)
    A = bf(torch.randn(M, K, generator=g))
    W = bf(torch.randn(N, K, generator=g) * 0.05)
    b = bf(torch.randn(N, generator=g))
    out = torch.zeros(M, N, dtype=torch.bfloat16).cuda()
    ops.gemm(A.cuda(), W.cuda(), b.cuda(), out)
    if M * N * K <= 2 ** 31:
        ref = (A.double() @ W.double().T + b.double())
    else:
        ref = (A.cuda().float() @ W.cuda().float().T + b.cuda().float()).cpu().double()
    err = (out.cpu().double() - ref).abs()
    tol = 2 ** -8 * ref.abs() + 1e-2           # one bf16 rounding of the result + fp32 accumulation noise
    assert bool((err <= tol).all()), float((err / tol).max())
    assert rel_err(out.cpu(), ref) < 3e-3


def test_gemm_strided_views_scatter_and_epilogues():
    from regione_amd import ops
    g = torch.Generator().manual_seed(5)
    M, N, K = 200, 256, 128
    big = bf(torch.randn(M, 512, generator=g)).cuda()
    A = big[:, 128:256]                                           # lda = 512
    W = bf(torch.randn(N, K, generator=g) * 0.1).cuda()
    b = bf(torch.randn(N, generator=g)).cuda()
    # GELU from a column
    outbuf = torch.zeros(M, 640, dtype=torch.bfloat16).cuda()
    out = outbuf[:, 64:64 + N]                                    # ldc = 640
    ops.gemm(A, W, b, out, epilogue=ops.EPI_GELU, gelu_from_col=128)
    lin = bf(F.linear(A.cpu().float(), W.cpu().float(), b.cpu().float()))
    ref = lin.clone()
    ref[:, 128:] = F.gelu(lin[:, 128:], approximate="tanh")
    assert rel_err(out.cpu(), ref) < 4e-3
    assert float((out.cpu().float() - ref.float()).abs().max()) <= 2 ** -6 * float(ref.float().abs().max())
    assert float(outbuf[:, :64].abs().max()) == 0 and float(outbuf[:, 64 + N:].abs().max()) == 0
    # gated residual in place
    resid = bf(torch.randn(M, N, generator=g)).cuda()
    gate = bf(torch.randn(N, generator=g)).cuda()
    r0 = resid.clone()
    ops.gemm(A, W, b, resid, epilogue=ops.EPI_GATE_RESID, gate=gate, resid=resid)
    ref = r0.cpu() + gate.cpu().unsqueeze(0) * lin
    assert rel_err(resid.cpu(), ref) < 4e-3
    # row scatter (the _partially_linear replacement, fused_kernels.py:77-80)
    cache = bf(torch.randn(1000, N, generator=g)).cuda()
    c0 = cache.clone().cpu()
    idx = torch.randperm(1000, generator=g)[:M].sort().values
    ops.gemm(A, W, b, cache, out_rows=idx.cuda())
    ref = O.partially_linear(A.cpu().unsqueeze(0), W.cpu(), b.cpu(), idx, c0.unsqueeze(0).clone(), fp16_roundtrip=False)[0]
    untouched = torch.ones(1000, dtype=torch.bool)
    untouched[idx] = False
    assert torch.equal(cache.cpu()[untouched], c0[untouched])
    assert rel_err(cache.cpu()[idx], ref[idx]) < 3e-3


@pytest.mark.parametrize("variant", ["128", "256c", "256"])
def test_gemm_pair_two_problems_one_launch(variant):
    from regione_amd import ops
    P.geometry(variant)
    g = torch.Generator().manual_seed(9)
    N, K, M0, M1 = 384, 256, 700, 90
    A0, A1 = bf(torch.randn(M0, K, generator=g)).cuda(), bf(torch.randn(M1, K, generator=g)).cuda()
    W0, W1 = bf(torch.randn(N, K, generator=g) * 0.1).cuda(), bf(torch.randn(N, K, generator=g) * 0.1).cuda()
    b0, b1 = bf(torch.randn(N, generator=g)).cuda(), bf(torch.randn(N, generator=g)).cuda()
    g0, g1 = bf(torch.randn(N, generator=g)).cuda(), bf(torch.randn(N, generator=g)).cuda()
    r0, r1 = bf(torch.randn(M0, N, generator=g)).cuda(), bf(torch.randn(M1, N, generator=g)).cuda()
    o0, o1 = torch.empty_like(r0), torch.empty_like(r1)
    ops.gemm_pair(A0, W0, b0, o0, A1, W1, b1, o1)
    s0, s1 = torch.empty_like(r0), torch.empty_like(r1)
    ops.gemm(A0, W0, b0, s0)
    ops.gemm(A1, W1, b1, s1)
    assert torch.equal(o0, s0) and torch.equal(o1, s1)          # identical to two separate launches
    ref0 = F.linear(A0.cpu().double(), W0.cpu().double(), b0.cpu().double())
    assert rel_err(o0.cpu(), ref0) < 3e-3
    # gated residual, in place, both problems
    x0, x1 = r0.clone(), r1.clone()
    ops.gemm_pair(A0, W0, b0, x0, A1, W1, b1, x1, epilogue=ops.EPI_GATE_RESID, gate0=g0, resid0=x0, gate1=g1, resid1=x1)
    y0, y1 = r0.clone(), r1.clone()
    ops.gemm(A0, W0, b0, y0, epilogue=ops.EPI_GATE_RESID, gate=g0, resid=y0)
    ops.gemm(A1, W1, b1, y1, epilogue=ops.EPI_GATE_RESID, gate=g1, resid=y1)
    assert torch.equal(x0, y0) and torch.equal(x1, y1)


@pytest.mark.parametrize("B,N,K,silu", [(1, 18432, 3072, True), (2, 1000, 256, False), (1, 3072, 768, False)])
def test_gemv(B, N, K, silu):
    from regione_amd import ops
    g = torch.Generator().manual_seed(B + N)
    x = bf(torch.randn(B, K, generator=g))
    W = bf(torch.randn(N, K, generator=g) * 0.05)
    b = bf(torch.randn(N, generator=g))
    y = ops.gemv(x.cuda(), W.cuda(), b.cuda(), silu_input=silu)
    xin = F.silu(x) if silu else x
    ref = F.linear(xin.double(), W.double(), b.double())
    assert rel_err(y.cpu(), ref) < 3e-3


@pytest.mark.parametrize("d", [256, 3072])
def test_ln_modulate(d):
    from regione_amd import ops
    g = torch.Generator().manual_seed(d)
    M, T = 300, 40
    x = bf(torch.randn(M, d, generator=g) * 2 + 0.3)
    mods = [bf(torch.randn(1, d, generator=g) * 0.5) for _ in range(4)]
    out = torch.empty(M, d, dtype=torch.bfloat16).cuda()
    ops.ln_modulate(x.cuda(), out, mods[2].cuda(), mods[3].cuda(), split_row=T, shift0=mods[0].cuda(), scale0=mods[1].cuda())
    ref = torch.empty_like(x)
    ref[:T] = O.layer_norm(x[:T]) * (1 + mods[1]) + mods[0]
    ref[T:] = O.layer_norm(x[T:]) * (1 + mods[3]) + mods[2]
    diff = (out.cpu().float() - ref.float()).abs()
    assert float(diff.max()) <= 2 ** -6 * float(ref.float().abs().max())        # <= ~2 bf16 ulp at the largest magnitude
    assert float((diff > 0).float().mean()) < 0.02                               # and almost always bit-identical


def test_qk_norm_rope_store_and_attention_vs_reference_processor_math():
    """Whole attention path of a single-stream block at toy size: raw [k|v|q] projections ->
    RMSNorm/RoPE/store -> region attention, against the oracle's processor math
    (RegionE/FluxKontext/inplace.py:754-806)."""
    from regione_amd import ops
    from regione_amd import synth
    g = torch.Generator().manual_seed(11)
    H, T, h_tok, w_tok = 2, 32, 16, 16
    N = 2 * h_tok * w_tok
    S = T + N                                                       # 544: not a multiple of 64 -> tail mask
    D = H * 128
    raw = bf(torch.randn(S, 3 * D, generator=g))
    wq, wk = bf(1 + 0.1 * torch.randn(128, generator=g)), bf(1 + 0.1 * torch.randn(128, generator=g))
    ids = torch.cat([torch.zeros(T, 3), synth.flux_latent_ids(h_tok, w_tok)], 0)
    cos, sin = O.flux_pos_embed(ids)
    skv_pad = ops.padded(S)
    k_slab = torch.zeros(skv_pad, D, dtype=torch.bfloat16).cuda()
    vt_slab = torch.zeros(D, skv_pad, dtype=torch.bfloat16).cuda()
    buf = raw.cuda().clone()
    rope = (cos.cuda(), sin.cuda())
    ops.qk_norm_rope_store(buf, 0, D, 2 * D, H, wq.cuda(), wk.cuda(), rope, rope, k_slab, vt_slab)
    # reference
    k = raw[:, :D].view(1, S, H, 128).transpose(1, 2)
    v = raw[:, D:2 * D].view(1, S, H, 128).transpose(1, 2)
    q = raw[:, 2 * D:].view(1, S, H, 128).transpose(1, 2)
    qn = O.apply_rope(O.rms_norm(q, wq), cos, sin)
    kn = O.apply_rope(O.rms_norm(k, wk), cos, sin)
    q_out = buf[:, 2 * D:].cpu().view(S, H, 128).transpose(0, 1)
    assert float((q_out.float() - qn[0].float()).abs().max()) <= 2 ** -7 * float(qn.float().abs().max())
    assert float((q_out != qn[0]).float().mean()) < 0.01
    k_out = k_slab[:S].cpu().view(S, H, 128).transpose(0, 1)
    assert float((k_out != kn[0]).float().mean()) < 0.01
    # V^T slab: column kvpos(r) holds row r
    r = torch.arange(S)
    pos = (r & ~12) | ((r & 4) << 1) | ((r & 8) >> 1)
    vt = vt_slab.cpu()[:, pos]                                      # [D, S]
    assert torch.equal(vt.view(H, 128, S).permute(0, 2, 1), v[0])
    # attention over a compacted query subset (Sq != Skv)
    sel = torch.cat([torch.arange(T), T + torch.randperm(N, generator=g)[:100].sort().values])
    qbuf = buf[:, 2 * D:][sel.cuda()].contiguous()
    out = torch.empty_like(qbuf)
    ops.attention(qbuf, k_slab, vt_slab, out, S, H)
    ref = F.scaled_dot_product_attention(qn[:, :, sel].float(), kn.float(), v.float())
    ref = ref.transpose(1, 2).reshape(len(sel), D)
    assert rel_err(out.cpu(), ref) < 1e-2
    assert float((out.cpu().float() - ref).abs().max()) < 2e-2 * float(ref.abs().max()) + 1e-2


@pytest.mark.parametrize("Sq,Skv,H", [(128, 64, 1), (200, 1000, 3), (1536, 8704, 24)])
def test_attention_shapes(Sq, Skv, H):
    from regione_amd import ops
    g = torch.Generator().manual_seed(Sq + Skv)
    D = H * 128
    q = bf(torch.randn(Sq, D, generator=g))
    k = bf(torch.randn(Skv, D, generator=g))
    v = bf(torch.randn(Skv, D, generator=g))
    k[5] *= 6.0                                                       # a spiky key forces a late running-max jump
    pad = ops.padded(Skv)
    ks = torch.zeros(pad, D, dtype=torch.bfloat16)
    ks[:Skv] = k
    r = torch.arange(Skv)
    pos = (r & ~12) | ((r & 4) << 1) | ((r & 8) >> 1)
    vt = torch.zeros(D, pad, dtype=torch.bfloat16)
    vt[:, pos] = v.T
    out = torch.empty(Sq, D, dtype=torch.bfloat16).cuda()
    ops.attention(q.cuda(), ks.cuda(), vt.cuda(), out, Skv, H)
    dev = "cuda" if Sq * Skv * H > 5e6 else "cpu"
    qq, kk, vv = (t.to(dev).float().view(-1, H, 128).transpose(0, 1) for t in (q, k, v))
    ref = F.scaled_dot_product_attention(qq[None], kk[None], vv[None])[0].transpose(0, 1).reshape(Sq, D).cpu()
    assert rel_err(out.cpu(), ref) < 1e-2


@pytest.mark.parametrize("Sq,Skv,H", [(717, 8704, 24), (1024, 8704, 24), (1137, 8576, 24), (2537, 8704, 24), (1408, 1600, 24),
                                      (4608, 33280, 24)])
def test_attention_stream_k_remainder(Sq, Skv, H):
    """Region-step query sets (Sq = T + K_e) leave fewer items than CUs: the (item, KV tile) steps are dealt out in equal
    contiguous runs (stream-K), a run may cross one item boundary, attention_combine_sk_kernel merges the partials.  Must agree
    with the unsplit schedule, with the equal-split schedule and with an fp32 reference (ragged last query block, a spiky key
    late in the sequence so that partial maxima differ between runs)."""
    from regione_amd import ops
    g = torch.Generator().manual_seed(Sq * 7 + Skv)
    D = H * 128
    q = bf(torch.randn(Sq, D, generator=g)).cuda()
    k = bf(torch.randn(Skv, D, generator=g)).cuda()
    v = bf(torch.randn(Skv, D, generator=g)).cuda()
    k[Skv - 77] *= 5.0
    k[3] *= 4.0
    pad = ops.padded(Skv)
    ks = torch.zeros(pad, D, dtype=torch.bfloat16, device="cuda")
    ks[:Skv] = k
    r = torch.arange(Skv)
    pos = ((r & ~12) | ((r & 4) << 1) | ((r & 8) >> 1)).cuda()
    vt = torch.zeros(D, pad, dtype=torch.bfloat16, device="cuda")
    vt[:, pos] = v.T
    outs = {}
    for name, knobs in (("stream_k", dict(attn_streamk=1)), ("auto", {}), ("equal_split", dict(attn_streamk=0)),
                        ("unsplit", dict(attn_waves=8, attn_split=0))):
        with ops._lib.plan_override(**knobs):
            o = torch.empty_like(q)
            ops.attention(q, ks, vt, o, Skv, H)
            torch.cuda.synchronize()
            outs[name] = o.cpu()
    assert rel_err(outs["stream_k"], outs["unsplit"]) < 2e-3 and rel_err(outs["equal_split"], outs["unsplit"]) < 2e-3
    assert torch.equal(outs["auto"], outs["stream_k"]) or torch.equal(outs["auto"], outs["equal_split"]) or \
        torch.equal(outs["auto"], outs["unsplit"])
    rows = torch.randperm(Sq, generator=g)[:192].sort().values.cuda()
    qq = q[rows].float().view(-1, H, 128).transpose(0, 1)
    kk, vv = k.float().view(Skv, H, 128).transpose(0, 1), v.float().view(Skv, H, 128).transpose(0, 1)
    ref = F.scaled_dot_product_attention(qq[None], kk[None], vv[None])[0].transpose(0, 1).reshape(len(rows), D).cpu()
    assert rel_err(outs["stream_k"][rows.cpu()], ref) < 1e-2
    assert torch.isfinite(outs["stream_k"].float()).all()


def test_gemm_round_aware_split_k_path():
    """Shapes whose tile count leaves a partial last round take the split-K remainder + reduce pass
    (proj_out of a FLUX single block: 408 tiles of 256x256 on 256 CUs).  Reference: fp32 matmul on the GPU."""
    from regione_amd import ops
    g = torch.Generator().manual_seed(21)
    M, N, K = 8704, 3072, 15360
    A = bf(torch.randn(M, K, generator=g) * 0.5).cuda()
    W = bf(torch.randn(N, K, generator=g) * 0.02).cuda()
    b = bf(torch.randn(N, generator=g)).cuda()
    gate = bf(torch.randn(N, generator=g)).cuda()
    resid = bf(torch.randn(M, N, generator=g)).cuda()
    lin = (A.float() @ W.float().T + b.float())
    out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    ops.gemm(A, W, b, out)
    from regione_amd import _lib
    plan = _lib.lib().rgn_gemm_last_plan()
    assert (plan & 0x400) and (plan & 0xff) > 1, f"the remainder was not cut along K (plan {plan:#x})"
    assert rel_err(out.cpu(), lin.cpu()) < 3e-3
    with _lib.plan_override(gemm_pieces=1):
        out2 = torch.empty_like(out)
        ops.gemm(A, W, b, out2)
    # split and unsplit schedules agree to fp32 summation order (<= 1 bf16 ulp on a few elements)
    assert float((out != out2).float().mean()) < 0.02
    assert float((out.float() - out2.float()).abs().max()) <= 2 ** -7 * float(out2.float().abs().max())
    x = resid.clone()
    ops.gemm(A, W, b, x, epilogue=ops.EPI_GATE_RESID, gate=gate, resid=x)
    ref = resid.float() + (gate.float() * bf(lin).float())
    assert rel_err(x.cpu(), ref.cpu()) < 4e-3


def test_attention_full_size_round_aware_vs_unsplit():
    """Sq = Skv = 8704, H = 24 (FLUX full step): 816 items = 3 full rounds + 48 remainder items that are cut
    along KV and merged; must agree with the unsplit schedule and with an fp32 reference on sampled rows."""
    from regione_amd import ops
    g = torch.Generator().manual_seed(33)
    S, H = 8704, 24
    D = H * 128
    q = bf(torch.randn(S, D, generator=g)).cuda()
    k = bf(torch.randn(S, D, generator=g)).cuda()
    v = bf(torch.randn(S, D, generator=g)).cuda()
    r = torch.arange(S)
    pos = ((r & ~12) | ((r & 4) << 1) | ((r & 8) >> 1)).cuda()
    vt = torch.zeros(D, S, dtype=torch.bfloat16, device="cuda")
    vt[:, pos] = v.T
    out = torch.empty_like(q)
    ops.attention(q, k, vt, out, S, H)
    with ops._lib.plan_override(attn_waves=8, attn_split=0):
        out2 = torch.empty_like(q)
        ops.attention(q, k, vt, out2, S, H)
    assert rel_err(out.cpu(), out2.cpu()) < 2e-3
    rows = torch.randperm(S, generator=g)[:256].sort().values.cuda()
    qq = q[rows].float().view(-1, H, 128).transpose(0, 1)
    kk, vv = k.float().view(S, H, 128).transpose(0, 1), v.float().view(S, H, 128).transpose(0, 1)
    ref = F.scaled_dot_product_attention(qq[None], kk[None], vv[None])[0].transpose(0, 1).reshape(len(rows), D)
    assert rel_err(out[rows].cpu(), ref.cpu()) < 1e-2
    # in-place form used by the engine (O aliases Q)
    q2 = q.clone()
    ops.attention(q2, k, vt, q2, S, H)
    assert torch.equal(q2, out)


def _qkv_case(M, H, K, mlp, gen, kv_rows=None, skv=None, row_base=0):
    """Random single-problem QKV(+MLP) projection inputs; returns everything both paths need."""
    D = H * 128
    N = 3 * D + mlp
    A = bf(torch.randn(M, K, generator=gen)).cuda()
    W = bf(0.05 * torch.randn(N, K, generator=gen)).cuda()
    b = bf(0.1 * torch.randn(N, generator=gen)).cuda()
    wq = bf(1 + 0.1 * torch.randn(128, generator=gen)).cuda()
    wk = bf(1 + 0.1 * torch.randn(128, generator=gen)).cuda()
    skv = skv or (row_base + M)
    ang = torch.rand(max(skv, row_base + M), 64, generator=gen) * 6.28
    cos = torch.repeat_interleave(torch.cos(ang), 2, dim=1).contiguous().cuda()
    sin = torch.repeat_interleave(torch.sin(ang), 2, dim=1).contiguous().cuda()
    return A, W, b, wq, wk, (cos, sin), D, N, skv


@pytest.mark.parametrize("variant", ["128", "256c", "256"])
@pytest.mark.parametrize("M,H,K,mlp,gather,row_base", [(600, 2, 256, 1024, False, 0), (333, 2, 256, 0, True, 0),
                                                       (512, 4, 512, 2048, False, 16), (200, 2, 256, 0, True, 24)])
def test_gemm_qkv_fused_epilogue_bit_identical_to_separate_kernels(M, H, K, mlp, gather, row_base, variant):
    """rgn_gemm_bf16_qkv == rgn_gemm_bf16 followed by rgn_qk_norm_rope_store, bit for bit: Q (in place), the
    GELU(mlp) columns, the K slab and the V^T slab - identity rows, gathered cache rows (region step) and a
    problem that starts at a joint-sequence offset that is / is not a multiple of 16."""
    from regione_amd import ops
    P.geometry(variant)
    gen = torch.Generator().manual_seed(M + H)
    skv = row_base + (M if not gather else 3 * M)
    A, W, b, wq, wk, rope, D, N, skv = _qkv_case(M, H, K, mlp, gen, skv=skv, row_base=row_base)
    kv_rows = None
    if gather:                                     # joint rows -> scattered cache rows (ascending, like edited ids)
        kv_rows = torch.cat([torch.arange(row_base), row_base + torch.randperm(skv - row_base, generator=gen)[:M].sort().values]).cuda()
    skv_pad = ops.padded(skv)

    def slabs():
        return (torch.zeros(skv_pad, D, dtype=torch.bfloat16, device="cuda"), torch.zeros(D, skv_pad, dtype=torch.bfloat16, device="cuda"))
    # separate kernels: `wide` holds the joint sequence, this problem's rows start at row_base
    wide = torch.zeros(row_base + M, N, dtype=torch.bfloat16, device="cuda")
    ops.gemm(A, W, b, wide[row_base:], epilogue=ops.EPI_GELU, gelu_from_col=3 * D)
    k0, v0 = slabs()
    ops.qk_norm_rope_store(wide, 0, D, 2 * D, H, wq, wk, rope, rope, k0, v0, kv_rows)
    # fused
    out = torch.zeros(row_base + M, N, dtype=torch.bfloat16, device="cuda")
    k1, v1 = slabs()
    epi = ops.qkv_epilogue(wq=wq, wk=wk, rope_q=rope, rope_k=rope, k_slab=k1, vt_slab=v1, H=H, k_col=0, v_col=D, q_col=2 * D,
                           kv_rows=kv_rows, row_base=row_base)
    ops.gemm_qkv(A, W, b, out[row_base:], epi, gelu_from_col=3 * D)
    torch.cuda.synchronize()
    assert torch.equal(out[row_base:, 2 * D:], wide[row_base:, 2 * D:])         # Q and GELU(mlp)
    rows = (kv_rows[row_base:] if kv_rows is not None else torch.arange(row_base, row_base + M, device="cuda"))
    assert torch.equal(k1[rows], k0[rows])
    pos = ((rows & ~12) | ((rows & 4) << 1) | ((rows & 8) >> 1))
    assert torch.equal(v1[:, pos], v0[:, pos])
    k1[rows] = 0
    v1[:, pos] = 0
    assert not k1.any() and not v1.any()                                        # nothing outside this problem's cache rows
    assert not out[:, :2 * D].any() and not out[:row_base].any()                # K / V columns are not written to C


def test_gemm_qkv_pair_and_full_size_split_path():
    """(a) text + image problems of a double block in one launch (row_base = T for the image stream, separate norm
    weights); (b) the FLUX single-block shape (M = 8704, N = 21504): remainder tiles take the split-K + reduce path."""
    from regione_amd import ops
    gen = torch.Generator().manual_seed(3)
    H, K, T, Mi = 2, 256, 48, 400
    D, N = H * 128, 3 * H * 128
    Ai, W0, b0, wq0, wk0, rope, _, _, _ = _qkv_case(Mi, H, K, 0, gen, skv=T + Mi)
    At, W1, b1, wq1, wk1, _, _, _, _ = _qkv_case(T, H, K, 0, gen)
    skv_pad = ops.padded(T + Mi)
    wide = torch.zeros(T + Mi, N, dtype=torch.bfloat16, device="cuda")
    ops.gemm_pair(Ai, W0, b0, wide[T:], At, W1, b1, wide[:T])
    k0 = torch.zeros(skv_pad, D, dtype=torch.bfloat16, device="cuda"); v0 = torch.zeros(D, skv_pad, dtype=torch.bfloat16, device="cuda")
    ops.qk_norm_rope_store(wide, 0, D, 2 * D, H, wq0, wk0, rope, rope, k0, v0, None, split_row=T, wq0=wq1, wk0=wk1)
    out = torch.zeros_like(wide)
    k1, v1 = torch.zeros_like(k0), torch.zeros_like(v0)
    common = dict(rope_q=rope, rope_k=rope, k_slab=k1, vt_slab=v1, H=H, k_col=0, v_col=D, q_col=2 * D)
    ops.gemm_qkv_pair(Ai, W0, b0, out[T:], ops.qkv_epilogue(wq=wq0, wk=wk0, row_base=T, **common),
                      At, W1, b1, out[:T], ops.qkv_epilogue(wq=wq1, wk=wk1, row_base=0, **common))
    assert torch.equal(out[:, 2 * D:], wide[:, 2 * D:]) and torch.equal(k1, k0) and torch.equal(v1, v0)
    # (b) full size
    H, K, M = 24, 3072, 8704
    D = H * 128
    A, W, b, wq, wk, rope, D, N, skv = _qkv_case(M, H, K, 4 * D, gen)
    skv_pad = ops.padded(M)
    wide = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    ops.gemm(A, W, b, wide, epilogue=ops.EPI_GELU, gelu_from_col=3 * D)
    k0 = torch.zeros(skv_pad, D, dtype=torch.bfloat16, device="cuda"); v0 = torch.zeros(D, skv_pad, dtype=torch.bfloat16, device="cuda")
    ops.qk_norm_rope_store(wide, 0, D, 2 * D, H, wq, wk, rope, rope, k0, v0)
    out = torch.empty_like(wide)
    k1, v1 = torch.zeros_like(k0), torch.zeros_like(v0)
    ops.gemm_qkv(A, W, b, out, ops.qkv_epilogue(wq=wq, wk=wk, rope_q=rope, rope_k=rope, k_slab=k1, vt_slab=v1, H=H, k_col=0,
                                                v_col=D, q_col=2 * D), gelu_from_col=3 * D)
    assert torch.equal(out[:, 2 * D:], wide[:, 2 * D:]) and torch.equal(k1, k0) and torch.equal(v1, v0)


def test_gemm_qkv_fp16_roundtrip_option():
    """rgn_qkv_epilogue.fp16_roundtrip = 1 (the reference's partial-update kernel stores acc.to(fp16) into the bf16 cache,
    fused_kernels.py:80): only the K / V columns change, by at most one bf16 ulp; Q and the MLP half are untouched."""
    from regione_amd import ops
    gen = torch.Generator().manual_seed(5)
    M, H, K, mlp = 300, 2, 256, 512
    A, W, b, wq, wk, rope, D, N, skv = _qkv_case(M, H, K, mlp, gen)
    skv_pad = ops.padded(skv)
    res = []
    for rt in (False, True):
        out = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
        ks = torch.zeros(skv_pad, D, dtype=torch.bfloat16, device="cuda")
        vs = torch.zeros(D, skv_pad, dtype=torch.bfloat16, device="cuda")
        epi = ops.qkv_epilogue(wq=wq, wk=wk, rope_q=rope, rope_k=rope, k_slab=ks, vt_slab=vs, H=H, k_col=0, v_col=D, q_col=2 * D,
                               fp16_roundtrip=rt)
        ops.gemm_qkv(A, W, b, out, epi, gelu_from_col=3 * D)
        res.append((out, ks, vs))
    (o0, k0, v0), (o1, k1, v1) = res
    assert torch.equal(o0, o1)                                               # Q, GELU(mlp): single rounding either way
    dv = (v0.float() - v1.float()).abs()
    assert float((v0 != v1).float().mean()) > 0.0 and float((dv / v0.float().abs().clamp_min(1e-3)).max()) <= 2 ** -7
    assert float((k0.float() - k1.float()).abs().max()) <= 2 ** -5 * float(k0.float().abs().max())


@pytest.mark.parametrize("nsplit", [2, 3, 5])
@pytest.mark.parametrize("shape,epi", [((1536, 3072, 15360), "gate"), ((8704, 3072, 15360), "gate"), ((708, 3072, 12288), "bias"),
                                       (((1024, 512), 3072, 12288), "gate"), (((196, 512), 12288, 3072), "gelu")])
def test_gemm_split_k_hand_scheduled_pieces_bit_identical_to_compiler_scheduled(shape, epi, nsplit):
    """Remainder tiles are cut along K: every piece runs the hand-scheduled K loop over its K range and dumps fp32 fragments,
    a reduce launch of the same geometry sums them in index order (batched loads into the AGPR accumulators) and runs the
    epilogue.  Same piece count -> bit-identical to the 8-wave partial + reduce launches, launch after launch, and within
    rounding of the unsplit result."""
    from regione_amd import ops
    Ms, N, K = shape
    pair = isinstance(Ms, tuple)
    g = torch.Generator().manual_seed(N + K + nsplit)
    mk = lambda M: bf(torch.randn(M, K, generator=g)).cuda()
    As = [mk(M) for M in (Ms if pair else (Ms,))]
    Ws = [bf(torch.randn(N, K, generator=g) * 0.05).cuda() for _ in As]
    b = bf(torch.randn(N, generator=g)).cuda()
    gate = bf(torch.randn(N, generator=g)).cuda()
    xs = [bf(torch.randn(A.shape[0], N, generator=g)).cuda() for A in As]
    E = {"gate": ops.EPI_GATE_RESID, "gelu": ops.EPI_GELU, "bias": ops.EPI_BIAS}[epi]

    def run():
        outs = [x.clone() for x in xs]
        if pair:
            kw = dict(gate0=gate, resid0=outs[0], gate1=gate, resid1=outs[1]) if epi == "gate" else {}
            ops.gemm_pair(As[0], Ws[0], b, outs[0], As[1], Ws[1], b, outs[1], epilogue=E, **kw)
        else:
            kw = dict(gate=gate, resid=outs[0]) if epi == "gate" else {}
            ops.gemm(As[0], Ws[0], b, outs[0], epilogue=E, **kw)
        torch.cuda.synchronize()
        return torch.cat(outs)
    P.geometry("256")
    P.force(gemm_pieces=nsplit)
    fix = [run() for _ in range(3)]
    assert torch.equal(fix[0], fix[1]) and torch.equal(fix[0], fix[2])
    P.geometry("256c")                                            # 8-wave geometry for the pieces and the reduce pass
    red = run()
    assert torch.equal(fix[0], red), float((fix[0].float() - red.float()).abs().max())
    P.force(gemm_pieces=1)
    whole = run()
    assert rel_err(fix[0].cpu(), whole.cpu()) < 2e-3


@pytest.mark.parametrize("M,N,K,epi", [(8704, 3072, 3072, "bias"), (1536, 21504, 3072, "gelu"), (700, 3072, 15360, "gate"),
                                       (513, 520, 128, "bias"), (8192, 512, 192, "gelu"), (300, 704, 256, "gate"),
                                       (2000, 1000, 320, "bias")])
def test_gemm_hand_scheduled_loop_bit_identical_to_compiler_scheduled(M, N, K, epi):
    """The 4-wave asm K loop ("256": A ring of two / W ring of three 32 KiB slots for K >= 256, the two-stage loop for K = 128 /
    192) accumulates every output element over k in the same MFMA order as the 8-wave compiler-scheduled kernel ("256c", the
    fallback geometry): results must be bit-identical, for every epilogue, ragged edges included."""
    from regione_amd import ops
    g = torch.Generator().manual_seed(M + N + K)
    A = bf(torch.randn(M, K, generator=g)).cuda()
    W = bf(torch.randn(N, K, generator=g) * 0.05).cuda()
    b = bf(torch.randn(N, generator=g)).cuda()
    gate, x = bf(torch.randn(N, generator=g)).cuda(), bf(torch.randn(M, N, generator=g)).cuda()
    outs = []
    # whole-K tiles only: the two geometries' launch planners may cut a remainder into different numbers of K pieces
    # (split remainders at EQUAL piece counts: test_gemm_split_k_hand_scheduled_pieces_bit_identical_to_compiler_scheduled)
    P.force(gemm_pieces=1)
    for variant in ("256c", "256"):
        P.geometry(variant)
        if epi == "gate":
            o = x.clone()
            ops.gemm(A, W, b, o, epilogue=ops.EPI_GATE_RESID, gate=gate, resid=o)
        else:
            o = torch.full((M, N), 7.0, dtype=torch.bfloat16, device="cuda")
            ops.gemm(A, W, b, o, epilogue=ops.EPI_GELU if epi == "gelu" else ops.EPI_BIAS, gelu_from_col=N // 2 // 8 * 8)
        outs.append(o)
    torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[1]), float((outs[0].float() - outs[1].float()).abs().max())
    ref = F.linear(A.float(), W.float(), b.float())
    if epi == "bias":
        assert rel_err(outs[1].cpu(), ref.cpu()) < 3e-3


@pytest.mark.parametrize("M,N,K,epi,variant", [(8704, 3072, 3072, "bias", "256c"), (1536, 21504, 3072, "gelu", "256c"), (700, 3072, 15360, "gate", "256c"),
                                               (513, 520, 128, "bias", "128"), (300, 704, 256, "gate", "128"), (2000, 1000, 320, "bias", "256c")])
def test_gemm_fp8_weights_per_channel_scale(M, N, K, epi, variant):
    """rgn_gemm_w8: W stored as OCP e4m3fn + one fp32 scale per output channel.  Reference = the same GEMM on the
    DEQUANTISED weights in fp32 (the conversion fp8 -> bf16 in the kernel is exact; the scale multiplies the fp32
    accumulator): the result must agree to bf16 output rounding - tolerance 2^-8 relative + accumulation noise, like the
    bf16 kernel against its fp32 reference."""
    from regione_amd import ops
    P.geometry(variant)
    g = torch.Generator().manual_seed(M + N + K + 1)
    A = bf(torch.randn(M, K, generator=g)).cuda()
    W = bf(torch.randn(N, K, generator=g) * 0.05 * (1 + torch.arange(N)[:, None] % 7)).cuda()     # rows of different magnitude
    b = bf(torch.randn(N, generator=g)).cuda()
    Wq = ops.quantize_w8(W)
    assert Wq.dtype == torch.float8_e4m3fn and Wq._rgn_scale.shape == (N,)
    Wd = Wq.float() * Wq._rgn_scale[:, None]                                    # what the kernel multiplies with
    assert float((Wd - W.float()).abs().max() / W.float().abs().max()) < 0.07  # e4m3: 3 mantissa bits
    gate, x = bf(torch.randn(N, generator=g)).cuda(), bf(torch.randn(M, N, generator=g)).cuda()
    lin = A.float() @ Wd.T + b.float()
    if epi == "gate":
        o = x.clone()
        ops.gemm(A, Wq, b, o, epilogue=ops.EPI_GATE_RESID, gate=gate, resid=o)
        gated = (gate.float()[None] * bf(lin).float()).to(torch.bfloat16).float()
        ref = x.float() + gated
        # the terms, not their (cancelling) sum: a one-ulp flip of bf16(lin) (different fp32 accumulation order over K) moves
        # the result by |gate| * ulp(lin)
        mag = x.float().abs() + gated.abs() + gate.float().abs()[None] * lin.abs()
    elif epi == "gelu":
        o = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
        ops.gemm(A, Wq, b, o, epilogue=ops.EPI_GELU, gelu_from_col=N // 2 // 8 * 8)
        ref = bf(lin).float()
        c = N // 2 // 8 * 8
        ref[:, c:] = F.gelu(ref[:, c:], approximate="tanh")
    else:
        o = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
        ops.gemm(A, Wq, b, o)
        ref = lin
    torch.cuda.synchronize()
    err = (o.float() - ref).abs()
    tol = 2 ** -7 * (mag if epi == "gate" else ref.abs()) + 2e-2
    assert bool((err <= tol).all()), float((err / tol).max())
    assert rel_err(o.cpu(), ref.cpu()) < 4e-3
    # and close to the UNQUANTISED GEMM within the format's error (per-channel scaling keeps every row's relative error)
    full = A.float() @ W.float().T + b.float()
    if epi == "bias":
        assert rel_err(o.cpu(), full.cpu()) < 4e-2


def test_gemm_fp8_weights_fused_qkv_epilogue_and_pair():
    """The fused Q/K/V epilogue and the two-problem launch on fp8 weights == the same calls on the dequantised bf16
    weights up to the bf16 rounding of the dequantised matrix (K slab / V^T slab / Q compared)."""
    from regione_amd import ops
    M, H, K = 600, 2, 256
    gen = torch.Generator().manual_seed(77)
    A, W, b, wq, wk, rope, D, N, skv = _qkv_case(M, H, K, 1024, gen)
    Wq = ops.quantize_w8(W)
    Wd = bf(Wq.float() * Wq._rgn_scale[:, None])
    outs = []
    pad = ops.padded(skv)
    for w in (Wq, Wd):
        ks = torch.zeros(pad, D, dtype=torch.bfloat16, device="cuda")
        vs = torch.zeros(D, pad, dtype=torch.bfloat16, device="cuda")
        out = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
        epi = ops.qkv_epilogue(wq=wq, wk=wk, rope_q=rope, rope_k=rope, k_slab=ks, vt_slab=vs, H=H, k_col=0, v_col=D, q_col=2 * D)
        ops.gemm_qkv(A, w, b, out, epi, gelu_from_col=3 * D)
        outs.append((out[:, 2 * D:].clone(), ks, vs))
    for a, r in zip(outs[0], outs[1]):
        assert rel_err(a.cpu(), r.cpu()) < 6e-3
    # pair: text + image problems in one launch == two single launches (bit-identical)
    A1 = bf(torch.randn(90, K, generator=gen)).cuda()
    W1q = ops.quantize_w8(bf(torch.randn(512, K, generator=gen) * 0.1).cuda())
    W0q = ops.quantize_w8(bf(torch.randn(512, K, generator=gen) * 0.1).cuda())
    b0 = bf(torch.randn(512, generator=gen)).cuda()
    o0, o1 = torch.empty(M, 512, dtype=torch.bfloat16, device="cuda"), torch.empty(90, 512, dtype=torch.bfloat16, device="cuda")
    ops.gemm_pair(A, W0q, b0, o0, A1, W1q, b0, o1)
    s0, s1 = torch.empty_like(o0), torch.empty_like(o1)
    ops.gemm(A, W0q, b0, s0)
    ops.gemm(A1, W1q, b0, s1)
    assert torch.equal(o0, s0) and torch.equal(o1, s1)


# ---------------------------------------------------------------------------------------------------------------------
# round 3: up to four problems per launch (text / image stream x cond / uncond CFG branch), segmented LN-modulate
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("variant", ["128", "256c", "256"])
def test_gemm_group_four_problems_shared_weights_identical_to_separate_launches(variant):
    """[image_cond, image_uncond] share W0, [text_cond, text_uncond] share W1; per-problem gate / residual.  Small shapes
    take no split-K path, so every output element sees the same accumulation order as in a launch of its own."""
    from regione_amd import ops
    P.geometry(variant)
    g = torch.Generator().manual_seed(19)
    N, K = 384, 256
    Ms = [700, 650, 90, 77]
    As = [bf(torch.randn(m, K, generator=g)).cuda() for m in Ms]
    W0, W1 = bf(torch.randn(N, K, generator=g) * 0.1).cuda(), bf(torch.randn(N, K, generator=g) * 0.1).cuda()
    b0, b1 = bf(torch.randn(N, generator=g)).cuda(), bf(torch.randn(N, generator=g)).cuda()
    Ws, bs = [W0, W0, W1, W1], [b0, b0, b1, b1]
    gates = [bf(torch.randn(N, generator=g)).cuda() for _ in Ms]
    res = [bf(torch.randn(m, N, generator=g)).cuda() for m in Ms]
    outs = [torch.empty_like(r) for r in res]
    ops.gemm_group([ops.Problem(a, w, b, o) for a, w, b, o in zip(As, Ws, bs, outs)])
    for a, w, b, o in zip(As, Ws, bs, outs):
        s = torch.empty_like(o)
        ops.gemm(a, w, b, s)
        assert torch.equal(o, s)
    assert rel_err(outs[1].cpu(), F.linear(As[1].cpu().double(), W0.cpu().double(), b0.cpu().double())) < 3e-3
    xs = [r.clone() for r in res]
    ops.gemm_group([ops.Problem(a, w, b, x, gate=gt, resid=x) for a, w, b, x, gt in zip(As, Ws, bs, xs, gates)],
                   epilogue=ops.EPI_GATE_RESID)
    for a, w, b, r, gt, x in zip(As, Ws, bs, res, gates, xs):
        y = r.clone()
        ops.gemm(a, w, b, y, epilogue=ops.EPI_GATE_RESID, gate=gt, resid=y)
        assert torch.equal(x, y)
    # three problems, one of them empty; GELU epilogue
    o3 = [torch.empty_like(r) for r in res[:3]]
    ops.gemm_group([ops.Problem(As[0], W0, b0, o3[0]), ops.Problem(As[1][:0], W0, b0, o3[1][:0]), ops.Problem(As[2], W1, b1, o3[2])],
                   epilogue=ops.EPI_GELU)
    for i in (0, 2):
        s = torch.empty_like(o3[i])
        ops.gemm(As[i], Ws[i], bs[i], s, epilogue=ops.EPI_GELU)
        assert torch.equal(o3[i], s)


@pytest.mark.parametrize("split", ["0", None])
def test_gemm_group_region_step_shapes_of_two_cfg_branches(split):
    """The region-step shapes of a batched CFG forward at FLUX / Qwen dimensions: image rows 1024 + 1024, text rows
    512 + 384, K = 3072 -> N = 12288 (ff1) and K = 12288 -> N = 3072 (ff2, gated residual).  Without the split-K schedule
    (gemm_pieces = 1) the group is bit-identical to per-branch pair launches; with it the piece count may differ between
    the two launch shapes (fp32 summation order), so the comparison is against a float64 reference."""
    from regione_amd import ops
    if split is not None:
        P.force(gemm_pieces=1)
    g = torch.Generator().manual_seed(23)
    d, ff = 3072, 12288
    Mi, Ts = 1024, (512, 384)
    Wi, Wt = bf(torch.randn(ff, d, generator=g) * 0.02).cuda(), bf(torch.randn(ff, d, generator=g) * 0.02).cuda()
    bi, bt = bf(torch.randn(ff, generator=g) * 0.01).cuda(), bf(torch.randn(ff, generator=g) * 0.01).cuda()
    Ai = [bf(torch.randn(Mi, d, generator=g)).cuda() for _ in Ts]
    At = [bf(torch.randn(t, d, generator=g)).cuda() for t in Ts]
    hi = [torch.empty(Mi, ff, dtype=torch.bfloat16, device="cuda") for _ in Ts]
    ht = [torch.empty(t, ff, dtype=torch.bfloat16, device="cuda") for t in Ts]
    ops.gemm_group([ops.Problem(Ai[0], Wi, bi, hi[0]), ops.Problem(At[0], Wt, bt, ht[0]),
                    ops.Problem(Ai[1], Wi, bi, hi[1]), ops.Problem(At[1], Wt, bt, ht[1])], epilogue=ops.EPI_GELU)
    for b in range(2):
        si, st_ = torch.empty_like(hi[b]), torch.empty_like(ht[b])
        ops.gemm_pair(Ai[b], Wi, bi, si, At[b], Wt, bt, st_, epilogue=ops.EPI_GELU)
        if split == "0":
            assert torch.equal(hi[b], si) and torch.equal(ht[b], st_)
        ref = F.gelu(F.linear(At[b].cpu().double(), Wt.cpu().double(), bt.cpu().double()), approximate="tanh")
        assert rel_err(ht[b].cpu(), ref) < 4e-3
    # ff2 with the gated residual (long K: the shape the planner splits)
    W2i, W2t = bf(torch.randn(d, ff, generator=g) * 0.02).cuda(), bf(torch.randn(d, ff, generator=g) * 0.02).cuda()
    b2 = bf(torch.randn(d, generator=g) * 0.01).cuda()
    gates = [bf(torch.randn(d, generator=g)).cuda() for _ in range(4)]
    xi = [bf(torch.randn(Mi, d, generator=g)).cuda() for _ in Ts]
    xt = [bf(torch.randn(t, d, generator=g)).cuda() for t in Ts]
    yi, yt = [x.clone() for x in xi], [x.clone() for x in xt]
    ops.gemm_group([ops.Problem(hi[0], W2i, b2, yi[0], gate=gates[0], resid=yi[0]), ops.Problem(ht[0], W2t, b2, yt[0], gate=gates[1], resid=yt[0]),
                    ops.Problem(hi[1], W2i, b2, yi[1], gate=gates[2], resid=yi[1]), ops.Problem(ht[1], W2t, b2, yt[1], gate=gates[3], resid=yt[1])],
                   epilogue=ops.EPI_GATE_RESID)
    for b in range(2):
        zi, zt = xi[b].clone(), xt[b].clone()
        ops.gemm_pair(hi[b], W2i, b2, zi, ht[b], W2t, b2, zt, epilogue=ops.EPI_GATE_RESID, gate0=gates[2 * b], resid0=zi,
                      gate1=gates[2 * b + 1], resid1=zt)
        if split == "0":
            assert torch.equal(yi[b], zi) and torch.equal(yt[b], zt)
        lin = F.linear(ht[b].cpu().double(), W2t.cpu().double(), b2.cpu().double())
        ref = xt[b].cpu().double() + gates[2 * b + 1].cpu().double() * lin
        assert rel_err(yt[b].cpu(), ref) < 4e-3
        assert rel_err(yi[b].cpu(), zi.cpu().double()) < 2e-3


def test_gemm_group_fused_qkv_epilogue_per_branch_slabs():
    """Q/K/V epilogue with one descriptor per problem: two CFG branches (text lengths 48 / 40) write their own K / V^T slabs
    and take their own rotary rows; bit-identical to one gemm_qkv_pair launch per branch."""
    from regione_amd import ops
    gen = torch.Generator().manual_seed(5)
    H, K, Mi = 2, 256, 400
    D, N = H * 128, 3 * H * 128
    Ts = (48, 40)
    Ai0, W0, b0, wq0, wk0, _, _, _, _ = _qkv_case(Mi, H, K, 0, gen, skv=Ts[0] + Mi)
    At0, W1, b1, wq1, wk1, _, _, _, _ = _qkv_case(Ts[0], H, K, 0, gen)
    Ai = [Ai0, bf(torch.randn(Mi, K, generator=gen)).cuda()]
    At = [At0, bf(torch.randn(Ts[1], K, generator=gen)).cuda()]
    ropes = []
    for T in Ts:
        ang = torch.rand(T + Mi, 64, generator=gen) * 6.28
        ropes.append((ang.cos().repeat_interleave(2, 1).contiguous().cuda(), ang.sin().repeat_interleave(2, 1).contiguous().cuda()))
    grp_out, ref_out, grp_slabs, ref_slabs, probs = [], [], [], [], []
    for b, T in enumerate(Ts):
        pad = ops.padded(T + Mi)
        for outs, slabs in ((grp_out, grp_slabs), (ref_out, ref_slabs)):
            outs.append(torch.zeros(T + Mi, N, dtype=torch.bfloat16, device="cuda"))
            slabs.append((torch.zeros(pad, D, dtype=torch.bfloat16, device="cuda"), torch.zeros(D, pad, dtype=torch.bfloat16, device="cuda")))
        common = dict(rope_q=ropes[b], rope_k=ropes[b], H=H, k_col=0, v_col=D, q_col=2 * D)
        e_i = ops.qkv_epilogue(wq=wq0, wk=wk0, row_base=T, k_slab=grp_slabs[b][0], vt_slab=grp_slabs[b][1], **common)
        e_t = ops.qkv_epilogue(wq=wq1, wk=wk1, row_base=0, k_slab=grp_slabs[b][0], vt_slab=grp_slabs[b][1], **common)
        probs += [ops.Problem(Ai[b], W0, b0, grp_out[b][T:], epi=e_i), ops.Problem(At[b], W1, b1, grp_out[b][:T], epi=e_t)]
        ops.gemm_qkv_pair(Ai[b], W0, b0, ref_out[b][T:], ops.qkv_epilogue(wq=wq0, wk=wk0, row_base=T, k_slab=ref_slabs[b][0],
                                                                        vt_slab=ref_slabs[b][1], **common),
                          At[b], W1, b1, ref_out[b][:T], ops.qkv_epilogue(wq=wq1, wk=wk1, row_base=0, k_slab=ref_slabs[b][0],
                                                                        vt_slab=ref_slabs[b][1], **common))
    ops.gemm_group(probs, epilogue=3)
    for b in range(2):
        assert torch.equal(grp_out[b][:, 2 * D:], ref_out[b][:, 2 * D:])
        assert torch.equal(grp_slabs[b][0], ref_slabs[b][0]) and torch.equal(grp_slabs[b][1], ref_slabs[b][1])
        assert float(grp_slabs[b][0].float().abs().sum()) > 0


def test_ln_modulate_four_segments():
    from regione_amd import ops
    g = torch.Generator().manual_seed(31)
    d = 3072
    ends = [40, 300, 332, 600]
    x = bf(torch.randn(ends[-1], d, generator=g) * 2 + 0.3)
    mods = [(bf(torch.randn(d, generator=g) * 0.5), bf(torch.randn(d, generator=g) * 0.5)) for _ in ends]
    out = torch.empty(ends[-1], d, dtype=torch.bfloat16).cuda()
    ops.ln_modulate_segs(x.cuda(), out, [(e, sh.cuda(), sc.cuda()) for e, (sh, sc) in zip(ends, mods)])
    lo = 0
    for e, (sh, sc) in zip(ends, mods):
        one = torch.empty(e - lo, d, dtype=torch.bfloat16).cuda()
        ops.ln_modulate(x[lo:e].cuda(), one, sh.cuda(), sc.cuda())
        assert torch.equal(out[lo:e], one)                       # a segment == the two-set kernel on those rows
        lo = e
    two = torch.empty_like(out)
    ops.ln_modulate_segs(x.cuda(), two, [(ends[0], mods[0][0].cuda(), mods[0][1].cuda()), (ends[-1], mods[1][0].cuda(), mods[1][1].cuda())])
    ref = torch.empty_like(out)
    ops.ln_modulate(x.cuda(), ref, mods[1][0].cuda(), mods[1][1].cuda(), split_row=ends[0], shift0=mods[0][0].cuda(), scale0=mods[0][1].cuda())
    assert torch.equal(two, ref)



@pytest.mark.parametrize("M,N,K,epi", [(4608, 3072, 3072, "gate"), (8704, 21504, 3072, "gelu"), (2048, 704, 15360, "bias"),
                                       (1536, 3072, 15360, "gate"), (600, 520, 256, "bias"), (9216, 9216, 3072, "bias")])
def test_gemm_fp8_weights_in_the_hand_scheduled_loop_bit_identical_to_the_fp8_tile_kernel(M, N, K, epi):
    """Round 3: fp8 (e4m3fn) weight tiles INSIDE the hand-scheduled 4-wave K loop (W ring of three 16 KiB byte slots,
    ds_read_b64 + in-register v_cvt_scalef32_pk_bf16_fp8 ahead of the MFMAs; the default for 256 x 256 tiles) against the
    compiler-scheduled fp8-tile kernel (gemm_asm = 0: the fallback for K < 256 / operands >= 4 GiB): the conversion is exact and the
    MFMA order per output element is the same, so both agree bit for bit."""
    from regione_amd import ops
    g = torch.Generator().manual_seed(M + N + K)
    A = bf(torch.randn(M, K, generator=g)).cuda()
    Wq = ops.quantize_w8((torch.randn(N, K, generator=g) * 0.05).bfloat16().cuda())
    b = bf(torch.randn(N, generator=g)).cuda()
    gate, x = bf(torch.randn(N, generator=g)).cuda(), bf(torch.randn(M, N, generator=g)).cuda()
    P.force(gemm_pieces=1, gemm_geometry=256)      # whole-K 256 x 256 tiles on both paths: the planners may cut remainders differently
    outs = []
    for asm in (-1, 0):
        P.force(gemm_asm=asm)
        if epi == "gate":
            o = x.clone()
            ops.gemm(A, Wq, b, o, epilogue=ops.EPI_GATE_RESID, gate=gate, resid=o)
        else:
            o = torch.full((M, N), 3.0, dtype=torch.bfloat16, device="cuda")
            ops.gemm(A, Wq, b, o, epilogue=ops.EPI_GELU if epi == "gelu" else ops.EPI_BIAS, gelu_from_col=N // 2 // 8 * 8)
        torch.cuda.synchronize()
        outs.append(o)
    assert torch.equal(outs[0], outs[1]), float((outs[0].float() - outs[1].float()).abs().max())
    W = (Wq.float() * Wq._rgn_scale[:, None]).cpu()
    if epi == "bias":
        ref = F.linear(A.cpu().float(), W, b.cpu().float())
        assert rel_err(outs[0].cpu(), ref) < 4e-3


@pytest.mark.parametrize("nsplit", [2, 3])
def test_gemm_fp8_split_k_pieces_in_the_hand_scheduled_loop(nsplit):
    """Split-K remainder pieces on fp8 weights run the fp8 asm loop too (pieces of >= 4 K tiles): same result as the
    compiler-scheduled fp8 pieces at the same piece count."""
    from regione_amd import ops
    M, N, K = 1536, 3072, 15360
    g = torch.Generator().manual_seed(77)
    A = bf(torch.randn(M, K, generator=g)).cuda()
    Wq = ops.quantize_w8((torch.randn(N, K, generator=g) * 0.05).bfloat16().cuda())
    b = bf(torch.randn(N, generator=g)).cuda()
    P.force(gemm_pieces=nsplit, gemm_geometry=256)
    outs = []
    for asm in (-1, 0):
        P.force(gemm_asm=asm)
        o = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
        ops.gemm(A, Wq, b, o)
        torch.cuda.synchronize()
        outs.append(o)
    assert torch.equal(outs[0], outs[1])
    ref = F.linear(A.cpu().float(), (Wq.float() * Wq._rgn_scale[:, None]).cpu(), b.cpu().float())
    assert rel_err(outs[0].cpu(), ref) < 4e-3



@pytest.mark.parametrize("case", ["gelu_576", "gate_288", "bias_ragged", "group_qkv", "fp8_gelu"])
def test_gemm_quarter_tile_remainder_bit_identical_to_one_plain_launch(case):
    """Round 3: the tiles of a launch that do not fill a whole round of the 256 workgroup slots run as QUADRANTS (128 x 128 blocks,
    two per CU) of the same tile list instead of being cut along K - no partials, no reduce pass, and per output element the same
    accumulation order as an unsplit launch: gemm_quarter = 1 (forced) == gemm_pieces = 1 (one plain launch), bit for bit, for
    every epilogue, ragged edges, the fused Q/K/V epilogue of a two-branch group with gathered cache rows, and fp8 weights."""
    from regione_amd import ops
    g = torch.Generator().manual_seed(len(case))

    def run():
        if case in ("gelu_576", "fp8_gelu"):
            M, N, K = 2944, 12288, 3072                       # 576 tiles = 2 rounds + 64
            A, b = bf(torch.randn(M, K, generator=g)).cuda(), bf(torch.randn(N, generator=g)).cuda()
            W = bf(torch.randn(N, K, generator=g) * 0.05).cuda()
            if case == "fp8_gelu":
                W = ops.quantize_w8(W)
            return (A, W, b), lambda o: ops.gemm(A, W, b, o, epilogue=ops.EPI_GELU), (M, N)
        if case == "gate_288":
            M, N, K = 1536, 12288, 3072                       # 288 tiles = 1 round + 32
            A, b = bf(torch.randn(M, K, generator=g)).cuda(), bf(torch.randn(N, generator=g)).cuda()
            W, gate = bf(torch.randn(N, K, generator=g) * 0.05).cuda(), bf(torch.randn(N, generator=g)).cuda()
            x = bf(torch.randn(M, N, generator=g)).cuda()
            return None, lambda o: (o.copy_(x), ops.gemm(A, W, b, o, epilogue=ops.EPI_GATE_RESID, gate=gate, resid=o)), (M, N)
        if case == "bias_ragged":
            M, N, K = 2377, 9100 // 8 * 8, 1024               # ragged rows / columns: quadrants past the edge exit early
            A, b = bf(torch.randn(M, K, generator=g)).cuda(), bf(torch.randn(N, generator=g)).cuda()
            W = bf(torch.randn(N, K, generator=g) * 0.05).cuda()
            return None, lambda o: ops.gemm(A, W, b, o), (M, N)
        raise AssertionError
    if case != "group_qkv":
        _, fn, (M, N) = run()
        outs = []
        from regione_amd import _lib
        for env in (dict(gemm_quarter=1), dict(gemm_pieces=1, gemm_quarter=0)):
            with _lib.plan_override(**env):
                o = torch.full((M, N), 5.0, dtype=torch.bfloat16, device="cuda")
                fn(o)
                torch.cuda.synchronize()
                outs.append(o)
                plan = _lib.lib().rgn_gemm_last_plan()
            assert bool(plan & 0x100) == ("gemm_pieces" not in env), f"unexpected launch plan {plan:#x} under {env}"
        assert torch.equal(outs[0], outs[1]) and torch.isfinite(outs[0].float()).all()
        return
    # two CFG branches x (image, text) problems, fused Q/K/V epilogue, gathered cache rows on the image problems
    H, K = 24, 3072
    D, N = H * 128, 3 * H * 128
    Mi, Ts, S = 1024, (512, 384), 4096
    Wi, Wt = bf(torch.randn(N, K, generator=g) * 0.02).cuda(), bf(torch.randn(N, K, generator=g) * 0.02).cuda()
    bi = bf(torch.randn(N, generator=g) * 0.1).cuda()
    wq, wk = bf(1 + 0.1 * torch.randn(128, generator=g)).cuda(), bf(1 + 0.1 * torch.randn(128, generator=g)).cuda()
    Ai = [bf(torch.randn(Mi, K, generator=g)).cuda() for _ in Ts]
    At = [bf(torch.randn(t, K, generator=g)).cuda() for t in Ts]
    res = []
    for env in (dict(gemm_quarter=1), dict(gemm_pieces=1, gemm_quarter=0)):
        P.reset()
        P.force(**env)
        probs, keep = [], []
        for b, T in enumerate(Ts):
            skv = T + S
            pad = ops.padded(skv)
            ang = torch.rand(skv, 64, generator=torch.Generator().manual_seed(b)) * 6.28
            rope = (ang.cos().repeat_interleave(2, 1).contiguous().cuda(), ang.sin().repeat_interleave(2, 1).contiguous().cuda())
            ids = torch.randperm(S, generator=torch.Generator().manual_seed(7 + b))[:Mi].sort().values
            rows = torch.cat([torch.arange(T), T + ids]).cuda()
            rq = (rope[0][rows].contiguous(), rope[1][rows].contiguous())
            ks, vs = torch.zeros(pad, D, dtype=torch.bfloat16, device="cuda"), torch.zeros(D, pad, dtype=torch.bfloat16, device="cuda")
            out = torch.zeros(T + Mi, N, dtype=torch.bfloat16, device="cuda")
            common = dict(wq=wq, wk=wk, rope_q=rq, rope_k=rope, k_slab=ks, vt_slab=vs, H=H, k_col=0, v_col=D, q_col=2 * D, kv_rows=rows)
            probs += [ops.Problem(Ai[b], Wi, bi, out[T:], epi=ops.qkv_epilogue(row_base=T, fp16_roundtrip=True, **common)),
                      ops.Problem(At[b], Wt, bi, out[:T], epi=ops.qkv_epilogue(row_base=0, **common))]
            keep.append((out, ks, vs))
        ops.gemm_group(probs, epilogue=3)
        torch.cuda.synchronize()
        res.append(keep)
    for (o0, k0, v0), (o1, k1, v1) in zip(*res):
        assert torch.equal(o0[:, 2 * D:], o1[:, 2 * D:]) and torch.equal(k0, k1) and torch.equal(v0, v1)
        assert float(k0.float().abs().sum()) > 0


@pytest.mark.parametrize("Sq,Skv,H,w", [(256, 1024, 2, 1.0), (1536, 8704, 24, 1.3), (8704, 8704, 24, 1.0), (1408, 8576, 24, 2.5)])
def test_attention_with_a_caller_guaranteed_score_bound_needs_no_running_max(Sq, Skv, H, w):
    """rgn_attention_bounded (round 3): q and k RMS-normalised per head and scaled by weights of magnitude <= w, so
    |q . k| / sqrt(128) <= sqrt(128) * w^2; with that bound the hand-scheduled kernel keeps no running row maximum
    (P = exp2(s * log2 e) directly).  Same softmax: against the fp32 reference < 1e-2 relative, and against the tracked-max
    kernel (score_bound = 0) to bf16 rounding.  w = 2.5 -> bound * log2(e) = 102 > 96: the call falls back to the running
    max by itself (bit-identical to the unbounded call)."""
    from regione_amd import ops
    g = torch.Generator().manual_seed(Sq + H)
    D = H * 128

    def normed(n):
        x = torch.randn(n, H, 128, generator=g)
        x = x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True))
        wt = (torch.rand(128, generator=g) * 2 - 1) * w
        return bf((x * wt).reshape(n, D))
    q, k = normed(Sq), normed(Skv)
    v = bf(torch.randn(Skv, D, generator=g))
    bound = 1.05 * 128 * w * w / math.sqrt(128.0)
    pad = ops.padded(Skv)
    ks = torch.zeros(pad, D, dtype=torch.bfloat16)
    ks[:Skv] = k
    r = torch.arange(Skv)
    pos = (r & ~12) | ((r & 4) << 1) | ((r & 8) >> 1)
    vt = torch.zeros(D, pad, dtype=torch.bfloat16)
    vt[:, pos] = v.T
    qc, kc, vc = q.cuda(), ks.cuda(), vt.cuda()
    outs = {}
    for name, b in (("static", bound), ("tracked", 0.0)):          # score_bound = 0: the running-max loop
        o = torch.empty(Sq, D, dtype=torch.bfloat16).cuda()
        ops.attention(qc, kc, vc, o, Skv, H, score_bound=b)
        torch.cuda.synchronize()
        outs[name] = o.cpu()
    qq, kk, vv = (t.cuda().float().view(-1, H, 128).transpose(0, 1) for t in (q, k, v))
    s = torch.einsum("hqd,hkd->hqk", qq, kk) / math.sqrt(128.0)
    assert float(s.abs().max()) <= bound / 1.05 + 1e-3                   # the guarantee holds on these inputs
    ref = F.scaled_dot_product_attention(qq[None], kk[None], vv[None])[0].transpose(0, 1).reshape(Sq, D).cpu()
    assert torch.isfinite(outs["static"].float()).all()
    assert rel_err(outs["static"], ref) < 1e-2 and rel_err(outs["tracked"], ref) < 1e-2
    assert rel_err(outs["static"], outs["tracked"].double()) < 5e-3
    if w == 2.5:
        assert torch.equal(outs["static"], outs["tracked"])              # bound too large: the running max is kept


@pytest.mark.parametrize("variant", ["4", "8"])
def test_attention_static_shift_at_the_ends_of_its_dynamic_range(variant):
    """The rows the bounded softmax is argued safe for (attn.hip: static_m = 0, P = exp2(s * log2 e) within 2^+-96), which random
    draws never produce: with bound * log2(e) = 95.9, (i) a query ALIGNED with one key (s = +bound) and
    ANTI-ALIGNED with another (s = -bound) in the same row - P spans 2^+95.9 ... 2^-95.9 in one row sum; (ii) the mirrored query;
    (iii) a head in which EVERY score is -bound (all keys identical, queries opposite): every P is 2^-95.9, the row sum
    1024 * 2^-95.9 must not vanish and the output is the plain mean of V.  Checked against the fp32 softmax and against the
    tracked-max kernel; all finite."""
    from regione_amd import ops
    P.force(attn_waves=int(variant))          # 8 = the 8-wave hand-scheduled kernel the pipeline runs, 4 = tiny query sets
    g = torch.Generator().manual_seed(5)
    Sq, Skv, H = 256, 1024, 2
    D = H * 128
    bound = 95.9 / 1.4426950408889634                     # |s| <= bound  <=>  |s * log2 e| <= 95.9 (static shift engaged: <= 96)
    c = math.sqrt(bound * math.sqrt(128.0) / 128.0)       # q = k = c * u, u_i = +-1:  q . k / sqrt(128) = c^2 * 128 / sqrt(128)
    u = torch.where(torch.rand(128, generator=g) < 0.5, -1.0, 1.0)
    q = torch.randn(Sq, H, 128, generator=g) * 0.3
    k = torch.randn(Skv, H, 128, generator=g) * 0.3
    v = torch.randn(Skv, H, 128, generator=g)
    q[0, 0], q[1, 0] = c * u, -c * u                      # head 0: rows 0 / 1 against keys 0 / 1
    k[0, 0], k[1, 0] = c * u, -c * u
    q[:, 1] = -c * u                                      # head 1: every score = -bound
    k[:, 1] = c * u
    q, k, v = bf(q.reshape(Sq, D)), bf(k.reshape(Skv, D)), bf(v.reshape(Skv, D))
    qq, kk, vv = (t.float().view(-1, H, 128).transpose(0, 1) for t in (q, k, v))
    s = torch.einsum("hqd,hkd->hqk", qq, kk) / math.sqrt(128.0)
    assert float(s.abs().max()) <= bound and float(s[0, 0, 0]) > 0.99 * bound and float(s[0, 0, 1]) < -0.99 * bound
    assert float(s[1].max()) < -0.99 * bound
    pad = ops.padded(Skv)
    ks = torch.zeros(pad, D, dtype=torch.bfloat16)
    ks[:Skv] = k
    r = torch.arange(Skv)
    pos = (r & ~12) | ((r & 4) << 1) | ((r & 8) >> 1)
    vt = torch.zeros(D, pad, dtype=torch.bfloat16)
    vt[:, pos] = v.T
    outs = {}
    for name, sb in (("static", bound), ("tracked", 0.0)):          # score_bound = 0: the running-max loop
        o = torch.empty(Sq, D, dtype=torch.bfloat16).cuda()
        ops.attention(q.cuda(), ks.cuda(), vt.cuda(), o, Skv, H, score_bound=sb)
        torch.cuda.synchronize()
        outs[name] = o.cpu()
    ref = torch.einsum("hqk,hkd->hqd", torch.softmax(s.double(), -1), vv.double()).transpose(0, 1).reshape(Sq, D)
    for name, o in outs.items():
        assert torch.isfinite(o.float()).all(), name
        assert rel_err(o, ref) < 1e-2, (name, rel_err(o, ref))
        # the two crafted rows pick out ONE value row each; the all-negative head returns the mean of V
        assert rel_err(o[0, :128], vv[0, 0].double()) < 1e-2 and rel_err(o[1, :128], vv[0, 1].double()) < 1e-2, name
        assert rel_err(o[:, 128:], vv[1].double().mean(0).expand(Sq, 128)) < 2e-2, name
    assert rel_err(outs["static"], outs["tracked"].double()) < 5e-3


@pytest.mark.parametrize("Ms,N,K", [((1137,), 3072, 15360), ((625, 512), 3072, 12288), ((1056,), 3072, 15360), ((448, 512), 3072, 12288)])
def test_gemm_planner_long_k_region_shapes_take_the_256_geometry_with_split_k(Ms, N, K):
    """Round 4 (tools/probes/plan_sweep.py --dense): FLUX proj_out / FF-down at K_e 8 ... 18 % are 48 ... 60 tiles of 256 x 256 with
    K = 12288 / 15360.  The round-3 cost model priced the 128 x 128 geometry (and the quarter-tile remainder built on it) at its
    one-block-per-CU rate although 190+ blocks are co-located two per CU, and took it: 212 us where 256 x 256 + split-K runs in 117.
    The planner must now pick the 256 geometry with 3 ... 5 K pieces for them (rgn_gemm_last_plan: bit 10 = 256 geometry, bits 0-7 =
    pieces, bit 8 = quarter), and the result equals the unsplit launch to rounding."""
    from regione_amd import ops
    g = torch.Generator().manual_seed(sum(Ms) + K)
    As = [bf(torch.randn(m, K, generator=g)).cuda() for m in Ms]
    Ws = [bf(torch.randn(N, K, generator=g) * 0.05).cuda() for _ in Ms]
    b, gate = bf(torch.randn(N, generator=g)).cuda(), bf(torch.randn(N, generator=g)).cuda()
    xs = [bf(torch.randn(m, N, generator=g)).cuda() for m in Ms]

    def run():
        outs = [x.clone() for x in xs]
        ops.gemm_group([ops.Problem(a, w, b, o, gate=gate, resid=o) for a, w, o in zip(As, Ws, outs)], epilogue=ops.EPI_GATE_RESID)
        torch.cuda.synchronize()
        return torch.cat(outs), ops._lib.lib().rgn_gemm_last_plan()
    out, plan = run()
    assert plan & (1 << 10), f"plan {plan:#x}: 128 geometry"
    assert not plan & (1 << 8), f"plan {plan:#x}: quarter-tile remainder"
    assert 3 <= (plan & 255) <= 5, plan & 255
    with ops._lib.plan_override(gemm_pieces=1):
        whole, _ = run()
    assert rel_err(out.cpu(), whole.cpu()) < 2e-3
