#!/usr/bin/env python3
"""What each fused epilogue costs on the full-step FLUX projections (GPU box only), same box, sustained rate:
the plain bias epilogue against the epilogue the pipeline actually runs (GELU / gated residual / fused Q-K-V)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from regione_amd import ops
from bench_kernels import timeit, rnd

H, D = 24, 3072


def line(name, variant, M, N, K, t):
    med, best = t
    fl = 2.0 * M * N * K
    print(f"{name:<16} {variant:<22} {med*1e3:8.1f} us  {fl/med/1e9:7.1f} TF (best {fl/best/1e9:7.1f})", flush=True)


def qkv_setup(M, N, K):
    A, W, b = rnd(M, K), rnd(N, K) * 0.05, rnd(N)
    wq, wk = rnd(128) * 0.1 + 1, rnd(128) * 0.1 + 1
    ang = torch.rand(M, 64, device="cuda") * 6.28
    rope = (torch.repeat_interleave(torch.cos(ang), 2, dim=1).contiguous(), torch.repeat_interleave(torch.sin(ang), 2, dim=1).contiguous())
    skv = ops.padded(M)
    ks = torch.zeros(skv, D, dtype=torch.bfloat16, device="cuda")
    vs = torch.zeros(D, skv, dtype=torch.bfloat16, device="cuda")
    out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    epi = ops.qkv_epilogue(wq=wq, wk=wk, rope_q=rope, rope_k=rope, k_slab=ks, vt_slab=vs, H=H, k_col=0, v_col=D, q_col=2 * D)
    return A, W, b, out, epi, (wq, wk, rope, ks, vs)


def main():
    M = 8704
    # single block: [K;V;Q;mlp] projection
    N, K = 21504, 3072
    A, W, b, out, epi, keep = qkv_setup(M, N, K)
    line("kvq+mlp", "bias", M, N, K, timeit(lambda: ops.gemm(A, W, b, out)))
    line("kvq+mlp", "gelu(mlp half)", M, N, K, timeit(lambda: ops.gemm(A, W, b, out, epilogue=ops.EPI_GELU, gelu_from_col=3 * D)))
    line("kvq+mlp", "fused qkv + gelu", M, N, K, timeit(lambda: ops.gemm_qkv(A, W, b, out, epi, gelu_from_col=3 * D)))
    del A, W, out, epi, keep
    N, K = 9216, 3072
    A, W, b, out, epi, keep = qkv_setup(M, N, K)
    line("qkv", "bias", M, N, K, timeit(lambda: ops.gemm(A, W, b, out)))
    line("qkv", "fused qkv", M, N, K, timeit(lambda: ops.gemm_qkv(A, W, b, out, epi)))
    del A, W, out, epi, keep
    for name, N, K in (("proj_out", 3072, 15360), ("ff2", 3072, 12288), ("attn out", 3072, 3072)):
        A, W, b = rnd(M, K), rnd(N, K) * 0.05, rnd(N)
        out, x, gate = torch.empty(M, N, dtype=torch.bfloat16, device="cuda"), rnd(M, N), rnd(N)
        line(name, "bias", M, N, K, timeit(lambda: ops.gemm(A, W, b, out)))
        line(name, "gate*out + resid", M, N, K, timeit(lambda: ops.gemm(A, W, b, x, epilogue=ops.EPI_GATE_RESID, gate=gate, resid=x)))
    N, K = 12288, 3072
    A, W, b = rnd(M, K), rnd(N, K) * 0.05, rnd(N)
    out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    line("ff1", "bias", M, N, K, timeit(lambda: ops.gemm(A, W, b, out)))
    line("ff1", "gelu", M, N, K, timeit(lambda: ops.gemm(A, W, b, out, epilogue=ops.EPI_GELU)))


if __name__ == "__main__":
    main()
