#!/usr/bin/env python3
"""Sustained GEMM rate vs operand statistics and weight residency (GPU box only): the in-pipeline GEMMs read
≈ 15 % below tools/bench_kernels.py - is it the data (power), cold weights, or the sustained power state?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from regione_amd import ops

M, N, K = 8704, 21504, 3072


def run(tag, A, Ws, b, secs=2.0):
    out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    for W in Ws[:2]:
        ops.gemm(A, W, b, out)
    torch.cuda.synchronize()
    n = 0
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    t0 = time.time()
    while time.time() - t0 < secs:
        for W in Ws:
            ops.gemm(A, W, b, out)
            n += 1
        torch.cuda.synchronize()
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / n
    print(f"{tag:<46} {ms*1e3:8.1f} us  {2.0*M*N*K/ms/1e9:7.1f} TF  ({n} launches)")


g = torch.Generator(device="cuda").manual_seed(0)
uni = lambda *s: (torch.rand(*s, device="cuda", generator=g) * 2 - 1).to(torch.bfloat16)
nrm = lambda std, *s: (torch.randn(*s, device="cuda", generator=g) * std).to(torch.bfloat16)
b = nrm(0.01, N)
run("uniform A, uniform*0.05 W, 1 weight buffer", uni(M, K), [uni(N, K) * 0.05], b)
run("N(0,1) A, N(0,0.02) W, 1 weight buffer", nrm(1.0, M, K), [nrm(0.02, N, K)], b)
run("N(0,1) A, N(0,0.02) W, 24 weight buffers (cold)", nrm(1.0, M, K), [nrm(0.02, N, K) for _ in range(24)], b)
run("zeros", torch.zeros(M, K, dtype=torch.bfloat16, device="cuda"), [torch.zeros(N, K, dtype=torch.bfloat16, device="cuda")], b)
run("N(0,1) A, N(0,0.02) W, 24 buffers, 6 s sustained", nrm(1.0, M, K), [nrm(0.02, N, K) for _ in range(24)], b, secs=6.0)

# ---- epilogue cost on the two single-block GEMMs ------------------------------------------------------------
def timeit(fn, inner=25, rounds=5):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(rounds):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(inner):
            fn()
        e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) / inner)
    return sorted(ts)[len(ts) // 2]


A, W = nrm(1.0, M, K), nrm(0.02, N, K)
out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
t0 = timeit(lambda: ops.gemm(A, W, b, out))
t1 = timeit(lambda: ops.gemm(A, W, b, out, epilogue=ops.EPI_GELU, gelu_from_col=9216))
D, H = 3072, 24
ang = torch.rand(M, 64, device="cuda") * 6.28
rope = (torch.repeat_interleave(torch.cos(ang), 2, 1).contiguous(), torch.repeat_interleave(torch.sin(ang), 2, 1).contiguous())
ks = torch.zeros(ops.padded(M), D, dtype=torch.bfloat16, device="cuda"); vs = torch.zeros(D, ops.padded(M), dtype=torch.bfloat16, device="cuda")
wq = torch.ones(128, dtype=torch.bfloat16, device="cuda")
epi = ops.qkv_epilogue(wq=wq, wk=wq, rope_q=rope, rope_k=rope, k_slab=ks, vt_slab=vs, H=H, k_col=0, v_col=D, q_col=2 * D)
t2 = timeit(lambda: ops.gemm_qkv(A, W, b, out, epi, gelu_from_col=9216))
print(f"kvq+mlp 8704x21504x3072: bias {t0*1e3:.1f} us | +GELU(mlp cols) {t1*1e3:.1f} us | fused QKV+GELU {t2*1e3:.1f} us")
M2, N2, K2 = 8704, 3072, 15360
A2, W2, b2 = nrm(1.0, M2, K2), nrm(0.02, N2, K2), nrm(0.01, N2)
x = nrm(1.0, M2, N2); gate = nrm(1.0, N2)
o2 = torch.empty(M2, N2, dtype=torch.bfloat16, device="cuda")
t3 = timeit(lambda: ops.gemm(A2, W2, b2, o2))
t4 = timeit(lambda: ops.gemm(A2, W2, b2, x, epilogue=ops.EPI_GATE_RESID, gate=gate, resid=x))
print(f"proj_out 8704x3072x15360: bias {t3*1e3:.1f} us | gate+residual {t4*1e3:.1f} us")
