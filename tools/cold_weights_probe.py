#!/usr/bin/env python3
"""Hot vs cold weights on the region-step GEMM shapes (GPU box only): in the pipeline every layer brings its own
weights from HBM, the kernel micro-benchmark re-uses one buffer (L2 / Infinity-Cache resident)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from regione_amd import ops

rnd = lambda *s: (torch.rand(*s, device="cuda") * 2 - 1).to(torch.bfloat16)


def run(name, M, N, K, nbuf):
    A, b = rnd(M, K), rnd(N)
    Ws = [rnd(N, K) * 0.05 for _ in range(nbuf)]
    out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    for W in Ws:
        ops.gemm(A, W, b, out)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = max(1, 200 // nbuf)
    s.record()
    for _ in range(reps):
        for W in Ws:
            ops.gemm(A, W, b, out)
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) / (reps * nbuf) * 1e3
    print(f"{name:<22} M={M:<5} N={N:<6} K={K:<6} weights x{nbuf:<3} {us:8.1f} us  {2.0*M*N*K/us/1e6:7.1f} TF")


for name, M, N, K in [("R proj_out", 1536, 3072, 15360), ("R kvq+mlp", 1536, 21504, 3072), ("R img out", 1536, 3072, 3072),
                      ("R ff2", 1536, 3072, 12288), ("F proj_out", 8704, 3072, 15360), ("F kvq+mlp", 8704, 21504, 3072)]:
    nb = max(2, int(3e9 // (N * K * 2)))            # ~3 GB of distinct weights: far beyond L2 + Infinity Cache
    run(name, M, N, K, 1)
    run(name, M, N, K, min(nb, 40))
