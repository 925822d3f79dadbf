#!/usr/bin/env python3
"""Which vendor (hipBLASLt / Tensile) kernels run the FLUX GEMM shapes - reference point for DESIGN.md.
    rocprofv3 --kernel-trace --stats -d out -- python tools/lib_gemm_probe.py"""
import torch
rnd = lambda *s: (torch.rand(*s, device="cuda") * 2 - 1).to(torch.bfloat16)
for M, N, K in [(8704, 21504, 3072), (8704, 3072, 15360), (8192, 12288, 3072), (8192, 8192, 8192)]:
    A, W, b = rnd(M, K), rnd(N, K) * 0.05, rnd(N)
    out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    for _ in range(5):
        torch.addmm(b, A, W.t(), out=out)
    torch.cuda.synchronize()
