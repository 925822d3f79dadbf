import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tools"))
from regione_amd import ops
from bench_kernels import rnd
M, N, K = 1536, 3072, 15360
A, b, gate, x = rnd(M, K), rnd(N), rnd(N), rnd(M, N)
Ws = [rnd(N, K) * 0.05 for _ in range(6)]
for mode in ("1", "0"):
    os.environ["RGN_GEMM_FINE_REDUCE"] = mode
    for i in range(60):
        ops.gemm(A, Ws[i % 6], b, x, epilogue=ops.EPI_GATE_RESID, gate=gate, resid=x)
    torch.cuda.synchronize()
