"""Cost of the fused Q/K/V epilogue on the single-stream block's projection (FLUX full step: 8704 x 21504 x 3072, columns
[K | V | Q | MLP]) against the same GEMM with the plain GELU epilogue; cold weights (a different copy of W per launch).
    python tools/probes/qkv_epi_bench.py [full|region]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from regione_amd import ops  # noqa: E402
from tools.bench_kernels import timeit, rnd  # noqa: E402


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "full"
    H, d, K = 24, 3072, 3072
    N = 3 * d + 4 * d
    T, L = 512, 8192
    skv = T + L
    M = skv if which == "full" else T + 1024
    ncopy = 6
    Ws = [rnd(N, K) * 0.05 for _ in range(ncopy)]
    b = rnd(N) * 0.1
    x = rnd(M, K)
    out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    g = torch.Generator(device="cuda").manual_seed(0)
    ang = torch.rand(skv, 64, generator=g, device="cuda") * 6.28
    cos = torch.repeat_interleave(torch.cos(ang), 2, 1).contiguous()
    sin = torch.repeat_interleave(torch.sin(ang), 2, 1).contiguous()
    nq, nk = torch.ones(128, dtype=torch.bfloat16, device="cuda"), torch.ones(128, dtype=torch.bfloat16, device="cuda")
    skv_pad = ops.padded(skv)
    kc = torch.zeros(skv_pad, d, dtype=torch.bfloat16, device="cuda")
    vc = torch.zeros(d, skv_pad, dtype=torch.bfloat16, device="cuda")
    if which == "full":
        kv_rows, rq = None, (cos, sin)
    else:
        ids = torch.randperm(L, device="cuda")[:1024].sort().values + T
        kv_rows = torch.cat([torch.arange(T, device="cuda"), ids])
        rq = (cos[kv_rows].contiguous(), sin[kv_rows].contiguous())
    epi = ops.qkv_epilogue(wq=nq, wk=nk, rope_q=rq, rope_k=(cos, sin), k_slab=kc, vt_slab=vc, H=H, k_col=0, v_col=d, q_col=2 * d,
                           kv_rows=kv_rows)
    i = [0]

    def plain():
        i[0] += 1
        ops.gemm(x, Ws[i[0] % ncopy], b, out, epilogue=ops.EPI_GELU, gelu_from_col=3 * d)

    def fused():
        i[0] += 1
        ops.gemm_qkv(x, Ws[i[0] % ncopy], b, out, epi, gelu_from_col=3 * d)

    fl = 2.0 * M * N * K
    for name, fn in (("plain GELU", plain), ("fused QKV ", fused), ("plain GELU", plain), ("fused QKV ", fused)):
        med, best = timeit(fn)
        print(f"{which} {name} M={M}: {med * 1e3:8.1f} us  {fl / med / 1e9:7.1f} TF (best {fl / best / 1e9:7.1f})")


if __name__ == "__main__":
    main()
