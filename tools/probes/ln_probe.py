#!/usr/bin/env python3
"""ln_modulate (AdaLN-Zero modulate, HBM-bound) in isolation: us per launch and TB/s at the full-step and region-step row counts."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from regione_amd import ops
from bench_kernels import timeit, rnd

d = 3072
for M in (8704, 1536, 708):
    xs = [rnd(M, d) for _ in range(6)]
    out = torch.empty(M, d, dtype=torch.bfloat16, device="cuda")
    sh, sc = rnd(d), rnd(d)
    i = [0]

    def run():
        i[0] += 1
        ops.ln_modulate(xs[i[0] % 6], out, sh, sc, split_row=512, shift0=sc, scale0=sh)
    med, best = timeit(run)
    print(f"ln_modulate M={M}: {med * 1e3:7.1f} us  {2 * M * d * 2 / med / 1e9:6.2f} TB/s", flush=True)
