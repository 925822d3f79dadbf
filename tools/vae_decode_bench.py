#!/usr/bin/env python3
"""f4: time the HIP VAE decode of a 128 x 128 x 16 latent (-> 1024 x 1024) next to the eager bf16 PyTorch module of the same architecture
(GPU box only).  `rocprofv3 --kernel-trace --stats -- python tools/vae_decode_bench.py --reps 5 --no-eager` gives the per-kernel split.
    python tools/vae_decode_bench.py [--reps 10] [--no-eager] [--size 128]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from regione_amd import vae as V
from tests import host_vae


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--size", type=int, default=128, help="latent height = width")
    ap.add_argument("--no-eager", action="store_true")
    ns = ap.parse_args()
    m = host_vae.seeded(5)
    dec = V.HipVaeDecoder(m.state_dict(), "cuda")
    z = torch.randn(1, 16, ns.size, ns.size, generator=torch.Generator().manual_seed(1)).cuda()
    for _ in range(3):
        dec.decode(z)
    torch.cuda.synchronize()
    ts = []
    for _ in range(ns.reps):
        t0 = time.perf_counter()
        dec.decode(z)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    res = {"latent": [ns.size, ns.size], "hip_decode_ms_median": 1e3 * ts[len(ts) // 2], "hip_decode_ms_min": 1e3 * ts[0],
           "algorithmic_tflop": dec.flops(ns.size, ns.size) / 1e12}
    res["hip_tflops"] = res["algorithmic_tflop"] / (res["hip_decode_ms_median"] * 1e-3)
    if not ns.no_eager:
        mb = m.cuda().to(torch.bfloat16)
        with torch.no_grad():
            for _ in range(2):
                mb.decode(z.bfloat16(), return_dict=False)
            torch.cuda.synchronize()
            te = []
            for _ in range(5):
                t0 = time.perf_counter()
                mb.decode(z.bfloat16(), return_dict=False)
                torch.cuda.synchronize()
                te.append(time.perf_counter() - t0)
        res["eager_bf16_decode_ms_median"] = 1e3 * sorted(te)[2]
    print(json.dumps(res))


if __name__ == "__main__":
    main()
