#!/usr/bin/env python3
"""Region-step attention launch (Sq = T + K_e against the full cache) in isolation, several K/V slabs in rotation: time per launch.
    [RGN_LIB=...alt.so] python tools/probes/attn_region_probe.py [Sq ...]"""
import hashlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from regione_amd import ops, _lib
from bench_kernels import timeit, rnd

torch.manual_seed(0)
H, Skv = 24, 8704
D = H * 128
slabs = [(rnd(Skv, D), rnd(D, Skv)) for _ in range(4)]
for Sq in [int(x) for x in sys.argv[1:]] or [708, 1137, 1536, 2537]:
    q = rnd(Sq, D)
    out = torch.empty_like(q)
    i = [0]

    def run():
        i[0] += 1
        k, vt = slabs[i[0] % 4]
        ops.attention(q, k, vt, out, Skv, H, score_bound=20.0)
    med, best = timeit(run)
    fl = 4.0 * Sq * Skv * D
    # a digest of one launch's output: two builds that must agree bit for bit print the same word
    ops.attention(q, slabs[0][0], slabs[0][1], out, Skv, H, score_bound=20.0)
    torch.cuda.synchronize()
    digest = hashlib.sha256(out.view(torch.int16).cpu().numpy().tobytes()).hexdigest()[:12]
    print(f"Sq={Sq:<5} {med * 1e3:8.1f} us  {fl / med / 1e9:7.1f} TF (best {fl / best / 1e9:7.1f})  plan {_lib.lib().rgn_attention_last_plan():#x}  out {digest}", flush=True)
