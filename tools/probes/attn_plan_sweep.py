#!/usr/bin/env python3
"""Does attention_schedule (attn.hip) pick the fastest remainder schedule for the region-step query counts?  For Sq = T + K_e across K_e
and the KV lengths of the three resolutions, time the schedules the library can be switched into: auto | equal KV split only
(attn_streamk = 0) | stream-K forced (= 1) | no split (attn_split = 0).  Rotating K/V slabs; GPU box only, measurement tool."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402

from regione_amd import ops  # noqa: E402
from bench_kernels import timeit, rnd, force  # noqa: E402

H = 24
D = H * 128
MODES = [("auto", {}), ("equal split", dict(attn_streamk=0)), ("stream-K", dict(attn_streamk=1)), ("no split", dict(attn_waves=8, attn_split=0))]


def main():
    for Skv, label in ((8704, "1024^2"), (2560, "512^2")):
        slabs = [(rnd(Skv, D), rnd(D, Skv)) for _ in range(4)]
        for ke in (64, 196, 320, 448, 625, 768, 896, 1024, 1280, 1536, 2025, 2560, 3072):
            if ke > (Skv - 512) // 2:
                continue
            Sq = 512 + ke
            q = rnd(Sq, D)
            out = torch.empty_like(q)
            i = [0]

            def run():
                i[0] += 1
                k, vt = slabs[i[0] % 4]
                ops.attention(q, k, vt, out, Skv, H, score_bound=20.0)
            for _ in range(30):
                run()
            res = {}
            for name, env in MODES:
                force(env)
                res[name] = timeit(run, iters=5, warm=2, inner=20)[0] * 1e3
            force({})
            best = min((n for n in res if n != "auto"), key=lambda n: res[n])
            fl = 4.0 * Sq * Skv * D
            print(f"{label} Skv={Skv:<5} Sq={Sq:<5} items={H * ((Sq + 255) // 256):<4} auto {res['auto']:7.1f} us ({fl / res['auto'] / 1e6:6.0f} TF) | "
                  f"best forced: {best:<11} {res[best]:7.1f} us ({(res['auto'] / res[best] - 1) * 100:+5.1f} %) | " +
                  " ".join(f"{n}:{res[n]:.0f}" for n, _ in MODES[1:]), flush=True)


if __name__ == "__main__":
    main()
