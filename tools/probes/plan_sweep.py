#!/usr/bin/env python3
"""Does the GEMM launch planner (gemm.hip: plan256 / estimate128 / gemm_dispatch) pick the fastest schedule for the REGION-step
shapes?  For every shape a region step launches (FLUX / Qwen / Step1X, several K_e), time each schedule the library can be forced
into - cold weights (every launch another copy of W, like the pipeline), the pipeline's epilogue - next to the planner's own choice.

    python tools/probes/plan_sweep.py [--only substr] [--json out.json]

Schedules: auto (planner) | 128 (128 x 128 geometry, 2 blocks / CU) | 256 plain (one launch, partial last round) |
256 + split-K S = 2..8 of the remainder (hand-scheduled pieces + reduce pass) | 256 + quarter-tile remainder.
GPU box only; measurement tool (nothing here is on the product path)."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402

from regione_amd import ops, _lib  # noqa: E402
from bench_kernels import timeit, rnd  # noqa: E402

from bench_kernels import force  # noqa: E402  (rgn_plan_override: the launch-plan knobs, include/regione_hip.h)

SCHEDULES = [("auto", {}), ("128", dict(gemm_geometry=128)), ("256 plain", dict(gemm_geometry=256, gemm_pieces=1))]
SCHEDULES += [(f"256 S={s}", dict(gemm_geometry=256, gemm_pieces=s)) for s in range(2, 9)]
SCHEDULES += [("256 quarter", dict(gemm_geometry=256, gemm_quarter=1))]


def shapes(dense=False, full=False):
    out = []
    T = 512
    if full:              # the FULL-step launches (88 % of the headline edit): FLUX 1024^2, Qwen batched branches, Step1X 512^2 batched
        for fam, ms1, ms2 in (("FLUX F", [8192, T], [8704]), ("S1X512 F", [2048, 2048, T, T], [2560, 2560])):
            out += [(f"{fam} qkv", ms1, 9216, 3072, "bias"), (f"{fam} out", ms1, 3072, 3072, "gate"), (f"{fam} ff1", ms1, 12288, 3072, "gelu"),
                    (f"{fam} ff2", ms1, 3072, 12288, "gate"), (f"{fam} kvq+mlp", ms2, 21504, 3072, "gelu"), (f"{fam} proj_out", ms2, 3072, 15360, "gate")]
        ms = [8192, 8192, 512, 384]
        out += [("Qwen F qkv", ms, 9216, 3072, "bias"), ("Qwen F out", ms, 3072, 3072, "gate"), ("Qwen F ff1", ms, 12288, 3072, "gelu"),
                ("Qwen F ff2", ms, 3072, 12288, "gate")]
        return out
    if dense:             # the long-K / small-N projections across K_e: where the 128 geometry, split-K and the plain launch trade places
        for ke in range(64, 2177, 96):
            out += [(f"FLUX Ke{ke} ff2", [ke, T], 3072, 12288, "gate"), (f"FLUX Ke{ke} proj_out", [T + ke], 3072, 15360, "gate"),
                    (f"FLUX Ke{ke} out", [ke, T], 3072, 3072, "gate")]
        return out
    for pct, ke in ((5, 196), (15, 625), (25, 1024), (50, 2025)):
        out += [(f"FLUX R{pct}% qkv", [ke, T], 9216, 3072, "bias"), (f"FLUX R{pct}% out", [ke, T], 3072, 3072, "gate"),
                (f"FLUX R{pct}% ff1", [ke, T], 12288, 3072, "gelu"), (f"FLUX R{pct}% ff2", [ke, T], 3072, 12288, "gate"),
                (f"FLUX R{pct}% kvq+mlp", [T + ke], 21504, 3072, "gelu"), (f"FLUX R{pct}% proj_out", [T + ke], 3072, 15360, "gate")]
    for pct, ke in ((5, 196), (15, 625), (25, 1024)):
        ms = [ke, ke, 512, 384]                                  # both CFG branches batched: image x2, text 512 / 384
        out += [(f"Qwen R{pct}% qkv", ms, 9216, 3072, "bias"), (f"Qwen R{pct}% out", ms, 3072, 3072, "gate"),
                (f"Qwen R{pct}% ff1", ms, 12288, 3072, "gelu"), (f"Qwen R{pct}% ff2", ms, 3072, 12288, "gate")]
    ms = [256, 256, 512, 512]                                    # Step1X-Edit 512^2, K_e 25 %, B = 2 batched CFG
    out += [("S1X512 R qkv", ms, 9216, 3072, "bias"), ("S1X512 R out", ms, 3072, 3072, "gate"), ("S1X512 R ff1", ms, 12288, 3072, "gelu"),
            ("S1X512 R ff2", ms, 3072, 12288, "gate"), ("S1X512 R kvq+mlp", [768, 768], 21504, 3072, "gelu"),
            ("S1X512 R proj_out", [768, 768], 3072, 15360, "gate")]
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    ap.add_argument("--json", default=None)
    ap.add_argument("--dense", action="store_true", help="K_e sweep of the long-K projections instead of the family table")
    ap.add_argument("--full", action="store_true", help="the full-step launches instead")
    ns = ap.parse_args()
    rows = []
    for name, Ms, N, K, epi in shapes(ns.dense, ns.full):
        if ns.only and ns.only not in name:
            continue
        # problems 0/1 (image rows of the two branches) share one weight matrix, 2/3 (text rows) the other - like the engine
        nW = 1 if len(Ms) == 1 else 2
        ncopies = max(2, int(600e6 // (N * K * 2 * nW)))
        Wsets = [[rnd(N, K) * 0.05 for _ in range(nW)] for _ in range(ncopies)]
        As = [rnd(m, K) for m in Ms]
        b, gate = rnd(N), rnd(N)
        xs = [rnd(m, N) for m in Ms]
        st = {"i": 0}

        def widx(i):
            return 0 if (len(Ms) == 1 or (len(Ms) == 2 and i == 0) or (len(Ms) == 4 and i < 2)) else 1

        def run():
            st["i"] = (st["i"] + 1) % ncopies
            Ws = Wsets[st["i"]]
            if epi == "gate":
                probs = [ops.Problem(As[i], Ws[widx(i)], b, xs[i], gate=gate, resid=xs[i]) for i in range(len(Ms))]
                ops.gemm_group(probs, epilogue=ops.EPI_GATE_RESID)
            else:
                probs = [ops.Problem(As[i], Ws[widx(i)], b, xs[i]) for i in range(len(Ms))]
                ops.gemm_group(probs, epilogue=ops.EPI_GELU if epi == "gelu" else ops.EPI_BIAS, gelu_from_col=(N // 2 if epi == "gelu" else 0))
        res = {}
        for _ in range(40):                 # clocks / allocator settled before the first schedule (auto) is timed
            run()
        for label, env in SCHEDULES:
            force(env)
            med, best = timeit(run, iters=5, warm=2, inner=20)
            plan = _lib.lib().rgn_gemm_last_plan()
            res[label] = dict(us=med * 1e3, plan=plan)
        force({})
        fl = 2.0 * sum(Ms) * N * K
        auto = res["auto"]
        best_label = min((l for l in res if l != "auto"), key=lambda l: res[l]["us"])
        p = auto["plan"]
        pick = ("256" if p & (1 << 10) else "128") + (f" S={p & 255}" if (p & 255) > 1 else "") + (" quarter" if p & (1 << 8) else "")
        tiles256 = sum((m + 255) // 256 for m in Ms) * ((N + 255) // 256)
        line = (f"{name:<22} M={'+'.join(map(str, Ms)):<20} N={N:<6} K={K:<6} tiles256={tiles256:<4} auto[{pick:<12}] {auto['us']:7.1f} us "
                f"({fl / auto['us'] / 1e6:6.0f} TF) | best forced: {best_label:<12} {res[best_label]['us']:7.1f} us "
                f"({(auto['us'] / res[best_label]['us'] - 1) * 100:+5.1f} %) | " + " ".join(f"{l}:{res[l]['us']:.0f}" for l, _ in SCHEDULES[1:]))
        print(line, flush=True)
        rows.append(dict(name=name, M=Ms, N=N, K=K, epilogue=epi, tiles256=tiles256, auto_pick=pick, flops=fl,
                         us={l: round(res[l]["us"], 2) for l in res}))
        del Wsets, As, xs
        torch.cuda.empty_cache()
    if ns.json:
        json.dump(rows, open(ns.json, "w"), indent=1)


if __name__ == "__main__":
    main()
