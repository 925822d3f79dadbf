#!/usr/bin/env python3
"""Micro-benchmarks of the dominant kernels at the FLUX 1024^2 shapes (GPU box only), sustained rate.
    python tools/bench_kernels.py [gemm] [attn] [gemv]
Reports TFLOP/s (algorithmic) per shape, median of N interleaved rounds, uniform random [-1,1) data."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from regione_amd import ops


def timeit(fn, iters=7, warm=2, inner=25):
    """Median / best over `iters` rounds of `inner` BACK-TO-BACK launches (no host sync inside a round): the
    engine runs these kernels in a dense stream, and a sync per launch adds ~10 % of clock-ramp/idle time."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(inner):
            fn()
        e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) / inner)
    ts.sort()
    return ts[len(ts) // 2], ts[0]


def rnd(*shape):
    return (torch.rand(*shape, device="cuda") * 2 - 1).to(torch.bfloat16)


def bench_gemm():
    shapes = [("single kvq+mlp", 8704, 21504, 3072), ("single proj_out", 8704, 3072, 15360),
              ("double img qkv", 8192, 9216, 3072), ("double img out", 8192, 3072, 3072),
              ("double img ff1", 8192, 12288, 3072), ("double img ff2", 8192, 3072, 12288),
              ("double txt qkv", 512, 9216, 3072), ("double txt ff2", 512, 3072, 12288),
              ("region kvq+mlp", 1536, 21504, 3072), ("region proj_out", 1536, 3072, 15360),
              ("region img out", 1024, 3072, 3072), ("square 8192", 8192, 8192, 8192)]
    if os.environ.get("GEMM_KSWEEP"):      # 1024 tiles = 4 exact rounds: per-tile time = a*K + b separates loop rate from tile overhead
        shapes = [(f"ksweep {k}", 8192, 8192, k) for k in (512, 1024, 2048, 3072, 4096, 8192)]
    only = os.environ.get("GEMM_ONLY")
    # GEMM_COLD=1: every launch reads a DIFFERENT copy of W (enough copies to overflow the 256 MiB Infinity Cache), like
    # the pipeline, where each layer's weights arrive from HBM; without it 25 back-to-back launches re-read W from the MALL
    cold = os.environ.get("GEMM_COLD")
    for name, M, N, K in shapes:
        if only and only not in name:
            continue
        A, W, b = rnd(M, K), rnd(N, K) * 0.05, rnd(N)
        out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
        Ws = [W]
        if cold:
            Ws += [W.clone() for _ in range(max(1, int(600e6 // (N * K * 2))))]
        state = {"i": 0}

        def run():
            state["i"] = (state["i"] + 1) % len(Ws)
            ops.gemm(A, Ws[state["i"]], b, out)
        for variant in VARIANTS:
            force(GEOM[variant])
            med, best = timeit(run)
            fl = 2.0 * M * N * K
            print(f"gemm[{variant:>4}] {name:<18} M={M:<5} N={N:<6} K={K:<6} {med*1e3:8.1f} us  {fl/med/1e9:7.1f} TF (best {fl/best/1e9:7.1f})")
        if os.environ.get("GEMM_VENDOR"):      # reference point only: the vendor library (hipBLASLt via torch), bias epilogue
            med, best = timeit(lambda: torch.addmm(b, A, W.t(), out=out))
            print(f"gemm[ lib] {name:<18} M={M:<5} N={N:<6} K={K:<6} {med*1e3:8.1f} us  {fl/med/1e9:7.1f} TF (best {fl/best/1e9:7.1f})")
        del A, W, out


def bench_small():
    """Region-step shapes (M = T + K_e): single GEMMs and the text+image pairs of the double blocks."""
    singles = [("R proj_out", 1536, 3072, 15360), ("R kvq+mlp", 1536, 21504, 3072), ("R5% proj_out", 708, 3072, 15360),
               ("R5% kvq+mlp", 708, 21504, 3072), ("R50% proj_out", 2537, 3072, 15360)]
    pairs = [("R qkv", 1024, 512, 9216, 3072), ("R out", 1024, 512, 3072, 3072), ("R ff1", 1024, 512, 12288, 3072),
             ("R ff2", 1024, 512, 3072, 12288), ("R5% ff2", 196, 512, 3072, 12288), ("R5% ff1", 196, 512, 12288, 3072),
             ("Qwen R ff2 T384", 1024, 384, 3072, 12288)]
    if os.environ.get("GEMM_BATCHED_BRANCHES"):      # what batching the cond + uncond branches of a Qwen region step would buy
        pairs = [(f"Q {n} cond", 1024, 512, N, K) for n, N, K in (("qkv", 9216, 3072), ("out", 3072, 3072), ("ff1", 12288, 3072), ("ff2", 3072, 12288))]
        pairs += [(f"Q {n} uncond", 1024, 384, N, K) for n, N, K in (("qkv", 9216, 3072), ("out", 3072, 3072), ("ff1", 12288, 3072), ("ff2", 3072, 12288))]
        pairs += [(f"Q {n} both", 2048, 896, N, K) for n, N, K in (("qkv", 9216, 3072), ("out", 3072, 3072), ("ff1", 12288, 3072), ("ff2", 3072, 12288))]
        singles = []
    gated = os.environ.get("GEMM_EPI") == "gate"          # gated-residual epilogue (out-projections, ff2, proj_out) instead of bias
    cold = os.environ.get("GEMM_COLD")                    # rotate over enough weight copies to overflow the Infinity Cache
    only = os.environ.get("GEMM_ONLY")

    def copies(W):
        return [W] + ([W.clone() for _ in range(max(1, int(600e6 // (W.numel() * 2))))] if cold else [])
    st = {"i": 0}

    def nxt(Ws):
        st["i"] = (st["i"] + 1) % len(Ws)
        return Ws[st["i"]]
    for name, M, N, K in singles:
        if only and only not in name:
            continue
        A, W, b = rnd(M, K), rnd(N, K) * 0.05, rnd(N)
        Ws = copies(W)
        out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
        x, gate = rnd(M, N), rnd(N)
        for variant in VARIANTS:
            force(GEOM[variant])
            if gated:
                med, best = timeit(lambda: ops.gemm(A, nxt(Ws), b, x, epilogue=ops.EPI_GATE_RESID, gate=gate, resid=x))
            else:
                med, best = timeit(lambda: ops.gemm(A, nxt(Ws), b, out))
            fl = 2.0 * M * N * K
            print(f"gemm[{variant:>4}] {name:<18} M={M:<5} N={N:<6} K={K:<6} {med*1e3:8.1f} us  {fl/med/1e9:7.1f} TF  (ideal@1150 {fl/1150e6:6.1f} us)")
    for name, M0, M1, N, K in pairs:
        if only and only not in name:
            continue
        A0, A1, W0, W1, b = rnd(M0, K), rnd(M1, K), rnd(N, K) * 0.05, rnd(N, K) * 0.05, rnd(N)
        W0s, W1s = copies(W0), copies(W1)
        o0, o1 = torch.empty(M0, N, dtype=torch.bfloat16, device="cuda"), torch.empty(M1, N, dtype=torch.bfloat16, device="cuda")
        for variant in VARIANTS:
            force(GEOM[variant])
            if gated:
                x0, x1, gate = rnd(M0, N), rnd(M1, N), rnd(N)
                med, best = timeit(lambda: ops.gemm_pair(A0, nxt(W0s), b, x0, A1, W1s[st["i"]], b, x1, epilogue=ops.EPI_GATE_RESID,
                                                         gate0=gate, resid0=x0, gate1=gate, resid1=x1))
            else:
                med, best = timeit(lambda: ops.gemm_pair(A0, nxt(W0s), b, o0, A1, W1s[st["i"]], b, o1))
            fl = 2.0 * (M0 + M1) * N * K
            print(f"pair[{variant:>4}] {name:<18} M={M0}+{M1:<4} N={N:<6} K={K:<6} {med*1e3:8.1f} us  {fl/med/1e9:7.1f} TF  (ideal@1150 {fl/1150e6:6.1f} us)")


def bench_w8():
    """fp8 (e4m3fn) weights: bf16 weights vs the three fp8 paths - tiles inside the hand-scheduled loop (round 3 default),
    compiler-scheduled fp8 tiles (gemm_asm = 0) - cold weights (every launch a different copy)."""
    shapes = [("v1p2 R kvq+mlp x2", 9088, 21504, 3072), ("v1p2 R proj_out x2", 9088, 3072, 15360), ("v1p2 F kvq+mlp", 33280, 21504, 3072),
              ("v1p2 F proj_out", 33280, 3072, 15360), ("R kvq+mlp", 1536, 21504, 3072), ("R proj_out", 1536, 3072, 15360),
              ("R5% kvq+mlp", 708, 21504, 3072), ("R5% proj_out", 708, 3072, 15360)]
    only = os.environ.get("GEMM_ONLY")
    for name, M, N, K in shapes:
        if only and only not in name:
            continue
        A, b = rnd(M, K), rnd(N)
        out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
        ncopy = max(2, int(600e6 // (N * K)))
        W16 = [rnd(N, K) * 0.05 for _ in range(max(2, ncopy // 2))]
        W8 = [ops.quantize_w8(rnd(N, K) * 0.05) for _ in range(ncopy)]
        st = {"i": 0}

        def run(Ws):
            st["i"] = (st["i"] + 1) % len(Ws)
            ops.gemm(A, Ws[st["i"]], b, out)
        fl = 2.0 * M * N * K
        rows = [("bf16 W", W16, {}), ("fp8 asm loop", W8, {}), ("fp8 compiler tiles", W8, dict(gemm_asm=0))]
        for label, Ws, env in rows:
            force(env)
            med, best = timeit(lambda: run(Ws), iters=5, inner=15)
            print(f"w8 {name:<20} M={M:<6} N={N:<6} K={K:<6} {label:<20} {med*1e3:8.1f} us  {fl/med/1e9:7.1f} TF")
        del W16, W8, A, out


def bench_region():
    """The region-partition ops (O(L * 64) bytes -> launch-latency bound; achieved GB/s reported for
    honesty) next to the oracle's CPU time for the same call."""
    import time
    from oracle import regione_oracle as O
    for L, h, w in ((4096, 64, 64), (16384, 128, 128)):
        g = torch.Generator().manual_seed(0)
        sample = torch.randn(1, L, 64, generator=g)
        cond = torch.randn(1, L, 64, generator=g).to(torch.bfloat16)
        v = ((cond.float() - sample) / -0.7 + (torch.rand(1, L, 1, generator=g) > 0.7) * torch.randn(1, L, 64, generator=g)).to(torch.bfloat16)
        sc, cc, vc = sample.cuda(), cond.cuda(), v.cuda()
        e, u, mask, raw, _ = ops.arp_partition(sc, vc, cc, -0.7, 0.88, h, w, True)
        K = e.numel()
        cases = [
            ("arp_partition", lambda: ops.arp_partition(sc, vc, cc, -0.7, 0.88, h, w, True), L * 64 * (4 + 2 + 2) + L + 8 * L,
             lambda: O.token_selector(sample + torch.tensor(-0.7) * v, cond, 0.88, h, w, True)),
            ("gather_rows", lambda: ops.gather_rows(cc, e), 2 * K * 64 * 2, lambda: O.ids_gather(cond, e.cpu())),
            ("scatter_rows_", lambda: ops.scatter_rows_(ops.gather_rows(cc, e), e, cc), 2 * K * 64 * 2, None),
            ("euler_step (split)", lambda: ops.euler_step(sc, vc, -0.03, mask, -0.5), L * 64 * (4 + 2) + L * 64 * 2, None),
            ("avd_apply (gather)", lambda: ops.avd_apply(vc, 1.0173, e), 2 * K * 64 * 2, lambda: O.ids_gather(v, e.cpu()) * torch.tensor(1.0173)),
        ]
        for name, fn, nbytes, cpu in cases:
            med, best = timeit(fn, iters=5, warm=2, inner=50)
            line = f"region L={L:<6} K_e={K:<6} {name:<20} {med*1e3:7.1f} us  {nbytes/med/1e6:7.1f} GB/s"
            if cpu is not None:
                t0 = time.perf_counter()
                for _ in range(5):
                    cpu()
                line += f"   | oracle on CPU {(time.perf_counter()-t0)/5*1e6:9.1f} us ({torch.get_num_threads()} threads)"
            print(line)


def bench_attn():
    shapes = [("full", 8704, 8704, 24), ("region 25%", 1536, 8704, 24), ("region 5%", 717, 8704, 24)]
    if os.environ.get("ATTN_MORE"):       # other families' region steps: Qwen uncond branch, Step1X v1p2 2048^2, K_e 15 / 50 %
        shapes += [("qwen R T384", 1408, 8576, 24), ("v1p2 2048 R", 4608, 33280, 24), ("region 15%", 1137, 8704, 24),
                   ("region 50%", 2537, 8704, 24)]
    for name, Sq, Skv, H in shapes:
        D = H * 128
        q, k, vt = rnd(Sq, D), rnd(Skv, D), rnd(D, Skv)
        out = torch.empty_like(q)
        for variant in AVARIANTS:
            force(AGEOM[variant])
            med, best = timeit(lambda: ops.attention(q, k, vt, out, Skv, H))
            fl = 4.0 * Sq * Skv * D
            print(f"attn[{variant:>4}] {name:<12} Sq={Sq:<5} Skv={Skv:<5} {med*1e3:8.1f} us  {fl/med/1e9:7.1f} TF (best {fl/best/1e9:7.1f})")


def bench_gemv():
    N, K = 1056768, 3072
    W, b, x = rnd(N, K), rnd(N), rnd(1, K)
    med, best = timeit(lambda: ops.gemv(x, W, b, silu_input=True), iters=5)
    print(f"gemv modulation N={N} K={K}: {med*1e3:.1f} us  {N*K*2/med/1e9:.2f} TB/s")


GEOM = {"auto": {}, "128": dict(gemm_geometry=128), "256c": dict(gemm_geometry=256, gemm_asm=0), "256": dict(gemm_geometry=256)}
AGEOM = {"auto": {}, "8": dict(attn_waves=8), "8n": dict(attn_waves=8, attn_split=0), "4": dict(attn_waves=4), "4n": dict(attn_waves=4, attn_split=0)}


def force(knobs):
    """Launch-plan knobs for the following launches (rgn_plan_override): every knob back to -1 first."""
    from regione_amd import _lib
    _lib.lib().rgn_plan_override(None, 0)
    for k, v in knobs.items():
        _lib.check(_lib.lib().rgn_plan_override(k.encode(), int(v)), k)


VARIANTS = os.environ.get("GEMM_VARIANTS", "auto").split(",")          # auto | 128 | 256c (compiler-scheduled) | 256
AVARIANTS = os.environ.get("ATTN_VARIANTS", "auto").split(",")   # e.g. 8,8n,4,4n (n = no KV split)
if __name__ == "__main__":
    which = sys.argv[1:] or ["gemm", "attn", "gemv"]
    if "gemm" in which: bench_gemm()
    if "small" in which: bench_small()
    if "region" in which: bench_region()
    if "w8" in which: bench_w8()
    if "attn" in which: bench_attn()
    if "gemv" in which: bench_gemv()
