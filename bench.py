#!/usr/bin/env python3
"""bench.py - denoising steps/s of the RegionE hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" of this contract = one complete 28-step RegionE edit (one pass of the hot path over one
image): the F/R/C kind of a denoising step depends on its index, so the edit is the smallest unit
whose cost is well defined.  `value` = denoising steps per second = 28 * K * N / T (whole job, all
ranks, max-over-ranks time), `ms_per_step` = wall-clock per edit (the "end-to-end edit wall-clock"
half of the metric; encoders / VAE are outside the hot path).

Workload (BASELINE.json configs[1]): FLUX.1-Kontext-dev dims (19 double + 38 single blocks,
d = 3072, 24 x 128 heads, d_ff = 12288; 11.9 B parameters), 1024 x 1024 -> L = L_c = 4096 tokens,
T = 512 text tokens, 28 steps, warmup 6 / post 2 / refresh "16", threshold 0.88, bf16.  Weights
N(0, 0.02^2) and latents N(0,1) are synthetic (no network for checkpoints); the edited region is
forced to 25 % of the tokens by construction by substituting, at step
warmup-1 only, a velocity whose one-step estimate is `condition + 0.1 N(0,1)` outside a rectangle -
the transformer still runs on that step, nothing is skipped.

Multi-GPU: one process per GPU, one image per rank (the path shards per image; SURVEY.md section
8e), weights replicated, no collective on the data path; the final latents are all-gathered to
every rank over RCCL inside the timed region (512 KB per image).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0          # MI355X dense bf16 MFMA, /opt/skills/guides/MI355X_MICROARCH.md
N_STEPS = 28
# counter summaries of THIS command under rocprofv3 --pmc (tools/probes/measure_counters.sh); quoted only when their csrc_sha16 is the
# hash of the kernel sources this process runs (load_stamped)
PMC_TRAFFIC_FILE = "profiles/r06_pmc_traffic.json"
PMC_MFMA_FILE = "profiles/r06_pmc_mfma.json"
PARITY_FILES = ("profiles/r06_parity_headline.json", "profiles/r06_parity_full_depth.json")


def algorithmic_flops(cfg, T, N, K_e):
    """G = 8 d^2 + 4 d d_ff per token per block."""
    d, ff, nl = cfg.d, cfg.d * cfg.mlp_ratio, cfg.n_layers
    G = 8 * d * d + 4 * d * ff
    S = T + N
    f_full = nl * (S * G + 4 * S * S * d)
    f_reg = nl * (T + K_e) * (G + 4 * S * d)
    return f_full, f_reg


class KernelTimer:
    """HIP-event timing of individual launches on the launch stream (torch's current stream)."""

    def __init__(self):
        self.rec = {}
        self.shapes = {}

    def shape_table(self, top=24):
        rows = []
        for (m0, m1, n, k), lst in self.shapes.items():
            ms = sum(s.elapsed_time(e) for s, e in lst)
            rows.append(dict(M=[m0, m1], N=n, K=k, launches=len(lst), total_ms=round(ms, 2),
                             tflops=round(2.0 * (m0 + m1) * n * k * len(lst) / (ms * 1e-3) / 1e12, 1)))
        rows.sort(key=lambda r: -r["total_ms"])
        return rows[:top]

    def wrap(self, ops_mod):
        import regione_amd.ops as ops
        self._orig_gemm, self._orig_attn, self._orig_pair = ops.gemm, ops.attention, ops.gemm_pair
        self._orig_qkv, self._orig_qkv_pair = ops.gemm_qkv, ops.gemm_qkv_pair
        self._orig_group = ops.gemm_group
        timer = self

        def gemm_group(problems, **kw):              # batched CFG branches: up to four problems, one launch
            ps = [p for p in problems if p.A.shape[0] > 0]
            if not ps:
                return None
            N, K = ps[0].W.shape
            ms = [p.A.shape[0] for p in ps]
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            r = timer._orig_group(problems, **kw)
            e.record()
            nw = len({p.W.data_ptr() for p in ps})
            timer.rec.setdefault("gemm_bf16_kernel", []).append((s, e, 2.0 * sum(ms) * N * K, gemm_bytes(sum(ms), N, K, nw, ps[0].W)))
            timer.shapes.setdefault((sum(ms[0::2]), sum(ms[1::2]), N, K), []).append((s, e))
            return r

        def gemm_bytes(m, N, K, n_weights, W):
            """Algorithmic operand bytes of one op launch: A once + every distinct W once + C once (bf16; fp8 W one byte)."""
            return 2.0 * m * K + float(n_weights) * N * K * W.element_size() + 2.0 * m * N

        def timed_gemm(fn, m0, m1, N, K, *a, W=None, **kw):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            r = fn(*a, **kw)
            e.record()
            timer.rec.setdefault("gemm_bf16_kernel", []).append((s, e, 2.0 * (m0 + m1) * N * K, gemm_bytes(m0 + m1, N, K, 2 if m1 else 1, W)))
            timer.shapes.setdefault((m0, m1, N, K), []).append((s, e))
            return r

        def gemm_qkv(A, W, bias, out, epi, **kw):          # fused Q/K/V epilogue: same FLOPs, epilogue work included
            return timed_gemm(timer._orig_qkv, A.shape[0], 0, W.shape[0], A.shape[1], A, W, bias, out, epi, W=W, **kw)

        def gemm_qkv_pair(A0, W0, b0, o0, e0, A1, W1, b1, o1, e1):
            return timed_gemm(timer._orig_qkv_pair, A0.shape[0], A1.shape[0], W0.shape[0], W0.shape[1],
                              A0, W0, b0, o0, e0, A1, W1, b1, o1, e1, W=W0)

        def gemm_pair(A0, W0, b0, o0, A1, W1, b1, o1, **kw):
            N, K = W0.shape
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            r = timer._orig_pair(A0, W0, b0, o0, A1, W1, b1, o1, **kw)
            e.record()
            timer.rec.setdefault("gemm_bf16_kernel", []).append((s, e, 2.0 * (A0.shape[0] + A1.shape[0]) * N * K,
                                                                 gemm_bytes(A0.shape[0] + A1.shape[0], N, K, 2, W0)))
            timer.shapes.setdefault((A0.shape[0], A1.shape[0], N, K), []).append((s, e))
            return r

        def gemm(A, W, bias, out, **kw):
            M, K = A.shape
            N = W.shape[0]
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            r = timer._orig_gemm(A, W, bias, out, **kw)
            e.record()
            timer.rec.setdefault("gemm_bf16_kernel", []).append((s, e, 2.0 * M * N * K, gemm_bytes(M, N, K, 1, W)))
            timer.shapes.setdefault((M, 0, N, K), []).append((s, e))
            return r

        def attention(q, k_slab, vt_slab, out, skv, H, scale=None, workspace=None, score_bound=0.0):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            r = timer._orig_attn(q, k_slab, vt_slab, out, skv, H, scale, workspace, score_bound)
            e.record()
            timer.rec.setdefault("attention_kernel", []).append((s, e, 4.0 * q.shape[0] * skv * H * 128, attn_bytes(q.shape[0], skv, H)))
            if q.shape[0] < skv:             # region step: the K / V^T cache slabs are streamed once per launch
                timer.rec.setdefault("_region_attention_kv", []).append((s, e, 2.0 * skv * H * 128 * 2, 0.0))
            return r

        def attn_bytes(sq, skv, H):
            """Q read + O written + the K and V^T slabs read once (bf16)."""
            return 2.0 * (2.0 * sq + 2.0 * skv) * H * 128

        ops.gemm, ops.attention, ops.gemm_pair = gemm, attention, gemm_pair
        ops.gemm_qkv, ops.gemm_qkv_pair = gemm_qkv, gemm_qkv_pair
        ops.gemm_group = gemm_group
        # the C++ registration (torch.ops.regione_mi.* -> libregione_torch.so -> C ABI) does not pass through regione_amd.ops: the
        # fused Q/K/V projections and the attention launches are timed at the engine's accessor of the op surface instead
        from regione_amd import torch_ops as TO
        self._R = None
        if TO.REGISTRATION == "cpp" and not isinstance(TO.R, type):
            R = self._R = TO.R
            o_kv, o_pair, o_group, o_attn = R.kv_partial_update_, R.kv_partial_update_pair_, R.kv_partial_update_group_, R.region_attention

            def r_kv(x, w, *a, **kw):
                return timed_gemm(o_kv, x.shape[0], 0, w.shape[0], x.shape[1], x, w, *a, W=w, **kw)

            def r_pair(x_img, w_img, b_img, out_img, nq, nk, x_txt, *a, **kw):
                return timed_gemm(o_pair, x_img.shape[0], x_txt.shape[0], w_img.shape[0], w_img.shape[1],
                                  x_img, w_img, b_img, out_img, nq, nk, x_txt, *a, W=w_img, **kw)

            def r_group(x, w_kvq, *a, **kw):
                ms = [t.shape[0] for t in x]
                N, K = w_kvq[0].shape
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                r = o_group(x, w_kvq, *a, **kw)
                e.record()
                timer.rec.setdefault("gemm_bf16_kernel", []).append((s, e, 2.0 * sum(ms) * N * K,
                                                                     gemm_bytes(sum(ms), N, K, len({t.data_ptr() for t in w_kvq}), w_kvq[0])))
                timer.shapes.setdefault((sum(ms[0::2]), sum(ms[1::2]), N, K), []).append((s, e))
                return r

            def r_attn(q, k_cache, vt_cache, out, skv, heads, *a, **kw):
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                r = o_attn(q, k_cache, vt_cache, out, skv, heads, *a, **kw)
                e.record()
                timer.rec.setdefault("attention_kernel", []).append((s, e, 4.0 * q.shape[0] * skv * heads * 128, attn_bytes(q.shape[0], skv, heads)))
                if q.shape[0] < skv:
                    timer.rec.setdefault("_region_attention_kv", []).append((s, e, 2.0 * skv * heads * 128 * 2, 0.0))
                return r
            R.kv_partial_update_, R.kv_partial_update_pair_, R.kv_partial_update_group_, R.region_attention = r_kv, r_pair, r_group, r_attn
        # per-launch durations need launches that do not share the chip: the batched CFG pass keeps both branches' attention on
        # one stream while the timer is installed (the throughput legs run without the timer and with the default)
        from regione_amd.harness import flux as HF
        self._prev_streams, HF.ATTN_BRANCH_STREAMS = HF.ATTN_BRANCH_STREAMS, False

    def unwrap(self):
        import regione_amd.ops as ops
        from regione_amd.harness import flux as HF
        HF.ATTN_BRANCH_STREAMS = self._prev_streams
        ops.gemm, ops.attention, ops.gemm_pair = self._orig_gemm, self._orig_attn, self._orig_pair
        ops.gemm_qkv, ops.gemm_qkv_pair = self._orig_qkv, self._orig_qkv_pair
        ops.gemm_group = self._orig_group
        if self._R is not None:          # instance attributes off: the class's own accessors show through again
            for n in ("kv_partial_update_", "kv_partial_update_pair_", "kv_partial_update_group_", "region_attention"):
                self._R.__dict__.pop(n, None)
            self._R = None

    def summary(self):
        out = {}
        for name, lst in self.rec.items():
            if name.startswith("_"):
                continue
            ms = sum(s.elapsed_time(e) for s, e, _, _ in lst)
            fl = sum(f for _, _, f, _ in lst)
            out[name] = dict(launches=len(lst), total_ms=ms, avg_us=1e3 * ms / len(lst), flops_per_launch=fl / len(lst),
                             achieved_tflops=fl / (ms * 1e-3) / 1e12, algorithmic_bytes_per_launch=sum(b for _, _, _, b in lst) / len(lst))
        return out

    def region_kv_read(self):
        """"KV-read HBM GB/s in region attention" - algorithmic K + V^T slab bytes of the region-step attention
        launches / their time (the launches are MFMA-bound at K_e = 25 %: this is the rate the cache is consumed at, not a
        bandwidth ceiling)."""
        lst = self.rec.get("_region_attention_kv", [])
        if not lst:
            return None
        ms = sum(s.elapsed_time(e) for s, e, _, _ in lst)
        return dict(launches=len(lst), bytes_per_launch=lst[0][2], avg_launch_us=1e3 * ms / len(lst),
                    achieved_gbs=sum(b for _, _, b, _ in lst) / (ms * 1e-3) / 1e9, peak_gbs=8000.0)


def physical_cores():
    """(physical cores, logical CPUs) of the host from /proc/cpuinfo."""
    logical, pairs, phys = os.cpu_count() or 1, set(), None
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                phys = line.split(":")[1].strip()
            elif line.startswith("core id"):
                pairs.add((phys, line.split(":")[1].strip()))
    except OSError:
        pass
    return (len(pairs) or logical), logical


def csrc_hash():
    """sha256 of the kernel sources: stamps the PMC files so that bench.py only quotes traffic measured on THIS build (the same hash
    build_lib() writes beside the library and _lib.lib() checks at load)."""
    from regione_amd.build import csrc_hash as h
    return h()


def _built_from():
    from regione_amd.build import built_from
    return built_from()


def load_stamped(rel_path, stamp=None):
    """A committed counter summary, or {} when it is missing or was measured on other kernel sources than the ones running."""
    try:
        d = json.load(open(os.path.join(ROOT, rel_path)))
    except (OSError, ValueError):
        return {}
    return d if d.get("csrc_sha16") == (stamp or csrc_hash()) else {}


def build_pipeline(cfg, device, seed):
    from regione_amd import synth
    from regione_amd.harness import flux as H
    g = torch.Generator(device=device)
    g.manual_seed(seed)

    def gen():
        for name, shape in synth.flux_param_shapes(cfg).items():
            if name.endswith(".bias"):
                t = torch.randn(shape, generator=g, device=device, dtype=torch.float32) * 0.01
            elif len(shape) == 1:
                t = 1.0 + 0.1 * torch.randn(shape, generator=g, device=device, dtype=torch.float32)
            else:
                t = torch.randn(shape, generator=g, device=device, dtype=torch.float32) * 0.02
            yield name, t.to(torch.bfloat16)
    tr = H.FluxTransformer2DModel(cfg, device).load_state_dict_stream(gen())
    return H.FluxKontextPipeline(tr)


def install_region_injection(pipe, h_tok, w_tok, box, image_latents, seed):
    """At step warmup-1 replace the model output by a velocity whose one-step estimate is the target
    (condition + 0.1 noise outside `box`, fresh noise inside): fixes K_e by construction."""
    from regione_amd import synth
    tgt = synth.region_target(h_tok, w_tok, box, image_latents.cpu(), seed=seed, ramp=0.0)
    g = torch.Generator().manual_seed(seed + 1)
    tgt = tgt + 0.1 * torch.randn(tgt.shape, generator=g)
    r0, r1, c0, c1 = box
    pipe._bench_target = tgt.to(image_latents.device)      # a later call only swaps the target (the K_e = 5 % leg)
    sch = pipe.scheduler
    if getattr(sch, "_bench_injection", False):
        return
    orig = sch.step
    M = pipe._regione_manager

    def step(model_output, timestep, sample, **kw):
        if M.current_step == M.warmup_step - 1:
            i = sch._step_index if sch._step_index is not None else M.current_step
            dt_final = float(sch.sigmas[-1] - sch.sigmas[i])
            model_output = ((pipe._bench_target[None] - sample.float()) / dt_final).to(model_output.dtype)
        return orig(model_output, timestep, sample, **kw)
    sch.step = step
    sch._bench_injection = True


def cpu_baseline(cfg, T, N, K_e, plan):
    """Oracle ('port') timed on the host cores on a bounded sample: ONE double and ONE single block
    (fp32, torch-CPU eager, all cores) at FULL (T+N rows) and REGION (T+K_e query rows) length,
    extrapolated x layer counts x the F/R/C plan to steps/s."""
    from oracle import regione_oracle as O
    from regione_amd import synth
    phys, logical = physical_cores()
    torch.set_num_threads(phys)
    one = synth.FluxConfig(in_channels=cfg.in_channels, n_double=1, n_single=1, heads=cfg.heads, head_dim=cfg.head_dim,
                           joint_dim=cfg.joint_dim, pooled_dim=cfg.pooled_dim)
    w = synth.make_flux_weights(one, seed=1, dtype=torch.float32)
    d = cfg.d
    g = torch.Generator().manual_seed(0)
    L = N // 2
    h_tok = int(round(L ** 0.5))
    ids = torch.cat([torch.zeros(T, 3), synth.flux_latent_ids(h_tok, L // h_tok)], 0)
    rope_full = O.flux_pos_embed(ids)
    st = O.RegionState()
    st.set_parameters(28, 6, 2, "16", 0.88, 0.04, True)
    st.refresh(None, None, T, h_tok, L // h_tok)
    temb = torch.randn(1, d, generator=g)
    caches = [O.KVCache(), O.KVCache()]
    times, samples = {}, {}
    REPS = 3                                   # median of 3 per block (a single sample moved 9 % box to box)

    def med(name, fn):
        ts = []
        for _ in range(REPS):
            t0 = time.perf_counter()
            r = fn()
            ts.append(time.perf_counter() - t0)
        samples[name] = [round(t, 4) for t in ts]
        times[name] = sorted(ts)[REPS // 2]
        return r
    with torch.no_grad():
        # FULL + store
        st.current_step = st.warmup_step - 1
        h, c = torch.randn(1, N, d, generator=g), torch.randn(1, T, d, generator=g)
        c2, h2 = med("double_full", lambda: O.double_block(w, "transformer_blocks.0", cfg.heads, st, caches[0], h, c, temb,
                                                           rope_full, rope_full))
        med("single_full", lambda: O.single_block(w, "single_transformer_blocks.0", cfg.heads, st, caches[1], h2, c2, temb,
                                                  rope_full, rope_full))
        # REGION (update phase)
        st.current_step = st.warmup_step
        st.edited_ids = torch.arange(K_e).unsqueeze(0)
        sel = torch.cat([torch.arange(T), T + st.edited_ids[0]])
        rope_q = (rope_full[0][sel], rope_full[1][sel])
        hr, cr = torch.randn(1, K_e, d, generator=g), torch.randn(1, T, d, generator=g)
        med("single_region", lambda: O.single_block(w, "single_transformer_blocks.0", cfg.heads, st, caches[1], hr, cr, temb,
                                                    rope_q, rope_full))
        # the double block at REGION length is not timed (keeps the sample near 30 s): scaled from the single
        # block by the FULL-length ratio of the two block types
        times["double_region"] = times["single_region"] * times["double_full"] / times["single_full"]
    n_full = sum(1 for p in plan if p in "FS")
    n_reg = plan.count("R")
    edit_s = n_full * (cfg.n_double * times["double_full"] + cfg.n_single * times["single_full"]) + \
        n_reg * (cfg.n_double * times["double_region"] + cfg.n_single * times["single_region"])
    cpu_model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                cpu_model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return dict(value=N_STEPS / edit_s, unit="steps/s", cores=phys, logical_cpus=logical, kind="port",
                sample=(f"oracle (torch-CPU eager fp32, {phys} threads = physical cores, {cpu_model}): 1 double + 1 single block at "
                        f"FULL ({T}+{N} rows), 1 single block at REGION ({T}+{K_e} query rows), each timed {REPS}x (median used; double-block "
                        f"REGION time scaled by the FULL ratio) = {sum(sum(v) for v in samples.values()):.1f} s of CPU "
                        f"work, extrapolated x{cfg.n_double}/{cfg.n_single} layers x plan {n_full}F/{n_reg}R/"
                        f"{plan.count('C')}C -> {edit_s:.0f} s per edit"),
                block_seconds=times, block_samples_s=samples)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3, help="timed edits (28 denoising steps each)")
    ap.add_argument("--warmup", type=int, default=1, help="untimed warm-up edits")
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--edit-frac", type=float, default=0.25)
    ap.add_argument("--edit-fracs", default="0.05,0.15,0.25,0.50",
                    help="N > 1: rank r edits fraction fracs[r %% len] of its image (SURVEY.md section 8e: the only scaling loss the "
                         "path has is K_e imbalance between images); N = 1 uses --edit-frac")
    ap.add_argument("--uniform-edit-frac", action="store_true", help="N > 1: every rank uses --edit-frac")
    ap.add_argument("--true-cfg", type=float, default=0.0,
                    help="true CFG scale (> 1: a cond and an uncond forward per computed step, one K/V cache per branch): "
                         "`--gpus 8 --true-cfg 6.0` is BASELINE.json configs[3]")
    ap.add_argument("--toy", action="store_true", help="toy model (debugging only; result is not a valid bench line)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-5pct", action="store_true", help="profiling runs: skip the untimed K_e = 5 %% leg")
    ap.add_argument("--no-ktimer", action="store_true", help="profiling runs: no per-launch HIP events (drops the roofline objects)")
    ap.add_argument("--no-vanilla", action="store_true")
    ap.add_argument("--dist-backend", default="nccl", help="debugging: 'gloo' lets several ranks share ONE GPU (with --share-gpu)")
    ap.add_argument("--share-gpu", action="store_true", help="debugging: every rank uses cuda:0 (single-GPU box smoke of the multi-rank path)")
    ap.add_argument("--force-collectives", action="store_true",
                    help="initialise the process group and run barrier / all_gather / MAX-reduce even with ONE rank (launch under "
                         "torch.distributed.run --nproc-per-node 1): RCCL executes on a one-GPU box")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1 and args.gpus > 1 and "RANK" not in os.environ:
        # plain `python bench.py --gpus N`: become the launcher (one process per GPU, rendezvous on 127.0.0.1 - the container hostname
        # may not resolve); under torch.distributed.run this branch is never taken.  One JSON line either way (rank 0 prints it).
        import socket
        with socket.socket() as s_:
            s_.bind(("127.0.0.1", 0))
            port = s_.getsockname()[1]
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // args.gpus)))
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.stdout.flush()
        os.execv(sys.executable, cmd)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch one process per GPU (torch.distributed.run --nproc-per-node "
                         f"{args.gpus}) or run plain `python bench.py --gpus {args.gpus}`, which launches them itself")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    from regione_amd import dist as D
    force = args.force_collectives
    dist = D.init(args.dist_backend, device) if (world > 1 or force) else None

    from regione_amd import RegionEHelper, synth, ops
    from oracle import regione_oracle as O            # checker / cpu_baseline leg only

    cfg = synth.FluxConfig(**synth.TOY) if args.toy else synth.FluxConfig()
    T = 32 if args.toy else 512
    h_tok = w_tok = args.size // 16
    L = h_tok * w_tok
    N = 2 * L
    edit_frac = args.edit_frac
    if world > 1 and not args.uniform_edit_frac:
        fr = [float(x) for x in args.edit_fracs.split(",")]
        edit_frac = fr[rank % len(fr)]
    side = int(round((edit_frac * L) ** 0.5))
    box_side = max(side - 2, 3)                      # erosion -1 ring, dilation +2 rings -> side x side
    r0 = (h_tok - box_side) // 2
    box = (r0, r0 + box_side, r0, r0 + box_side)

    t_build = time.perf_counter()
    pipe = build_pipeline(cfg, device, seed=42)
    torch.cuda.synchronize()
    t_build = time.perf_counter() - t_build
    lat, img, prompt, pooled = synth.make_edit_inputs(h_tok, w_tok, T, cfg, seed=110 + rank, dtype=torch.bfloat16)
    lat, img, prompt, pooled = lat.to(device), img.to(device), prompt.to(device), pooled.to(device)
    cfg_kw = {}
    if args.true_cfg > 1:
        _, _, nprompt, npooled = synth.make_edit_inputs(h_tok, w_tok, T, cfg, seed=910 + rank, dtype=torch.bfloat16)
        cfg_kw = dict(true_cfg_scale=args.true_cfg, negative_prompt_embeds=nprompt.to(device),
                      negative_pooled_prompt_embeds=npooled.to(device))

    helper = RegionEHelper(pipe)
    import contextlib
    with contextlib.redirect_stdout(sys.stderr):        # stdout carries exactly one JSON line
        helper.set_params(threshold=0.88, cache_threshold=0.04, warmup_step=6, post_step=2, refresh_step="16")
    helper.enable()
    install_region_injection(pipe, h_tok, w_tok, box, img[0:1], seed=7)

    def edit(trace=None, **kw):
        return pipe(image=img, prompt_embeds=prompt, pooled_prompt_embeds=pooled, height=args.size, width=args.size,
                    latents=lat, guidance_scale=2.5, return_dict=False, trace=trace, **cfg_kw, **kw)[0]

    def step_times(trace):
        """GPU time of every denoising step of one (untimed) edit: an event at each `callback_on_step_end`."""
        evs = [torch.cuda.Event(enable_timing=True)]
        evs[0].record()

        def cb(p, i, t, kw):
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            evs.append(e)
            return {}
        o = edit(trace, callback_on_step_end=cb)
        torch.cuda.synchronize()
        ms = [a.elapsed_time(b) for a, b in zip(evs[:-1], evs[1:])]
        # a second instrumented edit, per-step minimum of the two: one untimed edit is exposed to one-off host stalls between two events
        # (the driver's round-5 line showed ONE region step at 44.7 ms among 29.2 ms ones; four builder runs of the same code had none)
        evs[:] = [torch.cuda.Event(enable_timing=True)]
        evs[0].record()
        edit(None, callback_on_step_end=cb)
        torch.cuda.synchronize()
        ms = [min(m, a.elapsed_time(b)) for m, a, b in zip(ms, evs[:-1], evs[1:])]
        by = {}
        for k, m in zip(trace["kind"], ms):
            by.setdefault(k, []).append(m)
        return o, {k: dict(n=len(v), avg_ms=sum(v) / len(v), min_ms=min(v), max_ms=max(v)) for k, v in by.items()}

    rank_edit_s = []

    for _ in range(args.warmup):
        out = edit()
        D.gather_latents([out], [rank], world, dist, force=force)   # warm the collective too (communicator / channel setup)
    timer = KernelTimer()

    def job():
        for k in range(args.steps):
            # per-launch HIP events cost ~1.3 % of an edit (two marker packets around each of ~900 launches), so only
            # the LAST timed edit carries them: the roofline figures come from inside the timed region, the headline
            # number is not paying for its own instrumentation on the other edits
            instrument = (k == args.steps - 1) and not args.no_ktimer
            if instrument:
                timer.wrap(ops)
            if world > 1:
                torch.cuda.synchronize()
                t_e = time.perf_counter()
            o = edit()
            if world > 1:                    # this rank's own edit time (the collective below waits for the slowest rank)
                torch.cuda.synchronize()
                rank_edit_s.append(time.perf_counter() - t_e)
            if instrument:
                timer.unwrap()
            # every rank ends with every image's final latents (512 KB each): the only data collective
            got = D.gather_latents([o], [rank], world, dist, force=force)
            if force:
                assert len(got) == world and all(g is not None and g.shape == o.shape for g in got)
    elapsed = D.timed(job, torch.cuda.synchronize, dist, force=force)       # barrier + sync both sides, MAX over ranks

    # characterise the run (untimed): step kinds, K_e
    trace = {}
    out, step_ms = step_times(trace)
    kinds = "".join(trace["kind"])
    K_e = int(pipe._regione_manager.edited_ids.shape[1])
    f_full, f_reg = algorithmic_flops(cfg, T, N, K_e)
    n_full, n_reg, n_cache = kinds.count("F"), kinds.count("R"), kinds.count("C")
    flops_edit = (n_full * f_full + n_reg * f_reg) * (2 if args.true_cfg > 1 else 1)
    per_rank = None
    if world > 1:
        per_rank = [None] * world
        dist.all_gather_object(per_rank, dict(rank=rank, K_e=K_e, edit_frac=edit_frac,
                                              edit_s=sum(rank_edit_s) / max(len(rank_edit_s), 1)))
    edit_s = elapsed / args.steps
    ksum = timer.summary()

    result = {
        "metric": "denoising steps/sec (28-step RegionE edit, 1024x1024)", "value": N_STEPS * args.steps * world / elapsed,
        "unit": "steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * edit_s,
        "step_definition": "one bench step = one complete 28-step edit of one image per GPU (ms_per_step = ms per EDIT; "
                           "ms_per_denoising_step = ms_per_step / 28)",
        "ms_per_denoising_step": 1e3 * edit_s / N_STEPS,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "config": {"workload": f"FLUX.1-Kontext-dev {args.size}x{args.size} 28-step RegionE edit, warmup=6 post=2 refresh=16 "
                               f"thresh=0.88 cache_thresh=0.04, L=L_c={L}, T={T}, " +
                               (f"K_e={K_e} ({100.0 * K_e / L:.1f}% edited by construction)" if per_rank is None else
                                "K_e per rank: " + "/".join(str(r["K_e"]) for r in per_rank) + " (see per_rank)") +
                               f", plan {kinds}" + (f", true CFG {args.true_cfg} (two forwards per computed step)"
                                                                  if args.true_cfg > 1 else ""),
                   "images_per_gpu": 1, "parallelism": f"image-sharded x{world}",
                   "params_billion": round(sum(int(torch.tensor(s).prod()) for s in synth.flux_param_shapes(cfg).values()) / 1e9, 2)},
        "edit_wall_clock_s": edit_s,
        "native_library": {"path": os.path.relpath(ops._lib.LIB_PATH, ROOT), "abi_version": int(ops._lib.lib().rgn_version()),
                           "kernel_sources_sha16": csrc_hash(), "built_from_sha16": _built_from()},
        "algorithmic_tflop_per_edit": flops_edit / 1e12,
        "loop_mfma_frac": flops_edit / edit_s / 1e12 / PEAK_BF16_TFLOPS,
        "model_build_s": t_build,
        # F = full-token step (incl. the partition at step warmup-1), R = region step (T + K_e query rows), C = cache-served
        "step_ms_by_kind": step_ms,
    }
    if per_rank is not None:
        result["per_rank"] = per_rank          # K_e imbalance between images = the path's only scaling loss
    if dist is not None:
        result["collectives"] = {"backend": dist.get_backend(), "world": dist.get_world_size(), "forced_in_world_of_one": bool(force)}
    # PMC passes cannot run inside the timed region (counter collection serialises kernels): fabric traffic and MFMA-busy come from
    # separate rocprofv3 --pmc runs of THIS command (tools/pmc_traffic.py / pmc_summary.py), quoted only when those files were measured
    # on the same kernel sources (csrc_sha16); otherwise null.  The counters tally kernel DISPATCHES (an op launch = whole rounds +
    # remainder pieces + a reduce / merge pass); `traffic` here is per OP LAUNCH = the family's bytes per edit / the op launches per
    # edit, so that it sits beside `flops_per_launch` and `algorithmic_bytes_per_launch` in the same unit (verdict r4, weak #6).
    pmc, busy = load_stamped(PMC_TRAFFIC_FILE), load_stamped(PMC_MFMA_FILE)

    def traffic_fields(family, k):
        t = pmc.get(family, {})
        per_edit = t.get("traffic_bytes_per_edit")
        alg = k["algorithmic_bytes_per_launch"]
        per_launch = per_edit / k["launches"] if per_edit else None
        return {"traffic": per_launch, "algorithmic_bytes_per_launch": alg,
                "traffic_vs_algorithmic": (per_launch / alg) if per_launch else None,
                "traffic_unit": (f"bytes per OP LAUNCH = L2<-fabric reads (x2-corrected, Infinity-Cache hits included) + WRITE_SIZE of every "
                                 f"dispatch of the family in one edit / op launches per edit ({PMC_TRAFFIC_FILE}, same csrc_sha16; "
                                 f"{t.get('dispatches_per_edit')} dispatches per edit for {k['launches']} op launches: dispatches != launches)"),
                "hbm_side": t.get("hbm_side")}
    if "gemm_bf16_kernel" in ksum:
        k = ksum["gemm_bf16_kernel"]
        result["roofline"] = {"bound": "mfma", "kernel": "gemm_bf16_kernel", "achieved": k["achieved_tflops"],
                              "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": k["achieved_tflops"] / PEAK_BF16_TFLOPS,
                              **traffic_fields("gemm_bf16_kernel", k),
                              "mfma_busy": busy.get("ALL gemm_bf16_kernel", {}).get("mfma_util"),
                              "mfma_busy_unit": f"SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 1024 SIMDs), {PMC_MFMA_FILE}, same csrc_sha16",
                              "launches": k["launches"], "avg_launch_us": k["avg_us"],
                              "flops_per_launch": k["flops_per_launch"], "share_of_edit_time": k["total_ms"] * 1e-3 / edit_s}
    result["gemm_shapes"] = timer.shape_table()
    if "attention_kernel" in ksum:
        k = ksum["attention_kernel"]
        result["roofline_attention"] = {"bound": "mfma", "kernel": "attention_asm_kernel (+ attention_kernel for ragged KV lengths)", "achieved": k["achieved_tflops"],
                                        "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": k["achieved_tflops"] / PEAK_BF16_TFLOPS,
                                        **traffic_fields("attention_kernel", k),
                                        "mfma_busy": busy.get("ALL attention_kernel", {}).get("mfma_util"),
                                        "launches": k["launches"], "avg_launch_us": k["avg_us"], "flops_per_launch": k["flops_per_launch"],
                                        "share_of_edit_time": k["total_ms"] * 1e-3 / edit_s}

    kv = timer.region_kv_read()
    if kv is not None:
        result["region_attention_kv_read"] = kv
    if rank == 0 and world == 1 and not args.toy and not args.no_5pct:
        # untimed extra leg: the same edit with K_e = 5 % of the tokens - the case where the region-step attention launch is
        # bound by the K / V^T cache read rather than by MFMA ("KV-read HBM GB/s in region attention")
        side5 = int(round((0.05 * L) ** 0.5))
        b5 = max(side5 - 2, 3)
        r5 = (h_tok - b5) // 2
        keep = pipe._bench_target
        install_region_injection(pipe, h_tok, w_tok, (r5, r5 + b5, r5, r5 + b5), img[0:1], seed=7)
        edit()                                                   # warm (new K_e: new split plans / tables)
        t5 = KernelTimer()
        t5.wrap(ops)
        tr5 = {}
        _, sm5 = step_times(tr5)
        t5.unwrap()
        s5 = t5.summary()
        result["region_5pct"] = {"K_e": int(pipe._regione_manager.edited_ids.shape[1]), "step_ms_by_kind": sm5,
                                 "region_attention_kv_read": t5.region_kv_read(),
                                 "gemm_tflops": s5.get("gemm_bf16_kernel", {}).get("achieved_tflops"),
                                 "attention_tflops": s5.get("attention_kernel", {}).get("achieved_tflops")}
        pipe._bench_target = keep
    if rank == 0 and world == 1 and not args.no_vanilla:
        # full-token denoising on the same engine: the speed-up the reference headlines (README.md:23)
        helper.disable()
        van = pipe(image=img, prompt_embeds=prompt, pooled_prompt_embeds=pooled, height=args.size, width=args.size,
                   latents=lat, guidance_scale=2.5, return_dict=False)[0]       # warm
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        van = pipe(image=img, prompt_embeds=prompt, pooled_prompt_embeds=pooled, height=args.size, width=args.size,
                   latents=lat, guidance_scale=2.5, return_dict=False)[0]
        torch.cuda.synchronize()
        tv = time.perf_counter() - t0
        result["full_token"] = {"edit_wall_clock_s": tv, "steps_per_s": N_STEPS / tv,
                                "mfma_frac": N_STEPS * f_full / tv / 1e12 / PEAK_BF16_TFLOPS}
        result["speedup_vs_full_token"] = tv / edit_s
        # NOT the quality metric of BASELINE.json (PSNR >= 30.5 dB vs the full-token output): on N(0, 0.02^2) weights the trunk is no
        # denoiser (no contraction towards an image), so this number only says the two loops ran on the same inputs; the
        # measurable half of the target - agreement with the reference on identical inputs - is `parity` below
        result["latent_psnr_vs_full_token_random_weights_db"] = {
            "value": O.psnr(out.cpu(), van.cpu()),
            "note": "random weights: not the quality metric (needs a real checkpoint; unmeasurable in this image)"}
    if rank == 0 and world == 1 and not args.toy and not args.no_vanilla:
        # SURVEY.md section 8 row f4 / BASELINE.json "end-to-end edit wall-clock": the reference's timing protocol wraps the whole pipe(...) call
        # (src/FluxKontext/main.py:62-73), whose last stage is `vae.decode` (FluxKontext/inplace.py:396-402).  Untimed leg: the final latents
        # of the edit above through the HIP decoder (regione_amd/vae.py: [EXT] AutoencoderKL layout, synthetic weights - timing does not
        # depend on them), and a 1024 x 1024 condition image through the HIP encoder (the host's prepare_latents, inplace.py:210-226).
        from regione_amd import vae as V
        dec = V.HipVaeDecoder(V.synthetic_decoder_state_dict(3, device=device), device)
        enc = V.HipVaeEncoder(V.synthetic_decoder_state_dict(4, device=device, shapes=V.encoder_param_shapes()), device)
        z = out[0].view(h_tok, w_tok, 16, 2, 2).permute(2, 0, 3, 1, 4).reshape(1, 16, 2 * h_tok, 2 * w_tok)       # _unpack_latents
        cond_image = (torch.rand(1, 3, args.size, args.size, generator=torch.Generator().manual_seed(9)) * 2 - 1).to(device)

        def med_ms(fn):
            ms = []
            for k in range(6):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                r = fn()
                torch.cuda.synchronize()
                ms.append(1e3 * (time.perf_counter() - t0))
            ms = sorted(ms[1:])
            return ms[len(ms) // 2], r
        d_ms, img = med_ms(lambda: dec.decode(z))
        e_ms, mom = med_ms(lambda: enc.encode(cond_image))
        vae_s = (d_ms + e_ms) * 1e-3
        result["end_to_end"] = {
            "vae_decode_ms": d_ms, "vae_decode_tflops": dec.flops(2 * h_tok, 2 * w_tok) / d_ms / 1e9,
            "vae_encode_ms": e_ms, "vae_encode_tflops": enc.flops(args.size, args.size) / e_ms / 1e9,
            "image": list(img.shape), "end_to_end_s": edit_s + vae_s,
            "end_to_end_full_token_s": (result["full_token"]["edit_wall_clock_s"] + vae_s) if "full_token" in result else None,
            "note": "VAE encode of the condition image + loop + VAE decode, all on libregione_hip.so (synthetic AutoencoderKL weights); the "
                    "text encoders (T5-XXL / CLIP-L: ~17 ms eager, tools/f4_host_side.py) are host modules and not included"}
        del enc, mom
        del dec, img
        torch.cuda.empty_cache()
    if rank == 0 and world == 1:
        # parity with the oracle (torch-CPU bf16 = the reference's dtype path), committed tool reports (tools/parity_full_depth.py; the
        # -m gpu suite re-runs the 16 x 16-grid cases with assertions): round 5 = the HEADLINE shape itself (L = 4096, T = 512, K_e = 1024,
        # 57 blocks, d = 3072: one FULL step with store + one REGION step) and the 28-step loops at full width AND depth; round 4 = the
        # d = 512 loops and the 16 x 16-grid full-width steps
        par = {"note": "committed reports of tools/parity_full_depth.py (the -m gpu suite asserts the same cases); NOT measured in this run",
               "sources": [f for f in PARITY_FILES if os.path.exists(os.path.join(ROOT, f))]}
        for f in par["sources"]:
            rep = json.load(open(os.path.join(ROOT, f)))
            label = {"source": f, "measured_in_this_run": False, "csrc_sha16": rep.get("csrc_sha16"),
                     "stale": rep.get("csrc_sha16") != csrc_hash()}          # measured on other kernel sources than the ones running
            for c in rep["cases"]:
                name = c["case"]
                if "rows" in c:
                    par[name] = {"min_psnr_db": min(r["psnr_hip_vs_oracle_db"] for r in c["rows"]), "blocks": c["blocks"], "d": c["d"],
                                 "grid": c.get("grid"), "T": c.get("T"), "K_e": c.get("K_e"),
                                 "untouched_cache_rows_bit_identical": c.get("untouched_rows_bit_identical"),
                                 "oracle_own_spread_min_psnr_db": min((r["psnr_oracle_reordered_vs_oracle_db"] for r in c["rows"]
                                                                       if "psnr_oracle_reordered_vs_oracle_db" in r), default=None)}
                else:
                    par[name] = {"final_latents_psnr_db": c.get("psnr_final_db"), "ids_bit_exact": c.get("ids_bit_exact"),
                                 "plan_equal": c.get("hip_plan") == c.get("oracle_plan"), "blocks": c["blocks"], "d": c["d"],
                                 "oracle_own_spread_psnr_db": c.get("psnr_oracle_reordered_vs_oracle_db")}
                    w = c.get("worst_combined_step")
                    if w:          # true-CFG families: where the combined velocity's distance comes from (per-branch vs combine)
                        par[name]["worst_step_branch_attribution_db"] = {k: w[k] for k in ("step", "psnr_cond_db", "psnr_uncond_db", "psnr_combined_db")}
                par[name].update(label)
        if len(par) > 2:
            result["parity_full_depth_db"] = par
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        plan = "".join(O.derive_schedule(L, "flux", 6, 2, "16", 0.04))
        result["cpu_baseline"] = cpu_baseline(cfg, T, N, K_e, plan)
    if rank == 0:
        print(json.dumps(result), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
