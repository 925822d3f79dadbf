#!/usr/bin/env python3
"""Full-size runs of the BASELINE.json configurations that are NOT the bench line (GPU box only).

    python tools/run_configs.py [flux_sweep] [flux_cfg] [step1x_512] [step1x_v1p2_2048] [step1x_v1p2_2048_50] [step1x_v1p2_2048_50_fp8] [qwen_1024] [--out FILE]

Each case builds the family's engine at its public dimensions with synthetic weights, enables RegionE
through RegionEHelper exactly like a user would, fixes the edited region by construction (the same
velocity substitution at step warmup-1 as bench.py) and checks size-independent properties at full size:

  * the edited ids are exactly the ids of the constructed rectangle (erosion -1 ring, dilation +2 rings),
  * the F/R/C plan equals the plan derived from the reference's decision logic for that family / length,
  * latents are finite,

and reports wall-clock, steps/s and the speed-up over full-token denoising on the same engine.  These are
parity / scale cases (BASELINE.json configs[0], [2], [3] (one rank's share), [4] at bf16); the bench line is bench.py.
"""
from __future__ import annotations

import argparse
import contextlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench as B  # noqa: E402
from regione_amd import RegionEHelper, synth  # noqa: E402


def weights_stream(cfg, device, seed):
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    for name, shape in synth.flux_param_shapes(cfg).items():
        if name.endswith(".bias"):
            t = torch.randn(shape, generator=g, device=device, dtype=torch.float32) * 0.01
        elif len(shape) == 1:
            t = 1.0 + 0.1 * torch.randn(shape, generator=g, device=device, dtype=torch.float32)
        else:
            t = torch.randn(shape, generator=g, device=device, dtype=torch.float32) * 0.02
        yield name, t.to(torch.bfloat16)


def expected_ids(h_tok, w_tok, box):
    """Rectangle `box` of 'edited' tokens -> after cross-3 erosion and 5x5 dilation: box grown by one ring."""
    r0, r1, c0, c1 = box
    m = torch.zeros(h_tok, w_tok, dtype=torch.bool)
    m[max(r0 - 1, 0):min(r1 + 1, h_tok), max(c0 - 1, 0):min(c1 + 1, w_tok)] = True
    return torch.nonzero(m.flatten()).squeeze(1)


def make_box(h_tok, w_tok, frac):
    side = int(round((frac * h_tok * w_tok) ** 0.5))
    bs = max(side - 2, 3)
    r0, c0 = (h_tok - bs) // 2, (w_tok - bs) // 2
    return (r0, r0 + bs, c0, c0 + bs)


def run_case(name, family, size, frac, device, cfg_scale=None, T=512, Tn=None, timed_edits=1, vanilla_runs=2, steps=28,
             helper_kw=None, fp8=False):
    from regione_amd.harness import flux as HF, step1x as HS, qwen as HQ
    from oracle import regione_oracle as O      # checker only: derive_schedule / psnr
    h_tok = w_tok = size // 16
    L = h_tok * w_tok
    if family == "flux":
        cfg = synth.FluxConfig()
        pipe = HF.FluxKontextPipeline(HF.FluxTransformer2DModel(cfg, device).load_state_dict_stream(weights_stream(cfg, device, 42)))
        defaults = dict(threshold=0.88, cache_threshold=0.04)       # BASELINE configs[1] threshold
        fam_key = "flux"
    elif family in ("step1x", "step1x_v1p2"):
        cfg = synth.FluxConfig(guidance_embeds=False)
        tr = HS.Step1XEditTransformer2DModel(cfg, device).load_state_dict_stream(weights_stream(cfg, device, 42))
        pipe = HS.Step1XEditPipeline(tr) if family == "step1x" else HS.Step1XEditPipelineV1P2(tr)
        defaults = {}
        fam_key = family
    else:
        cfg = synth.FluxConfig(**synth.QWEN)
        pipe = HQ.QwenImageEditPipeline(HQ.QwenImageTransformer2DModel(cfg, device).load_state_dict_stream(weights_stream(cfg, device, 42)))
        defaults = {}
        fam_key = "qwen"
    if fp8:                                  # OCP e4m3fn block-GEMM weights, per-output-channel scales
        pipe.transformer.quantize_fp8_()
    torch.cuda.synchronize()
    Tn = Tn or T
    lat, img, prompt, pooled = synth.make_edit_inputs(h_tok, w_tok, T, cfg, seed=110, dtype=torch.bfloat16)
    _, _, nprompt, npooled = synth.make_edit_inputs(h_tok, w_tok, Tn, cfg, seed=111, dtype=torch.bfloat16)
    lat, img, prompt, nprompt = lat.to(device), img.to(device), prompt.to(device), nprompt.to(device)
    pooled = pooled.to(device) if pooled is not None else None
    npooled = npooled.to(device) if npooled is not None else None

    helper = RegionEHelper(pipe)
    with contextlib.redirect_stdout(sys.stderr):
        helper.set_params(**dict(defaults, **(helper_kw or {})))
    cfgd = dict(helper.config) if hasattr(helper, "config") else {}
    box = make_box(h_tok, w_tok, frac)

    def edit(trace=None, **extra):
        kw = dict(image=img, prompt_embeds=prompt, height=size, width=size, latents=lat, return_dict=False,
                  num_inference_steps=steps, **extra)
        if trace is not None:
            kw["trace"] = trace
        if family == "flux":
            kw.update(pooled_prompt_embeds=pooled, guidance_scale=2.5)
            if cfg_scale:
                kw.update(true_cfg_scale=cfg_scale, negative_prompt_embeds=nprompt, negative_pooled_prompt_embeds=npooled)
        elif family.startswith("step1x"):
            kw.update(pooled_prompt_embeds=pooled, negative_prompt_embeds=nprompt, negative_pooled_prompt_embeds=npooled,
                      true_cfg_scale=cfg_scale or 6.0)
        else:
            kw.update(negative_prompt_embeds=nprompt, true_cfg_scale=cfg_scale or 4.0)
        return pipe(**kw)[0]

    def timed(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            o = edit()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n, o

    for _ in range(vanilla_runs):      # full-token denoising (the first run also warms allocator / tables)
        tv, van = timed(1)
    helper.enable()
    B.install_region_injection(pipe, h_tok, w_tok, box, img[0:1], seed=7)
    trace = {}
    out = edit(trace)                  # warm + characterise
    tr_s, out = timed(timed_edits)
    M = pipe._regione_manager
    ids = M.edited_ids.squeeze(0).cpu()
    want = expected_ids(h_tok, w_tok, box)
    kinds = "".join(trace["kind"])
    thr = cfgd.get("cache_threshold", {"flux": 0.04, "qwen": 0.03}.get(fam_key, 0.02))
    plan = "".join(O.derive_schedule(L, fam_key, cfgd.get("warmup_step", 6), cfgd.get("post_step", 2), cfgd.get("refresh_step", "16"),
                                     thr, n=steps, gamma=cfgd.get("gamma"))).replace("S", "F")
    res = dict(case=name, family=family, size=size, L=L, T=T, T_neg=Tn, K_e=int(ids.numel()), edited_frac=ids.numel() / L,
               plan=kinds, plan_matches_reference_logic=(kinds == plan), ids_match_constructed_region=bool(torch.equal(ids, want)),
               finite=bool(torch.isfinite(out.float()).all()), steps=steps, regione_edit_s=tr_s, regione_steps_per_s=steps / tr_s,
               full_token_edit_s=tv, full_token_steps_per_s=steps / tv, speedup=tv / tr_s,
               latent_psnr_vs_full_token_random_weights_db=float(O.psnr(out.cpu(), van.cpu())), cfg_scale=cfg_scale,
               peak_mem_gb=torch.cuda.max_memory_allocated() / 2 ** 30, weights="fp8 e4m3fn + per-channel scale" if fp8 else "bf16")
    # GPU time per denoising step by kind (an event at every callback_on_step_end), and - `--ktimer` - the per-shape
    # launch table of one more edit (HIP events around every GEMM / attention launch, like bench.py)
    evs = [torch.cuda.Event(enable_timing=True)]
    evs[0].record()

    def cb(p, i, t, kw):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        evs.append(e)
        return {}
    tr2 = {}
    edit(tr2, callback_on_step_end=cb)
    torch.cuda.synchronize()
    by = {}
    for k, m in zip(tr2["kind"], [a.elapsed_time(b) for a, b in zip(evs[:-1], evs[1:])]):
        by.setdefault(k, []).append(m)
    res["step_ms_by_kind"] = {k: dict(n=len(v), avg_ms=round(sum(v) / len(v), 3), min_ms=round(min(v), 3)) for k, v in by.items()}
    if KTIMER:
        from regione_amd import ops
        kt = B.KernelTimer()
        kt.wrap(ops)
        edit()
        torch.cuda.synchronize()
        kt.unwrap()
        res["kernels"] = {k: {kk: (round(vv, 2) if isinstance(vv, float) else vv) for kk, vv in v.items()} for k, v in kt.summary().items()}
        res["gemm_shapes"] = kt.shape_table(16)
    # what the footprint is made of (hard part 3: the RIKV cache is 6.0 GB per CFG branch for FLUX at 1024^2)
    slab = lambda t: t.numel() * t.element_size()
    rikv = {}
    tr = pipe.transformer
    for blk in list(tr.transformer_blocks) + list(tr.single_transformer_blocks):
        for tag, (k_slab, vt_slab, _skv) in getattr(blk.attn.processor, "caches", {}).items():
            rikv[str(tag)] = rikv.get(str(tag), 0) + slab(k_slab) + slab(vt_slab)
    res["rikv_cache_gb_by_branch"] = {k: round(v / 2 ** 30, 3) for k, v in rikv.items()}
    res["rikv_cache_gb"] = round(sum(rikv.values()) / 2 ** 30, 3)
    res["allocated_after_edit_gb"] = round(torch.cuda.memory_allocated() / 2 ** 30, 3)
    helper.disable()
    # the injection closure (scheduler.step -> pipe) and the patched hooks form reference cycles: without a collection the
    # previous case's 24-41 GB trunk stays allocated into the next case and `peak_mem_gb` accumulates
    tr = blk = None
    del pipe, helper
    import gc
    gc.collect()
    torch.cuda.empty_cache()
    torch.cuda.reset_peak_memory_stats()
    res["allocated_after_release_gb"] = round(torch.cuda.memory_allocated() / 2 ** 30, 3)
    return res


KTIMER = False


CASES = {
    "flux_sweep": [("flux_1024_ke%02d" % int(f * 100), "flux", 1024, f, {}) for f in (0.05, 0.15, 0.25, 0.50)],
    "flux_cfg": [("flux_1024_truecfg6_ke25 (configs[3], one rank's image)", "flux", 1024, 0.25, dict(cfg_scale=6.0))],
    "step1x_512": [("step1x_v1p1_512_cfg6 (configs[0] on the GPU)", "step1x", 512, 0.25, dict(cfg_scale=6.0))],
    # the reference's own headline row: Step1X-Edit v1p1 at 1024^2, 2.572x (assets/result.jpg, README.md:23) - K_e 15 / 25 % here
    "step1x_1024": [("step1x_v1p1_1024_cfg6_ke%02d (the reference's published 2.572x configuration)" % int(f * 100), "step1x", 1024, f,
                     dict(cfg_scale=6.0, vanilla_runs=1)) for f in (0.15, 0.25)],
    "step1x_v1p2_2048": [("step1x_v1p2_2048_cfg6 bf16 28 steps (configs[4] shape at 28 steps)", "step1x_v1p2",
                          2048, 0.25, dict(cfg_scale=6.0, Tn=384, vanilla_runs=1))],
    "step1x_v1p2_2048_50": [("step1x_v1p2_2048_cfg6 bf16 50 steps, gamma re-sampled to 49 entries (configs[4] on bf16 weights)",
                             "step1x_v1p2", 2048, 0.25,
                             dict(cfg_scale=6.0, Tn=384, vanilla_runs=1, steps=50,
                                  helper_kw=dict(num_inference_steps=50, gamma="resample", warmup_step=10, post_step=4,
                                                 refresh_step="28", cache_threshold=0.02)))],
    "step1x_v1p2_2048_50_fp8": [("step1x_v1p2_2048_cfg6 fp8 weights 50 steps, gamma re-sampled to 49 entries (configs[4])",
                                 "step1x_v1p2", 2048, 0.25,
                                 dict(cfg_scale=6.0, Tn=384, vanilla_runs=1, steps=50, fp8=True,
                                      helper_kw=dict(num_inference_steps=50, gamma="resample", warmup_step=10, post_step=4,
                                                     refresh_step="28", cache_threshold=0.02)))],
    "qwen_sweep": [("qwen_image_edit_1024_cfg4_ke%02d" % int(f * 100), "qwen", 1024, f, dict(cfg_scale=4.0, Tn=384, vanilla_runs=1))
                   for f in (0.05, 0.15)],
    "qwen_quick": [("qwen_image_edit_1024_cfg4 (configs[2]), one full-token run", "qwen", 1024, 0.25, dict(cfg_scale=4.0, Tn=384, vanilla_runs=1))],
    "flux_cfg_quick": [("flux_1024_truecfg6_ke25, one full-token run", "flux", 1024, 0.25, dict(cfg_scale=6.0, vanilla_runs=1))],
    "qwen_1024": [("qwen_image_edit_1024_cfg4 (configs[2])", "qwen", 1024, 0.25, dict(cfg_scale=4.0, Tn=384))],
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("cases", nargs="*", default=[c for c in CASES if not c.startswith("step1x_v1p2_2048_50") and c not in ("qwen_sweep", "qwen_quick", "flux_cfg_quick")])
    ap.add_argument("--out", default=None)
    ap.add_argument("--ktimer", action="store_true", help="per-shape launch table of one more edit (HIP events around every GEMM / attention launch)")
    args = ap.parse_args()
    global KTIMER
    KTIMER = args.ktimer
    device = torch.device("cuda", 0)
    torch.cuda.set_device(device)
    results = []
    for group in args.cases:
        for name, family, size, frac, kw in CASES[group]:
            t0 = time.perf_counter()
            try:
                r = run_case(name, family, size, frac, device, **kw)
            except Exception as e:          # keep going: one failing configuration must not hide the others
                import traceback
                traceback.print_exc()
                r = dict(case=name, error=f"{type(e).__name__}: {e}")
            r["case_wall_s"] = time.perf_counter() - t0
            print(json.dumps(r), flush=True)
            results.append(r)
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        json.dump(results, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
