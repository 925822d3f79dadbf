#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE itself (build container only).

    python tools/gen_golden.py            # rewrites every fixture

The reference has no tests or golden vectors, so every vector is produced
here by importing /root/reference/RegionE under the stub `diffusers` of tools/ref_stubs.py and
driving the reference's own functions with seeded synthetic tensors.  Only the resulting data
(inputs + expected outputs) is committed; no reference source travels.

Fixture sets:
  G1 arp_*        token_selector: ids, raw mask, post-morphology mask; fp32 and bf16 condition
  G2 morph        remove_scattered_points on random / structured masks
  G3/G4/G5/G7 loop_*   the reference's *own* RegionEFluxKontextPipeline.__call__ run end to end
                  with an elementwise fake transformer: per-step kind (F/R/C), AVD ratio,
                  noise_pred, latents, prev_refresh trace, edited ids
  G5 avd_*        AVD decision vectors at seq-lens 1024 / 4096 / 16384
  G6 kv_*         attention-processor K/V-cache protocol at toy dims (store / update / plain)
  E2E toy_*       reference __call__ + reference transformer forward + reference processors
                  around [EXT] restated blocks at toy dims
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import ref_stubs  # noqa: E402
from regione_amd import synth  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def pack(d):
    """torch -> numpy; bf16 stored as its uint16 bit pattern under key + '__bf16'."""
    out = {}
    for k, v in d.items():
        if isinstance(v, torch.Tensor):
            if v.dtype == torch.bfloat16:
                out[k + "__bf16"] = v.contiguous().view(torch.int16).numpy().view(np.uint16)
            elif v.dtype == torch.float16:
                out[k + "__f16"] = v.contiguous().view(torch.int16).numpy().view(np.uint16)
            else:
                out[k] = v.numpy()
        else:
            out[k] = np.asarray(v)
    return out


ONLY = os.environ.get("GOLDEN_ONLY", "")       # substring filter: regenerate just the matching fixtures


def wanted(name):
    return ONLY in name


def save(name, d):
    if "_plan_" in name:
        # full-length plan fixtures (L = 4096 / 16384): keep the step plan and per-step scalars only
        ids = d.get("edited_ids")
        d = {k: v for k, v in d.items() if np.asarray(v if not isinstance(v, torch.Tensor) else v.float().numpy()).size <= 64}
        if ids is not None:
            d["n_edited"] = int(ids.numel())
            d["edited_ids_sum"] = int(ids.long().sum())
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **pack(d))
    print(f"wrote {path}  ({os.path.getsize(path) / 1024:.1f} KiB)")


# ------------------------------------------------------------------------------------------
def gen_arp(ns):
    """Inputs come from synth.arp_case(seed, ...) (seeded torch-CPU RNG, same image on the GPU
    box); the fixture holds the seeds, an fp64 checksum of the inputs (RNG-drift guard), the
    16x16 inputs in full, and every expected output."""
    u = ns.flux_utils
    d, idx = {}, 0
    for (h, w) in [(16, 16), (32, 32), (64, 64), (50, 83), (128, 128)]:
        for cond_dtype in (torch.float32, torch.bfloat16):
            for thr in (0.80, 0.88, 0.93):
                for ed in (True, False):
                    if h >= 64 and (not ed or thr == 0.80):
                        continue
                    seed = 1000 + idx
                    est, cond_in = synth.arp_case(seed, h, w, cond_dtype)
                    L = h * w
                    sim = torch.sum(torch.nn.functional.normalize(est, dim=-1) *
                                    torch.nn.functional.normalize(cond_in, dim=-1), dim=-1)
                    raw = (sim <= thr).squeeze(0)
                    e, un = u.token_selector(est, cond_in, thr, similarity_type="cosine", height=h * 16,
                                             width=w * 16, erosion_dilation=ed, patch_size=2, vae_scale_factor=8)
                    final = torch.zeros(L, dtype=torch.uint8)
                    final[e[0]] = 1
                    c = dict(h=h, w=w, thr=thr, ed=ed, seed=seed, bf16=int(cond_dtype == torch.bfloat16),
                             chk=float(est.double().sum() + cond_in.double().sum()),
                             sim=sim.squeeze(0), raw=np.packbits(raw.numpy().astype(np.uint8)),
                             final=np.packbits(final.numpy()), edited=e.squeeze(0).to(torch.int32),
                             unedited=un.squeeze(0).to(torch.int32))
                    if h == 16:
                        c["est"], c["cond"] = est, cond_in
                    for k, v in c.items():
                        d[f"c{idx}_{k}"] = v
                    idx += 1
    d["n"] = idx
    save("arp", d)


def gen_arp_adv(ns):
    """Adversarial partition cases (SURVEY.md section 7 hard-part 1 / 8c G1): every row's reference similarity lies within
    +-4 ulp of the threshold, on both sides and exactly on it.  Rows are found by a vectorised bisection on the mixing
    angle between the condition row and an orthogonal direction, evaluated with the SAME torch expressions token_selector
    uses (utils.py:310-312); the expected outputs come from the reference's token_selector on the assembled tensor."""
    u = ns.flux_utils
    Fn = torch.nn.functional.normalize
    d, idx = {}, 0
    h = w = 16
    L = h * w
    for cond_dtype in (torch.float32, torch.bfloat16):
        for thr in (0.80, 0.88, 0.93):
            g = torch.Generator().manual_seed(4000 + idx)
            R = 20000
            cond = torch.randn(1, R, 64, generator=g).to(cond_dtype)
            c64 = cond[0].double()
            uvec = c64 / c64.norm(dim=-1, keepdim=True)
            n = torch.randn(R, 64, generator=g, dtype=torch.float64)
            n = n - (n * uvec).sum(-1, keepdim=True) * uvec
            vvec = n / n.norm(dim=-1, keepdim=True)
            mag = torch.rand(R, 1, generator=g, dtype=torch.float64) * 7.5 + 0.5

            def sims(theta):
                est = ((torch.cos(theta)[:, None] * uvec + torch.sin(theta)[:, None] * vvec) * mag).float()[None]
                return est, torch.sum(Fn(est, dim=-1) * Fn(cond, dim=-1), dim=-1)[0]
            th0 = float(np.arccos(thr))
            lo = torch.full((R,), th0 - 0.02, dtype=torch.float64)      # smaller angle -> larger similarity
            hi = torch.full((R,), th0 + 0.02, dtype=torch.float64)
            target = torch.randint(-4, 5, (R,), generator=g)            # wanted offset in ulps of thr
            thr32 = np.float32(thr)
            ulp = float(np.spacing(thr32))
            goal = (torch.full((R,), float(thr32), dtype=torch.float64) + target.double() * ulp)
            for _ in range(60):
                mid = (lo + hi) / 2
                _, s = sims(mid)
                above = s.double() > goal
                lo = torch.where(above, mid, lo)
                hi = torch.where(above, hi, mid)
            est, s = sims((lo + hi) / 2)
            off = torch.round((s.double() - float(thr32)) / ulp).long()
            # stratified pick: as equal a share of every offset in [-4, 4] as the candidates allow
            pick = []
            for o in range(-4, 5):
                cand = torch.nonzero(off == o)[:, 0]
                pick.append(cand[: (L + 8) // 9 + 2])
            pick = torch.cat(pick)
            pick = pick[torch.randperm(pick.numel(), generator=g)][:L]
            assert pick.numel() == L, (idx, pick.numel())
            est_c, cond_c = est[:, pick].contiguous(), cond[:, pick].contiguous()
            sim = torch.sum(Fn(est_c, dim=-1) * Fn(cond_c, dim=-1), dim=-1)
            offs = torch.round((sim[0].double() - float(thr32)) / ulp).long()
            assert int(offs.abs().max()) <= 4 and int((offs == 0).sum()) > 0 and int((offs > 0).sum()) > 0
            raw = (sim <= thr).squeeze(0)
            for ed in (False, True):
                e, un = u.token_selector(est_c, cond_c, thr, similarity_type="cosine", height=h * 16, width=w * 16,
                                         erosion_dilation=ed, patch_size=2, vae_scale_factor=8)
                final = torch.zeros(L, dtype=torch.uint8)
                final[e[0]] = 1
                pair = idx // 2                      # the two erosion settings share one input pair
                d[f"p{pair}_est"], d[f"p{pair}_cond"] = est_c, cond_c
                c = dict(h=h, w=w, thr=thr, ed=ed, bf16=int(cond_dtype == torch.bfloat16), pair=pair,
                         sim=sim.squeeze(0), ulp_offset=offs.to(torch.int8), raw=np.packbits(raw.numpy().astype(np.uint8)),
                         final=np.packbits(final.numpy()), edited=e.squeeze(0).to(torch.int32),
                         unedited=un.squeeze(0).to(torch.int32))
                for k, v in c.items():
                    d[f"c{idx}_{k}"] = v
                idx += 1
            print(f"   arp_adv thr={thr} bf16={cond_dtype == torch.bfloat16}: offsets", np.bincount(offs.numpy() + 4, minlength=9),
                  " raw ones", int(raw.sum()))
    d["n"] = idx
    save("arp_adv", d)


def gen_morph(ns):
    u = ns.flux_utils
    torch.manual_seed(7)
    d, i = {}, 0
    for (h, w) in [(8, 8), (16, 16), (64, 64), (50, 83), (5, 7), (128, 128)]:
        for p in (0.2, 0.5, 0.8, 0.97):
            m = (torch.rand(h, w) < p).float()
            if p == 0.97:
                m[0, :] = 1; m[:, 0] = 1; m[-1, :] = 1; m[:, -1] = 1   # border stress
            r = u.remove_scattered_points(m, 5, "square")
            er = u.morphological_erosion(m, u.create_kernel(3, "cross"))
            d[f"c{i}_in"] = m.to(torch.uint8)
            d[f"c{i}_eroded"] = er.reshape(h, w).to(torch.uint8)
            d[f"c{i}_out"] = r.reshape(h, w).to(torch.uint8)
            i += 1
    for full in (0.0, 1.0):
        m = torch.full((12, 12), full)
        d[f"c{i}_in"] = m.to(torch.uint8)
        d[f"c{i}_eroded"] = u.morphological_erosion(m, u.create_kernel(3, "cross")).reshape(12, 12).to(torch.uint8)
        d[f"c{i}_out"] = u.remove_scattered_points(m).reshape(12, 12).to(torch.uint8)
        i += 1
    d["n"] = i
    save("morph", d)


# ------------------------------------------------------------------------------------------
class FakeTransformer:
    """Elementwise stand-in for the DiT used to pin the LOOP (not the blocks):
    v = (x - target[token]) * (1/sigma_t), computed in fp32 and rounded to x.dtype.
    With sigma_last = 0 the reference's one-step estimate (inplace.py:650) is then ~target."""

    def __init__(self, target_full, w_tok, L):
        self.config = ref_stubs._Cfg(in_channels=64, guidance_embeds=True)
        self.transformer_blocks = []
        self.single_transformer_blocks = []
        self.target, self.w_tok, self.L = target_full, w_tok, L

    def __call__(self, hidden_states=None, timestep=None, img_ids=None, **kw):
        tok = (img_ids[:, 0] * self.L + img_ids[:, 1] * self.w_tok + img_ids[:, 2]).long()
        n = hidden_states.shape[1]
        k = float(1.0 / timestep.float()[0].item())
        v = (hidden_states.float() - self.target[tok[:n]][None]) * k
        return (v.to(hidden_states.dtype),)


def make_fake_pipeline(ns, transformer, latents, image_latents, ids_full, prompt, pooled, txt_len):
    import diffusers
    pipe = diffusers.FluxKontextPipeline()
    pipe.scheduler = ref_stubs.FlowMatchEulerDiscreteScheduler()
    pipe.transformer = transformer
    L = latents.shape[1]
    text_ids = torch.zeros(txt_len, 3)
    pipe.encode_prompt = lambda **k: ((k.get("prompt_embeds") if k.get("prompt_embeds") is not None and k["prompt_embeds"].dim() == 3 and k["prompt_embeds"].shape[1] > 1 else prompt),
                                      (k.get("pooled_prompt_embeds") if k.get("pooled_prompt_embeds") is not None and k["pooled_prompt_embeds"].shape[-1] > 1 else pooled), text_ids)
    pipe.prepare_latents = lambda *a, **k: (latents.clone(), image_latents.clone(), ids_full[:L].clone(),
                                            ids_full[L:].clone())
    return pipe


def run_reference_loop(ns, pipe, cfg, h_tok, w_tok, record, extra=None):
    """Drive the reference __call__ and record per-step state through monkey-patched hooks."""
    ip = ns.flux
    ip.warp_modules(pipe, **cfg)
    sch = pipe.scheduler
    orig_step = sch.step
    orig_mstep = ip.MANAGER.step

    def step_hook(model_output, timestep, sample, **kw):
        record["noise_pred"].append(model_output.clone())
        out = orig_step(model_output, timestep, sample, **kw)
        record["prev_sample"].append(out[0].clone())
        return out

    def mstep_hook(latent, latent_ids):
        out = orig_mstep(latent, latent_ids)
        record["len"].append(out[0].shape[1])
        record["ids_len"].append(out[1].shape[0])
        record["prev_refresh"].append(-1 if ip.MANAGER.prev_refresh_step is None else ip.MANAGER.prev_refresh_step)
        record["next_refresh"].append(-1 if ip.MANAGER.next_refresh_step is None else ip.MANAGER.next_refresh_step)
        record["latents"].append(out[0].clone())
        return out

    sch.step = step_hook
    ip.MANAGER.step = mstep_hook
    record["calls"] = []
    tr = pipe.transformer
    if isinstance(tr, torch.nn.Module):
        handle = tr.register_forward_pre_hook(
            lambda mod, a, kw: record["calls"].append((ip.MANAGER.current_step, kw["hidden_states"].shape[1])),
            with_kwargs=True)
    else:
        inner = tr.__class__.__call__

        class _Rec(tr.__class__):
            def __call__(self, **kw):
                record["calls"].append((ip.MANAGER.current_step, kw["hidden_states"].shape[1]))
                return inner(self, **kw)
        tr.__class__ = _Rec
    try:
        img = torch.zeros(1, 16, 2 * h_tok, 2 * w_tok)           # takes the "already latent" branch
        out = pipe(image=img, prompt_embeds=torch.zeros(1, 1, 1), pooled_prompt_embeds=torch.zeros(1, 1),
                   height=h_tok * 16, width=w_tok * 16, max_area=h_tok * 16 * w_tok * 16,
                   num_inference_steps=28, guidance_scale=2.5, output_type="latent", return_dict=False, **(extra or {}))
    finally:
        ip.MANAGER.step = orig_mstep
    L_ = record["noise_pred"][0].shape[1]
    called = dict(record["calls"])
    record["kinds"] = ["C" if i not in called else ("F" if called[i] == 2 * L_ else "R") for i in range(28)]
    record["final"] = out[0]
    record["edited_ids"] = ip.MANAGER.edited_ids.clone()
    record["unedited_ids"] = ip.MANAGER.unedited_ids.clone()
    return record


def gen_loop(ns):
    cfgs = [
        ("loop_bf16_32", 32, 32, torch.bfloat16, dict(threshold=0.93, cache_threshold=0.04, refresh_step="16"), (8, 20, 6, 22)),
        ("loop_f32_16", 16, 16, torch.float32, dict(threshold=0.88, cache_threshold=0.02, refresh_step="12,20"), (4, 11, 4, 11)),
        ("loop_bf16_50x83", 50, 83, torch.bfloat16, dict(threshold=0.93, cache_threshold=0.04, refresh_step="16"), (10, 30, 20, 60)),
        # BASELINE sizes (1024^2 -> 64x64 tokens, 2048^2 -> 128x128): plan / per-step checksums only
        ("loop_plan_64", 64, 64, torch.bfloat16, dict(threshold=0.88, cache_threshold=0.04, refresh_step="16"), (16, 48, 16, 48)),
        ("loop_plan_128", 128, 128, torch.bfloat16, dict(threshold=0.88, cache_threshold=0.04, refresh_step="16"), (32, 96, 32, 96)),
    ]
    for name, h, w, dtype, over, box in cfgs:
        if not wanted(name):
            continue
        fcfg = synth.FluxConfig()
        latents, image_latents, _, _ = synth.make_edit_inputs(h, w, 8, fcfg, seed=42, dtype=dtype)
        L = h * w
        tgt = synth.region_target(h, w, box, image_latents, seed=7, ramp=0.9)
        target_full = torch.cat([tgt, image_latents[0].float()], 0)
        ids_full = synth.flux_latent_ids(h, w)
        tr = FakeTransformer(target_full, w, L)
        pipe = make_fake_pipeline(ns, tr, latents, image_latents, ids_full, torch.zeros(1, 8, 4).to(dtype),
                                  torch.zeros(1, 4).to(dtype), 8)
        cfg = dict(num_inference_steps=28, warmup_step=6, post_step=2, refresh_step="16", threshold=0.93,
                   cache_threshold=0.04, erosion_dilation=True)
        cfg.update(over)
        rec = {k: [] for k in ("noise_pred", "prev_sample", "len", "ids_len", "prev_refresh", "next_refresh", "latents")}
        run_reference_loop(ns, pipe, cfg, h, w, rec)
        kinds = rec["kinds"]
        d = dict(h=h, w=w, box=np.array(box), seed=42, tseed=7, ramp=0.9, bf16=int(dtype == torch.bfloat16),
                 chk=float(latents.double().sum() + image_latents.double().sum() + tgt.double().sum()),
                 kinds=np.array(kinds), len=np.array(rec["len"]), ids_len=np.array(rec["ids_len"]),
                 prev_refresh=np.array(rec["prev_refresh"]), next_refresh=np.array(rec["next_refresh"]),
                 final=rec["final"], edited_ids=rec["edited_ids"].to(torch.int32),
                 unedited_ids=rec["unedited_ids"].to(torch.int32),
                 threshold=cfg["threshold"], cache_threshold=cfg["cache_threshold"], refresh_step=cfg["refresh_step"],
                 np_sum=np.array([float(x.double().sum()) for x in rec["noise_pred"]]),
                 lat_sum=np.array([float(x.double().sum()) for x in rec["latents"]]),
                 lat_abs=np.array([float(x.double().abs().sum()) for x in rec["latents"]]))
        if L <= 1024:
            for i in (4, 5, 6, 7, 15, 16, 25, 26):
                d[f"lat{i}"] = rec["latents"][i]
            for i in (5, 6, 7, 15, 16):
                d[f"np{i}"] = rec["noise_pred"][i]
        save(name, d)
        print("   kinds:", "".join(kinds), " K_e =", rec["edited_ids"].shape[1], "/", L)


def gen_avd(ns):
    """AVD decision vectors from the reference arithmetic (inplace.py:295-313) at three seq-lens.
    The decision code is inline in __call__, so it is exercised through the loop fixtures; here
    we additionally tabulate ratio_i with the reference's dtype path for each L."""
    ip = ns.flux
    d = {}
    for L in (1024, 4096, 16384):
        sch = ref_stubs.FlowMatchEulerDiscreteScheduler()
        sig = np.linspace(1.0, 1 / 28, 28)
        mu = ns.flux_utils.calculate_shift(L)
        sch.set_timesteps(sigmas=sig, mu=mu)
        ts = sch.timesteps
        ratios = [float("nan")]
        for i in range(1, 28):
            ratios.append(float(ip.gamma[i - 1] * (1 + (ts[i] - ts[i - 1]) / 1000)))
        d[f"L{L}_timesteps"] = ts
        d[f"L{L}_sigmas"] = sch.sigmas
        d[f"L{L}_ratio"] = np.array(ratios, dtype=np.float32)
    d["gamma"] = ip.gamma
    # the other families' fitted tables, and their per-step ratios at the BASELINE lengths
    import importlib
    fams = {"step1x": ns.step1x, "qwen": ns.qwen, "step1x_v1p2": ns.step1x_v1p2}
    try:
        fams["qwen_plus"] = importlib.import_module("RegionE.QwenImageEditPlus.inplace")
    except Exception as e:                       # noqa: BLE001 - the Plus patch set needs more of diffusers than the stubs give
        print("   (QwenImageEditPlus not importable here:", type(e).__name__, e, ")")
    for fam, mod in fams.items():
        d[f"gamma_{fam}"] = mod.gamma
        for L in (1024, 4096, 16384):
            ts = d[f"L{L}_timesteps"]
            d[f"{fam}_L{L}_ratio"] = np.array([float("nan")] + [float(mod.gamma[i - 1] * (1 + (ts[i] - ts[i - 1]) / 1000))
                                                                  for i in range(1, 28)], dtype=np.float32)
    save("avd", d)


# ------------------------------------------------------------------------------------------
def load_weights_into(module, w):
    sd = module.state_dict()
    missing = [k for k in sd if k not in w]
    assert not missing, missing[:5]
    module.load_state_dict({k: w[k].clone() for k in sd})


def gen_kv_and_toy(ns):
    ip = ns.flux
    for name, dtype in (("toy_bf16", torch.bfloat16), ("toy_f32", torch.float32)):
        cfg = synth.FluxConfig(**synth.TOY)
        h = w = 16
        L, T = h * w, 32
        wts = synth.make_flux_weights(cfg, seed=42, dtype=dtype, w_std=0.05)
        model = ref_stubs.FluxTransformer2DModel(in_channels=cfg.in_channels, n_double=cfg.n_double,
                                                 n_single=cfg.n_single, heads=cfg.heads, head_dim=cfg.head_dim,
                                                 joint_dim=cfg.joint_dim, pooled_dim=cfg.pooled_dim,
                                                 axes_dim=cfg.axes_dim).to(dtype)
        load_weights_into(model, wts)
        model.eval()
        latents, image_latents, prompt, pooled = synth.make_edit_inputs(h, w, T, cfg, seed=42, dtype=dtype)
        # make the condition latent close to where the (random) model drives x0 outside a box so a
        # non-trivial region appears: run the reference for the warm-up, then craft the condition.
        ids_full = synth.flux_latent_ids(h, w)
        pipe = make_fake_pipeline(ns, model, latents, image_latents, ids_full, prompt, pooled, T)
        rcfg = dict(num_inference_steps=28, warmup_step=6, post_step=2, refresh_step="16", threshold=0.5,
                    cache_threshold=0.04, erosion_dilation=True)
        rec = {k: [] for k in ("noise_pred", "prev_sample", "len", "ids_len", "prev_refresh", "next_refresh", "latents")}
        # pass 1: find the one-step estimate at step warmup-1 with an arbitrary condition
        with torch.no_grad():
            run_reference_loop(ns, pipe, rcfg, h, w, rec)
        sch_sig = pipe.scheduler.sigmas
        x5 = latents if len(rec["latents"]) < 5 else rec["latents"][4]
        est = x5.float() + (sch_sig[-1] - sch_sig[5]) * rec["noise_pred"][5].float()
        box = torch.zeros(h, w, dtype=torch.bool)
        box[4:12, 3:11] = True
        cond = est.clone()
        g = torch.Generator().manual_seed(3)
        cond = cond + 0.35 * torch.randn(cond.shape, generator=g) * cond.std()
        cond[0, box.reshape(-1)] = torch.randn(int(box.sum()), 64, generator=g)
        # NOTE: the condition latent feeds the model on full steps, so the estimate moves; the box
        # is therefore only approximately reproduced - the fixture records whatever the reference does.
        image_latents2 = cond.to(dtype)
        pipe = make_fake_pipeline(ns, model, latents, image_latents2, ids_full, prompt, pooled, T)
        rec = {k: [] for k in ("noise_pred", "prev_sample", "len", "ids_len", "prev_refresh", "next_refresh", "latents")}
        with torch.no_grad():
            run_reference_loop(ns, pipe, rcfg, h, w, rec)
        wsum = float(sum(v.double().abs().sum() for v in wts.values()))
        d = dict(h=h, w=w, T=T, seed=42, bf16=int(dtype == torch.bfloat16), image_latents=image_latents2,
                 chk=float(latents.double().sum() + prompt.double().sum() + pooled.double().sum()),
                 len=np.array(rec["len"]), prev_refresh=np.array(rec["prev_refresh"]), final=rec["final"],
                 kinds=np.array(rec["kinds"]), edited_ids=rec["edited_ids"].to(torch.int32),
                 unedited_ids=rec["unedited_ids"].to(torch.int32), weight_abs_sum=wsum,
                 threshold=rcfg["threshold"], w_std=0.05,
                 np_sum=np.array([float(x.double().sum()) for x in rec["noise_pred"]]),
                 lat_sum=np.array([float(x.double().sum()) for x in rec["latents"]]))
        for i in (0, 5, 6, 14, 15, 27):
            d[f"np{i}"] = rec["noise_pred"][i]
            d[f"lat{i}"] = rec["latents"][i]
        # raw K cache of the first double block / V cache of the first single block after the run
        d["kcache_d0"] = model.transformer_blocks[0].attn.processor.k_cache
        d["vcache_s0"] = model.single_transformer_blocks[0].attn.processor.v_cache
        save(name, d)
        print("   lens:", rec["len"], " K_e =", rec["edited_ids"].shape[1])


def gen_toy_edges(ns):
    """Edge cases of the partition through the reference's own FLUX __call__ + forward + processors at toy dims, on the
    condition latent of `toy_bf16`: threshold 1.1 -> EVERY token edited (K_e = L: the region path runs on the full token set,
    partial K/V update of every image row) and threshold -1.0 -> NO token edited (K_e = 0: region steps carry the text rows
    only; recorded as whatever the reference does, including an exception)."""
    base = load_npz("toy_bf16")
    dtype = torch.bfloat16
    cfg = synth.FluxConfig(**synth.TOY)
    h = w = 16
    L, T = h * w, 32
    wts = synth.make_flux_weights(cfg, seed=42, dtype=dtype, w_std=0.05)
    latents, _, prompt, pooled = synth.make_edit_inputs(h, w, T, cfg, seed=42, dtype=dtype)
    ids_full = synth.flux_latent_ids(h, w)
    for name, thr in (("toy_bf16_all", 1.1), ("toy_bf16_none", -1.0)):
        model = ref_stubs.FluxTransformer2DModel(in_channels=cfg.in_channels, n_double=cfg.n_double, n_single=cfg.n_single,
                                                 heads=cfg.heads, head_dim=cfg.head_dim, joint_dim=cfg.joint_dim,
                                                 pooled_dim=cfg.pooled_dim, axes_dim=cfg.axes_dim).to(dtype)
        load_weights_into(model, wts)
        model.eval()
        pipe = make_fake_pipeline(ns, model, latents, base["image_latents"], ids_full, prompt, pooled, T)
        rcfg = dict(num_inference_steps=28, warmup_step=6, post_step=2, refresh_step="16", threshold=thr,
                    cache_threshold=0.04, erosion_dilation=True)
        rec = {k: [] for k in ("noise_pred", "prev_sample", "len", "ids_len", "prev_refresh", "next_refresh", "latents")}
        err = ""
        try:
            with torch.no_grad():
                run_reference_loop(ns, pipe, rcfg, h, w, rec)
        except Exception as e:                  # the reference's own behaviour on this input IS the fixture
            err = f"{type(e).__name__}: {e}"
        d = dict(h=h, w=w, T=T, seed=42, threshold=thr, w_std=0.05, error=err, len=np.array(rec["len"]),
                 steps_completed=len(rec["latents"]))
        if not err:
            d.update(final=rec["final"], kinds=np.array(rec["kinds"]), edited_ids=rec["edited_ids"].to(torch.int32),
                     np_sum=np.array([float(x.double().sum()) for x in rec["noise_pred"]]),
                     lat_sum=np.array([float(x.double().sum()) for x in rec["latents"]]))
            for i in (5, 6, 15, 27):
                d[f"np{i}"] = rec["noise_pred"][i]
                d[f"lat{i}"] = rec["latents"][i]
        save(name, d)
        print("  ", name, "error:" if err else "ok", err[:120], " lens:", rec["len"][:8])


class FakeTransformerB2:
    """Batch-2 elementwise stand-in for Step1X's batched CFG forward (Step1XEdit/inplace.py:381-399):
    row 0 (cond) is pulled towards target_pos, row 1 (uncond) towards target_neg."""

    def __init__(self, tpos_full, tneg_full, w_tok, L):
        self.config = ref_stubs._Cfg(in_channels=64, guidance_embeds=False)
        self.transformer_blocks, self.single_transformer_blocks = [], []
        self.t = (tpos_full, tneg_full)
        self.w_tok, self.L = w_tok, L

    def __call__(self, hidden_states=None, timestep=None, img_ids=None, **kw):
        tok = (img_ids[:, 0] * self.L + img_ids[:, 1] * self.w_tok + img_ids[:, 2]).long()
        n = hidden_states.shape[1]
        k = float(1.0 / timestep.float()[0].item())
        outs = [((hidden_states[b:b + 1].float() - self.t[b][tok[:n]][None]) * k).to(hidden_states.dtype)
                for b in range(hidden_states.shape[0])]
        return (torch.cat(outs, 0),)


def gen_step1x_loop(ns):
    """Reference RegionEStep1XEditPipeline.__call__ (batched CFG B=2, norm-rescaled CFG, Step1X gamma)
    with the elementwise fake transformer."""
    import diffusers
    ip = ns.step1x
    for name, h, w, dtype, box in (("s1x_loop_bf16_32", 32, 32, torch.bfloat16, (8, 20, 6, 22)),
                                   ("s1x_loop_f32_16", 16, 16, torch.float32, (4, 11, 4, 11)),
                                   ("s1x_plan_64", 64, 64, torch.bfloat16, (16, 48, 16, 48))):
        if not wanted(name):
            continue
        fcfg = synth.FluxConfig()
        latents, image_latents, _, _ = synth.make_edit_inputs(h, w, 8, fcfg, seed=42, dtype=dtype)
        L = h * w
        tpos = synth.region_target(h, w, box, image_latents, seed=7, ramp=0.9)
        g = torch.Generator().manual_seed(11)
        tneg = tpos + 0.05 * torch.randn(tpos.shape, generator=g)
        cond = image_latents[0].float()
        tr = FakeTransformerB2(torch.cat([tpos, cond], 0), torch.cat([tneg, cond], 0), w, L)
        pipe = diffusers.Step1XEditPipeline()
        pipe.scheduler = ref_stubs.FlowMatchEulerDiscreteScheduler()
        pipe.transformer = tr
        ids_full = synth.flux_latent_ids(h, w)
        text_ids = torch.zeros(8, 3)
        dummy = torch.zeros(1, 8, 4).to(dtype)
        pipe.encode_image = lambda image, width, height, device, n: (image, None, None, width, height)
        pipe.encode_prompt = lambda **k: (dummy, torch.ones(1, 8), text_ids)
        pipe.prepare_latents = lambda *a, **k: (latents.clone(), image_latents.clone(), ids_full[:L].clone(),
                                                ids_full[L:].clone())
        cfg = dict(num_inference_steps=28, warmup_step=6, post_step=2, refresh_step="16", threshold=0.88,
                   cache_threshold=0.02, erosion_dilation=True)
        ip.warp_modules(pipe, **cfg)
        rec = {k: [] for k in ("noise_pred", "len", "prev_refresh", "latents", "calls")}
        sch = pipe.scheduler
        orig_step, orig_mstep = sch.step, ip.MANAGER.step

        def step_hook(model_output, timestep, sample, **kw):
            rec["noise_pred"].append(model_output.clone())
            return orig_step(model_output, timestep, sample, **kw)

        def mstep_hook(latent, latent_ids):
            out = orig_mstep(latent, latent_ids)
            rec["len"].append(out[0].shape[1])
            rec["prev_refresh"].append(-1 if ip.MANAGER.prev_refresh_step is None else ip.MANAGER.prev_refresh_step)
            rec["latents"].append(out[0].clone())
            return out
        inner = tr.__class__.__call__

        class _Rec(tr.__class__):
            def __call__(self, **kw):
                rec["calls"].append((ip.MANAGER.current_step, kw["hidden_states"].shape[1]))
                return inner(self, **kw)
        tr.__class__ = _Rec
        sch.step, ip.MANAGER.step = step_hook, mstep_hook
        try:
            out = pipe(image=torch.zeros(1, 3, 8, 8), prompt_embeds=dummy, prompt_embeds_mask=torch.ones(1, 8),
                       negative_prompt_embeds=dummy, negative_prompt_embeds_mask=torch.ones(1, 8), height=h * 16,
                       width=w * 16, num_inference_steps=28, true_cfg_scale=6.0, guidance_scale=6.0,
                       output_type="latent", return_dict=False)
        finally:
            ip.MANAGER.step = orig_mstep
        called = dict(rec["calls"])
        kinds = ["C" if i not in called else ("F" if called[i] == 2 * L else "R") for i in range(28)]
        d = dict(h=h, w=w, box=np.array(box), seed=42, tseed=7, nseed=11, ramp=0.9, bf16=int(dtype == torch.bfloat16),
                 chk=float(latents.double().sum() + image_latents.double().sum() + tpos.double().sum() + tneg.double().sum()),
                 kinds=np.array(kinds), len=np.array(rec["len"]), prev_refresh=np.array(rec["prev_refresh"]),
                 final=out[0], edited_ids=ip.MANAGER.edited_ids.to(torch.int32), threshold=0.88, cache_threshold=0.02,
                 true_cfg_scale=6.0,
                 np_sum=np.array([float(x.double().sum()) for x in rec["noise_pred"]]),
                 lat_sum=np.array([float(x.double().sum()) for x in rec["latents"]]))
        for i in (4, 5, 6, 7, 15, 16, 26):
            d[f"lat{i}"] = rec["latents"][i]
        for i in (0, 5, 6, 7):
            d[f"np{i}"] = rec["noise_pred"][i]
        save(name, d)
        print("   kinds:", "".join(kinds), " K_e =", ip.MANAGER.edited_ids.shape[1], "/", L)


class FakeTransformerTagged:
    """Sequential-CFG stand-in (Step1X-v1p2 / Qwen): the branch arrives as joint_attention_kwargs['tag']."""

    def __init__(self, tpos_full, tneg_full, w_tok, L):
        self.config = ref_stubs._Cfg(in_channels=64, guidance_embeds=False)
        self.transformer_blocks, self.single_transformer_blocks = [], []
        self.t = {"cond": tpos_full, "uncond": tneg_full}
        self.w_tok, self.L = w_tok, L

    def __call__(self, hidden_states=None, timestep=None, img_ids=None, joint_attention_kwargs=None, **kw):
        tgt = self.t[joint_attention_kwargs["tag"]]
        tok = (img_ids[:, 0] * self.L + img_ids[:, 1] * self.w_tok + img_ids[:, 2]).long()
        n = hidden_states.shape[1]
        k = float(1.0 / timestep.float()[0].item())
        return (((hidden_states.float() - tgt[tok[:n]][None]) * k).to(hidden_states.dtype),)


def gen_step1x_v1p2_loop(ns):
    """Reference RegionEStep1XEditPipeline.__call__ of Step1XEditV1P2 (sequential CFG with tags, per-branch
    text lengths, norm-rescaled CFG, v1p2 gamma), reflection / thinking disabled."""
    import diffusers
    ip = ns.step1x_v1p2
    for name, h, w, dtype, box in (("s1xv2_loop_bf16_32", 32, 32, torch.bfloat16, (8, 20, 6, 22)),
                                   ("s1xv2_plan_128", 128, 128, torch.bfloat16, (32, 96, 32, 96))):
        if not wanted(name):
            continue
        fcfg = synth.FluxConfig()
        latents, image_latents, _, _ = synth.make_edit_inputs(h, w, 8, fcfg, seed=42, dtype=dtype)
        L = h * w
        tpos = synth.region_target(h, w, box, image_latents, seed=7, ramp=0.9)
        tneg = tpos + 0.05 * torch.randn(tpos.shape, generator=torch.Generator().manual_seed(11))
        cond = image_latents[0].float()
        tr = FakeTransformerTagged(torch.cat([tpos, cond], 0), torch.cat([tneg, cond], 0), w, L)
        pipe = diffusers.Step1XEditPipelineV1P2()
        pipe.scheduler = ref_stubs.FlowMatchEulerDiscreteScheduler()
        pipe.transformer = tr
        ids_full = synth.flux_latent_ids(h, w)
        dummy = torch.zeros(1, 8, 4).to(dtype)
        pe = ref_stubs._Cfg(embedding=dummy, mask=None, txt_ids=torch.zeros(8, 3), text_embeds=None, text_masks=None)
        ne = ref_stubs._Cfg(embedding=dummy[:, :5], mask=None, txt_ids=torch.zeros(5, 3), text_embeds=None, text_masks=None)
        pipe.encode_image = lambda image, width, height, size_level, device, n: (image, None, None, width, height)
        pipe.encode_prompt = lambda **k: (ne if k.get("prompt") == "" else pe)
        pipe.prepare_latents = lambda *a, **k: (latents.clone(), image_latents.clone(), ids_full[:L].clone(),
                                                ids_full[L:].clone())
        cfg = dict(num_inference_steps=28, warmup_step=6, post_step=2, refresh_step="16", threshold=0.88,
                   cache_threshold=0.02, erosion_dilation=True)
        ip.warp_modules(pipe, **cfg)
        rec = {k: [] for k in ("noise_pred", "len", "prev_refresh", "latents", "calls")}
        sch = pipe.scheduler
        orig_step, orig_mstep = sch.step, ip.MANAGER.step

        def step_hook(model_output, timestep, sample, **kw):
            rec["noise_pred"].append(model_output.clone())
            return orig_step(model_output, timestep, sample, **kw)

        def mstep_hook(latent, latent_ids):
            out = orig_mstep(latent, latent_ids)
            rec["len"].append(out[0].shape[1])
            rec["prev_refresh"].append(-1 if ip.MANAGER.prev_refresh_step is None else ip.MANAGER.prev_refresh_step)
            rec["latents"].append(out[0].clone())
            return out
        inner = tr.__class__.__call__

        class _Rec(tr.__class__):
            def __call__(self, **kw):
                rec["calls"].append((ip.MANAGER.current_step, kw["hidden_states"].shape[1]))
                return inner(self, **kw)
        tr.__class__ = _Rec
        sch.step, ip.MANAGER.step = step_hook, mstep_hook
        out = None
        try:
            out = pipe(image=torch.zeros(1, 3, 8, 8), prompt="edit", height=h * 16, width=w * 16, num_inference_steps=28,
                       true_cfg_scale=6.0, guidance_scale=6.0, output_type="latent", return_dict=False,
                       enable_thinking_mode=False, enable_reflection_mode=False)
        except RuntimeError as e:
            # reference quirk: with output_type="latent" the post-loop `if out_images` (Step1XEditV1P2/inplace.py:491)
            # evaluates a multi-element tensor; the denoise loop itself has completed and is fully recorded.
            assert "Boolean value of Tensor" in str(e) and len(rec["latents"]) == 28
        finally:
            ip.MANAGER.step = orig_mstep
        called = dict(rec["calls"])
        kinds = ["C" if i not in called else ("F" if called[i] == 2 * L else "R") for i in range(28)]
        d = dict(h=h, w=w, box=np.array(box), seed=42, tseed=7, nseed=11, ramp=0.9, bf16=int(dtype == torch.bfloat16),
                 txt_len=8, neg_txt_len=5,
                 chk=float(latents.double().sum() + image_latents.double().sum() + tpos.double().sum() + tneg.double().sum()),
                 kinds=np.array(kinds), len=np.array(rec["len"]), prev_refresh=np.array(rec["prev_refresh"]),
                 final=rec["latents"][-1], edited_ids=ip.MANAGER.edited_ids.to(torch.int32), threshold=0.88,
                 cache_threshold=0.02, true_cfg_scale=6.0,
                 np_sum=np.array([float(x.double().sum()) for x in rec["noise_pred"]]),
                 lat_sum=np.array([float(x.double().sum()) for x in rec["latents"]]))
        for i in (4, 5, 6, 7, 15, 16, 26):
            d[f"lat{i}"] = rec["latents"][i]
        save(name, d)
        print("   kinds:", "".join(kinds), " K_e =", ip.MANAGER.edited_ids.shape[1], "/", L, " out type", type(out))


class FakeTransformerQwen:
    """Qwen stand-in: 1-D `latent_ids`, branch in attention_kwargs['tag'], cache_context() manager."""

    def __init__(self, tpos_full, tneg_full):
        self.config = ref_stubs._Cfg(in_channels=64, guidance_embeds=False)
        self.transformer_blocks = []
        self.t = {"cond": tpos_full, "uncond": tneg_full}

    def cache_context(self, name):
        from contextlib import nullcontext
        return nullcontext()

    def __call__(self, hidden_states=None, timestep=None, latent_ids=None, attention_kwargs=None, **kw):
        tgt = self.t[attention_kwargs["tag"]]
        n = hidden_states.shape[1]
        k = float(1.0 / timestep.float()[0].item())
        return (((hidden_states.float() - tgt[latent_ids[:n].long()][None]) * k).to(hidden_states.dtype),)


def gen_qwen_loop(ns):
    """Reference RegionEQwenImageEditPipeline.__call__ (sequential tagged CFG, norm-preserving CFG,
    Qwen gamma, 1-D latent ids) with the elementwise fake transformer."""
    import diffusers
    ip = ns.qwen
    for name, h, w, dtype, box in (("qwen_loop_bf16_32", 32, 32, torch.bfloat16, (8, 20, 6, 22)),
                                   ("qwen_loop_f32_16", 16, 16, torch.float32, (4, 11, 4, 11)),
                                   ("qwen_plan_64", 64, 64, torch.bfloat16, (16, 48, 16, 48))):
        if not wanted(name):
            continue
        fcfg = synth.FluxConfig()
        latents, image_latents, _, _ = synth.make_edit_inputs(h, w, 8, fcfg, seed=42, dtype=dtype)
        L = h * w
        tpos = synth.region_target(h, w, box, image_latents, seed=7, ramp=0.9)
        tneg = tpos + 0.05 * torch.randn(tpos.shape, generator=torch.Generator().manual_seed(11))
        cond = image_latents[0].float()
        tr = FakeTransformerQwen(torch.cat([tpos, cond], 0), torch.cat([tneg, cond], 0))
        pipe = diffusers.QwenImageEditPipeline()
        pipe.scheduler = ref_stubs.FlowMatchEulerDiscreteScheduler()
        pipe.transformer = tr
        dummy = torch.zeros(1, 8, 4).to(dtype)
        pipe.image_processor = ref_stubs._Cfg(resize=lambda im, hh, ww: im,
                                              preprocess=lambda im, hh, ww: torch.zeros(1, 3, hh, ww))
        pipe.encode_prompt = lambda **k: ((dummy[:, :5], torch.ones(1, 5)) if k.get("prompt") == " " else (dummy, torch.ones(1, 8)))
        pipe.prepare_latents = lambda *a, **k: (latents.clone(), image_latents.clone())
        cfg = dict(num_inference_steps=28, warmup_step=6, post_step=2, refresh_step="16", threshold=0.80,
                   cache_threshold=0.03, erosion_dilation=True)
        ip.warp_modules(pipe, **cfg)
        rec = {k: [] for k in ("noise_pred", "len", "latents", "calls")}
        sch = pipe.scheduler
        orig_step, orig_mstep = sch.step, ip.MANAGER.step

        def step_hook(model_output, timestep, sample, **kw):
            rec["noise_pred"].append(model_output.clone())
            return orig_step(model_output, timestep, sample, **kw)

        def mstep_hook(latent, latent_ids):
            out = orig_mstep(latent, latent_ids)
            rec["len"].append(out[0].shape[1])
            rec["latents"].append(out[0].clone())
            return out
        inner = tr.__class__.__call__

        class _Rec(tr.__class__):
            def __call__(self, **kw):
                rec["calls"].append((ip.MANAGER.current_step, kw["hidden_states"].shape[1]))
                return inner(self, **kw)
        tr.__class__ = _Rec
        sch.step, ip.MANAGER.step = step_hook, mstep_hook
        img = ref_stubs._Cfg(size=(w * 16, h * 16))
        try:
            out = pipe(image=img, prompt="edit", negative_prompt=" ", height=h * 16, width=w * 16,
                       num_inference_steps=28, true_cfg_scale=4.0, output_type="latent", return_dict=False)
        finally:
            ip.MANAGER.step = orig_mstep
        called = dict(rec["calls"])
        kinds = ["C" if i not in called else ("F" if called[i] == 2 * L else "R") for i in range(28)]
        d = dict(h=h, w=w, box=np.array(box), seed=42, tseed=7, nseed=11, ramp=0.9, bf16=int(dtype == torch.bfloat16),
                 txt_len=8, neg_txt_len=5,
                 chk=float(latents.double().sum() + image_latents.double().sum() + tpos.double().sum() + tneg.double().sum()),
                 kinds=np.array(kinds), len=np.array(rec["len"]), final=out[0],
                 edited_ids=ip.MANAGER.edited_ids.to(torch.int32), threshold=0.80, cache_threshold=0.03,
                 true_cfg_scale=4.0,
                 np_sum=np.array([float(x.double().sum()) for x in rec["noise_pred"]]),
                 lat_sum=np.array([float(x.double().sum()) for x in rec["latents"]]))
        for i in (4, 5, 6, 7, 15, 16, 26):
            d[f"lat{i}"] = rec["latents"][i]
        save(name, d)
        print("   kinds:", "".join(kinds), " K_e =", ip.MANAGER.edited_ids.shape[1], "/", L)


def gen_toy_cfg(ns):
    """FLUX true-CFG (true_cfg_scale > 1, sequential cond / uncond forwards sharing ONE K/V cache,
    reference quirk A-4) at toy dims."""
    dtype = torch.bfloat16
    cfg = synth.FluxConfig(**synth.TOY)
    h = w = 16
    L, T = h * w, 32
    wts = synth.make_flux_weights(cfg, seed=42, dtype=dtype, w_std=0.05)
    model = ref_stubs.FluxTransformer2DModel(in_channels=cfg.in_channels, n_double=cfg.n_double, n_single=cfg.n_single,
                                             heads=cfg.heads, head_dim=cfg.head_dim, joint_dim=cfg.joint_dim,
                                             pooled_dim=cfg.pooled_dim, axes_dim=cfg.axes_dim).to(dtype)
    load_weights_into(model, wts)
    model.eval()
    latents, image_latents, prompt, pooled = synth.make_edit_inputs(h, w, T, cfg, seed=42, dtype=dtype)
    _, _, nprompt, npooled = synth.make_edit_inputs(h, w, T, cfg, seed=43, dtype=dtype)
    base = load_npz("toy_bf16")
    image_latents = base["image_latents"]
    ids_full = synth.flux_latent_ids(h, w)
    pipe = make_fake_pipeline(ns, model, latents, image_latents, ids_full, prompt, pooled, T)
    rcfg = dict(num_inference_steps=28, warmup_step=6, post_step=2, refresh_step="16", threshold=0.5,
                cache_threshold=0.04, erosion_dilation=True)
    rec = {k: [] for k in ("noise_pred", "prev_sample", "len", "ids_len", "prev_refresh", "next_refresh", "latents")}
    with torch.no_grad():
        run_reference_loop(ns, pipe, rcfg, h, w, rec, extra=dict(
            true_cfg_scale=4.0, negative_prompt_embeds=nprompt, negative_pooled_prompt_embeds=npooled))
    d = dict(h=h, w=w, T=T, seed=42, nseed=43, true_cfg_scale=4.0, threshold=0.5, w_std=0.05,
             image_latents=image_latents, len=np.array(rec["len"]), final=rec["final"], kinds=np.array(rec["kinds"]),
             edited_ids=rec["edited_ids"].to(torch.int32),
             np_sum=np.array([float(x.double().sum()) for x in rec["noise_pred"]]),
             lat_sum=np.array([float(x.double().sum()) for x in rec["latents"]]))
    for i in (0, 5, 6, 15, 27):
        d[f"np{i}"] = rec["noise_pred"][i]
        d[f"lat{i}"] = rec["latents"][i]
    save("toy_bf16_cfg", d)
    print("   lens:", rec["len"][:8], " K_e =", rec["edited_ids"].shape[1], "kinds", "".join(rec["kinds"]))


def gen_step1x_toy(ns, v1p2: bool):
    """Reference Step1X-Edit __call__ + transformer forward + attention processors around [EXT] FLUX-shaped block stubs
    at toy dims.  v1p1 (Step1XEdit/inplace.py): batched true CFG (B = 2 rows through ONE cache per processor);
    v1p2 (Step1XEditV1P2/inplace.py): sequential tagged CFG, one cache per tag, different cond / uncond text lengths."""
    import diffusers
    ip = ns.step1x_v1p2 if v1p2 else ns.step1x
    dtype = torch.bfloat16
    cfg = synth.FluxConfig(guidance_embeds=False, **synth.TOY)
    h = w = 16
    L, T = h * w, 32
    Tn = 24 if v1p2 else 32
    wts = synth.make_flux_weights(cfg, seed=5, dtype=dtype, w_std=0.05)
    model = ref_stubs.Step1XEditTransformer2DModel(in_channels=cfg.in_channels, n_double=cfg.n_double, n_single=cfg.n_single,
                                                   heads=cfg.heads, head_dim=cfg.head_dim, joint_dim=cfg.joint_dim,
                                                   pooled_dim=cfg.pooled_dim, axes_dim=cfg.axes_dim).to(dtype)
    rename = {"time_text_embed.timestep_embedder.": "time_embed.", "time_text_embed.text_embedder.": "vec_embed."}
    sd = {}
    for k, v in wts.items():
        for a, b in rename.items():
            if k.startswith(a):
                k = b + k[len(a):]
        sd[k] = v.clone()
    missing = [k for k in model.state_dict() if k not in sd]
    assert not missing, missing[:5]
    model.load_state_dict({k: sd[k] for k in model.state_dict()})
    model.eval()
    latents, image_latents, prompt, y = synth.make_edit_inputs(h, w, T, cfg, seed=9, dtype=dtype)
    _, _, nprompt, ny = synth.make_edit_inputs(h, w, Tn, cfg, seed=10, dtype=dtype)
    model.set_vec(prompt, y)
    model.set_vec(nprompt, ny)
    ids_full = synth.flux_latent_ids(h, w)

    def run(cond_latents):
        pipe = diffusers.Step1XEditPipelineV1P2() if v1p2 else diffusers.Step1XEditPipeline()
        pipe.scheduler = ref_stubs.FlowMatchEulerDiscreteScheduler()
        pipe.transformer = model
        pipe.prepare_latents = lambda *a, **k: (latents.clone(), cond_latents.clone(), ids_full[:L].clone(), ids_full[L:].clone())
        if v1p2:
            pe = ref_stubs._Cfg(embedding=prompt, mask=None, txt_ids=torch.zeros(T, 3), text_embeds=None, text_masks=None)
            ne = ref_stubs._Cfg(embedding=nprompt, mask=None, txt_ids=torch.zeros(Tn, 3), text_embeds=None, text_masks=None)
            pipe.encode_image = lambda image, width, height, size_level, device, n: (image, None, None, width, height)
            pipe.encode_prompt = lambda **k: (ne if k.get("prompt") == "" else pe)
        else:
            pipe.encode_image = lambda image, width, height, device, n: (image, None, None, width, height)
            pipe.encode_prompt = lambda **k: (k["prompt_embeds"], k["prompt_embeds_mask"], torch.zeros(k["prompt_embeds"].shape[1], 3))
        rcfg = dict(num_inference_steps=28, warmup_step=6, post_step=2, refresh_step="16", threshold=0.5,
                    cache_threshold=0.02, erosion_dilation=True)
        ip.warp_modules(pipe, **rcfg)
        rec = {k: [] for k in ("noise_pred", "len", "latents", "calls")}
        sch = pipe.scheduler
        orig_step, orig_mstep = sch.step, ip.MANAGER.step

        def step_hook(model_output, timestep, sample, **kw):
            rec["noise_pred"].append(model_output.clone())
            return orig_step(model_output, timestep, sample, **kw)

        def mstep_hook(latent, latent_ids):
            out = orig_mstep(latent, latent_ids)
            rec["len"].append(out[0].shape[1])
            rec["latents"].append(out[0].clone())
            return out
        handle = model.register_forward_pre_hook(
            lambda mod, a, kw: rec["calls"].append((ip.MANAGER.current_step, kw["hidden_states"].shape[1])), with_kwargs=True)
        sch.step, ip.MANAGER.step = step_hook, mstep_hook
        try:
            with torch.no_grad():
                if v1p2:
                    try:
                        out = pipe(image=torch.zeros(1, 3, 8, 8), prompt="edit", negative_prompt="", height=h * 16, width=w * 16,
                                   num_inference_steps=28, true_cfg_scale=4.0, output_type="latent", return_dict=False,
                                   enable_thinking_mode=False, enable_reflection_mode=False)
                    except RuntimeError as e:           # the v1p2 post-loop `if out_images` on a latent tensor
                        print("   (post-loop:", str(e)[:60], ")")
                        out = None
                else:
                    out = pipe(image=torch.zeros(1, 3, 8, 8), prompt_embeds=prompt, prompt_embeds_mask=torch.ones(1, T),
                               negative_prompt_embeds=nprompt, negative_prompt_embeds_mask=torch.ones(1, Tn), height=h * 16,
                               width=w * 16, num_inference_steps=28, true_cfg_scale=4.0, guidance_scale=6.0,
                               output_type="latent", return_dict=False)
        finally:
            ip.MANAGER.step = orig_mstep
            handle.remove()
        rec["final"] = rec["latents"][-1]
        rec["sigmas"] = sch.sigmas
        rec["edited_ids"] = ip.MANAGER.edited_ids
        called = dict(rec["calls"])
        rec["kinds"] = ["C" if i not in called else ("F" if called[i] == 2 * L else "R") for i in range(28)]
        p0 = model.transformer_blocks[0].attn.processor
        ps = model.single_transformer_blocks[0].attn.processor
        rec["caches"] = (dict(k_d0_even=p0.k_cache_even, v_s0_odd=ps.v_cache_odd) if v1p2 else dict(k_d0=p0.k_cache, v_s0=ps.v_cache))
        return rec
    rec = run(image_latents)
    x5 = rec["latents"][4]
    est = x5.float() + (rec["sigmas"][-1] - rec["sigmas"][5]) * rec["noise_pred"][5].float()
    box = torch.zeros(h, w, dtype=torch.bool)
    box[4:12, 3:11] = True
    g = torch.Generator().manual_seed(3)
    cond = est + 0.35 * torch.randn(est.shape, generator=g) * est.std()
    cond[0, box.reshape(-1)] = torch.randn(int(box.sum()), 64, generator=g)
    image_latents2 = cond.to(dtype)
    rec = run(image_latents2)
    d = dict(h=h, w=w, T=T, Tn=Tn, seed=9, nseed=10, wseed=5, w_std=0.05, threshold=0.5, cache_threshold=0.02, true_cfg_scale=4.0,
             image_latents=image_latents2, len=np.array(rec["len"]), final=rec["final"], kinds=np.array(rec["kinds"]),
             edited_ids=rec["edited_ids"].to(torch.int32),
             weight_abs_sum=float(sum(v.double().abs().sum() for v in wts.values())),
             np_sum=np.array([float(x.double().sum()) for x in rec["noise_pred"]]),
             lat_sum=np.array([float(x.double().sum()) for x in rec["latents"]]), **rec["caches"])
    for i in (0, 5, 6, 15, 27):
        d[f"np{i}"] = rec["noise_pred"][i]
        d[f"lat{i}"] = rec["latents"][i]
    name = "s1xv2_toy_bf16" if v1p2 else "s1x_toy_bf16"
    save(name, d)
    print("   kinds:", "".join(rec["kinds"]), " K_e =", rec["edited_ids"].shape[1], "/", L)


def gen_qwen_toy(ns):
    """Reference RegionEQwenImageEditPipeline.__call__ + RegionEQwenImageTransformer2DModelforward + the reference's
    two-cache tagged attention processor (QwenImageEdit/inplace.py:462-571, 725-890) around [EXT] Qwen block stubs,
    toy dims, true CFG 4.0 with different cond / uncond text lengths."""
    import diffusers
    ip = ns.qwen
    dtype = torch.bfloat16
    cfg = synth.FluxConfig(**synth.QWEN_TOY)
    h = w = 16
    L, T, Tn = h * w, 32, 24
    wts = synth.make_flux_weights(cfg, seed=6, dtype=dtype, w_std=0.05)
    model = ref_stubs.QwenImageTransformer2DModel(in_channels=cfg.in_channels, n_double=cfg.n_double, heads=cfg.heads,
                                                  head_dim=cfg.head_dim, joint_dim=cfg.joint_dim, axes_dim=cfg.axes_dim).to(dtype)
    rename = {"x_embedder.": "img_in.", "context_embedder.": "txt_in."}
    sd = {}
    for k, v in wts.items():
        for a, b in rename.items():
            if k.startswith(a):
                k = b + k[len(a):]
        sd[k] = v.clone()
    missing = [k for k in model.state_dict() if k not in sd]
    assert not missing, missing[:5]
    model.load_state_dict({k: sd[k] for k in model.state_dict()})
    model.eval()
    latents, image_latents, prompt, _ = synth.make_edit_inputs(h, w, T, cfg, seed=9, dtype=dtype)
    _, _, nprompt, _ = synth.make_edit_inputs(h, w, Tn, cfg, seed=10, dtype=dtype)

    def run(cond_latents):
        pipe = diffusers.QwenImageEditPipeline()
        pipe.scheduler = ref_stubs.FlowMatchEulerDiscreteScheduler()
        pipe.transformer = model
        pipe.image_processor = ref_stubs._Cfg(resize=lambda im, hh, ww: im,
                                              preprocess=lambda im, hh, ww: torch.zeros(1, 3, hh, ww))
        pipe.encode_prompt = lambda **k: ((nprompt, torch.ones(1, Tn)) if k.get("prompt") == " " else (prompt, torch.ones(1, T)))
        pipe.prepare_latents = lambda *a, **k: (latents.clone(), cond_latents.clone())
        rcfg = dict(num_inference_steps=28, warmup_step=6, post_step=2, refresh_step="16", threshold=0.5,
                    cache_threshold=0.03, erosion_dilation=True)
        ip.warp_modules(pipe, **rcfg)
        # the reference resizes every input image to ~1024^2 (calculate_dimensions); the toy condition latents are
        # h x w tokens, so the size helper is pinned to the toy size for this run (img_shapes must match the latents)
        orig_calc = ip.calculate_dimensions
        ip.calculate_dimensions = lambda area, ratio: (w * 16, h * 16, None)
        rec = {k: [] for k in ("noise_pred", "len", "latents", "calls")}
        sch = pipe.scheduler
        orig_step, orig_mstep = sch.step, ip.MANAGER.step

        def step_hook(model_output, timestep, sample, **kw):
            rec["noise_pred"].append(model_output.clone())
            return orig_step(model_output, timestep, sample, **kw)

        def mstep_hook(latent, latent_ids):
            out = orig_mstep(latent, latent_ids)
            rec["len"].append(out[0].shape[1])
            rec["latents"].append(out[0].clone())
            return out
        handle = model.register_forward_pre_hook(
            lambda mod, a, kw: rec["calls"].append((ip.MANAGER.current_step, kw["hidden_states"].shape[1])), with_kwargs=True)
        sch.step, ip.MANAGER.step = step_hook, mstep_hook
        img = ref_stubs._Cfg(size=(w * 16, h * 16))
        try:
            with torch.no_grad():
                out = pipe(image=img, prompt="edit", negative_prompt=" ", height=h * 16, width=w * 16,
                           num_inference_steps=28, true_cfg_scale=4.0, output_type="latent", return_dict=False)
        finally:
            ip.MANAGER.step = orig_mstep
            ip.calculate_dimensions = orig_calc
            handle.remove()
        rec["final"], rec["sigmas"] = out[0], sch.sigmas
        rec["edited_ids"] = ip.MANAGER.edited_ids
        # the reference's processors keep their caches: one (K, V) pair per CFG branch
        p0 = model.transformer_blocks[0].attn.processor
        rec["k_even"], rec["v_odd"] = p0.k_cache_even, p0.v_cache_odd
        called = dict(rec["calls"])
        rec["kinds"] = ["C" if i not in called else ("F" if called[i] == 2 * L else "R") for i in range(28)]
        return rec
    # pass 1 with an arbitrary condition -> craft a condition that gives a non-trivial region (see gen_kv_and_toy)
    rec = run(image_latents)
    x5 = rec["latents"][4]
    est = x5.float() + (rec["sigmas"][-1] - rec["sigmas"][5]) * rec["noise_pred"][5].float()
    box = torch.zeros(h, w, dtype=torch.bool)
    box[4:12, 3:11] = True
    g = torch.Generator().manual_seed(3)
    cond = est + 0.35 * torch.randn(est.shape, generator=g) * est.std()
    cond[0, box.reshape(-1)] = torch.randn(int(box.sum()), 64, generator=g)
    image_latents2 = cond.to(dtype)
    rec = run(image_latents2)
    d = dict(h=h, w=w, T=T, Tn=Tn, seed=9, nseed=10, wseed=6, w_std=0.05, threshold=0.5, cache_threshold=0.03, true_cfg_scale=4.0,
             image_latents=image_latents2, len=np.array(rec["len"]), final=rec["final"], kinds=np.array(rec["kinds"]),
             edited_ids=rec["edited_ids"].to(torch.int32),
             weight_abs_sum=float(sum(v.double().abs().sum() for v in wts.values())),
             np_sum=np.array([float(x.double().sum()) for x in rec["noise_pred"]]),
             lat_sum=np.array([float(x.double().sum()) for x in rec["latents"]]),
             kcache_even_b0=rec["k_even"], vcache_odd_b0=rec["v_odd"])
    for i in (0, 5, 6, 15, 27):
        d[f"np{i}"] = rec["noise_pred"][i]
        d[f"lat{i}"] = rec["latents"][i]
    save("qwen_toy_bf16", d)
    print("   kinds:", "".join(rec["kinds"]), " K_e =", rec["edited_ids"].shape[1], "/", L)


def load_npz(name):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import load_golden
    return load_golden(name)


def main():
    ns = ref_stubs.install()
    torch.set_num_threads(8)
    which = set(sys.argv[1:]) or {"arp", "morph", "loop", "avd", "toy"}
    if "arp" in which:
        gen_arp(ns)
    if "arp" in which or "arpadv" in which:
        gen_arp_adv(ns)
    if "morph" in which:
        gen_morph(ns)
    if "avd" in which:
        gen_avd(ns)
    if "loop" in which:
        gen_loop(ns)
    if "toy" in which:
        gen_kv_and_toy(ns)
    if "toycfg" in which or "toy" in which:
        gen_toy_cfg(ns)
    if "toyedges" in which or "toy" in which:
        gen_toy_edges(ns)
    if "step1x" in which or not sys.argv[1:]:
        gen_step1x_loop(ns)
    if "qwen" in which or not sys.argv[1:]:
        gen_qwen_loop(ns)
    if "v1p2" in which or not sys.argv[1:]:
        gen_step1x_v1p2_loop(ns)
    if "qwentoy" in which or not sys.argv[1:]:
        gen_qwen_toy(ns)
    if "s1xtoy" in which or not sys.argv[1:]:
        gen_step1x_toy(ns, v1p2=False)
    if "v1p2toy" in which or not sys.argv[1:]:
        gen_step1x_toy(ns, v1p2=True)


if __name__ == "__main__":
    main()
