"""Parity of the HIP engine with the oracle at the REAL DEPTH of the trunks.

Every >= 40 dB figure of rounds 1-3 was measured on 2-4 block trunks (tests/test_gpu_pipeline.py) or on one or two
full-width blocks (tests/test_gpu_full_dims.py).  This tool runs what the headline configuration actually stacks:

  (a) `full_width(family)`: FLUX.1-Kontext 19 + 38 blocks / Qwen-Image-Edit 60 blocks at d = 3072, 24 heads x 128,
      d_ff = 12288, on a small token set (16 x 16 grid -> L = L_c = 256, T = 64 text rows, so the CPU oracle's forward costs
      seconds, not minutes): one FULL step with K/V store (reference FluxKontext/inplace.py:507-555, :721-725), then one
      REGION step with the partial K/V update (:727-750; fp16 round trip of fused_kernels.py:80) - velocity of both
      steps and the LAST layer's K / V^T slabs against the oracle.  Three runs of the same two steps:
          hip   = this repository's kernels (bf16, fp32 accumulation),
          ora   = the oracle in bf16 on the CPU (the dtype path the reference itself runs: torch eager),
          truth = the oracle in fp32 on the same (bf16-valued) weights and inputs.
      Reported: PSNR(hip, ora) - the north-star number - and the triangulation err(hip, truth) vs err(ora, truth): two
      independent bf16 executions of a 57-block trunk differ from each other by about the SUM of their distances to the
      exact result, so "hip is no further from the truth than the reference's own arithmetic is" is the statement that
      stays meaningful when depth pushes two bf16 runs apart.
  (b) `narrow_loop(family)`: the same depth (19 + 38 / 60 blocks) at d = 512 (4 heads x 128) through ALL 28 steps of
      `RegionEHelper` against `oracle.denoise` (inplace.py:287-392): plan, edited ids (bit-exact) and PSNR of the final
      latents; the condition latent is crafted from a first oracle pass so that the partition finds a compact region
      (the construction tools/gen_golden.py uses for the toy fixtures).

`python tools/parity_full_depth.py [--out profiles/r04_parity_full_depth.json]` prints one line per compared tensor and
writes the report; tests/test_gpu_full_depth.py runs the same functions with assertions.  The oracle is the checker
here (test infrastructure); nothing in this file is on the product path.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import regione_oracle as O            # noqa: E402  (checker)
from regione_amd import synth                     # noqa: E402


class _Upcast(dict):
    """The bf16 state dict read as fp32, one tensor at a time (a full fp32 copy of 11.9 B parameters is 48 GB)."""

    def __getitem__(self, k):
        return dict.__getitem__(self, k).float()

    def get(self, k, default=None):
        return self[k] if k in self else default


class reversed_k_linears:
    """The oracle with every Linear summed in the OPPOSITE order along K (x and W flipped along the contraction axis: the same
    products, another fp32 accumulation order) - a second, equally valid execution of the reference's bf16 arithmetic.  Its
    distance to the plain oracle run is the run-to-run spread the reference's own dtype path has at this depth; the HIP
    path (yet another accumulation order: MFMA tiles, split-K pieces) is held to that yardstick where 40 dB is out of reach."""

    def __enter__(self):
        import torch.nn.functional as F
        self.orig = O._lin

        def _lin(w, name, x):
            return F.linear(x.flip(-1), w[name + ".weight"].flip(-1).contiguous(), w.get(name + ".bias"))
        O._lin = _lin
        return self

    def __exit__(self, *a):
        O._lin = self.orig


class FixtureMismatch(RuntimeError):
    pass


def _bits(t):
    return t.detach().to(torch.bfloat16).cpu().contiguous().view(torch.int16).numpy()


def _bf16(a):
    return torch.from_numpy(a.copy()).view(torch.bfloat16)


def rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).norm() / b.norm())


def compare(name, hip, ora, truth, rows, alt=None):
    """One compared tensor: PSNR / rel L2 of hip vs the bf16 oracle, each side's distance to the fp32 truth, and the
    oracle's own spread (its reversed-K run vs its plain run)."""
    r = dict(name=name, psnr_hip_vs_oracle_db=round(O.psnr(hip.float().cpu(), ora.float().cpu()), 2),
             rel_hip_vs_oracle=rel(hip, ora))
    if alt is not None:
        r.update(psnr_oracle_reordered_vs_oracle_db=round(O.psnr(alt.float().cpu(), ora.float().cpu()), 2),
                 rel_oracle_reordered_vs_oracle=rel(alt, ora))
    if truth is not None:
        r.update(rel_hip_vs_truth=rel(hip, truth), rel_oracle_vs_truth=rel(ora, truth),
                 psnr_hip_vs_truth_db=round(O.psnr(hip.float().cpu(), truth.float().cpu()), 2),
                 psnr_oracle_vs_truth_db=round(O.psnr(ora.float().cpu(), truth.float().cpu()), 2))
        r["hip_over_oracle_distance"] = r["rel_hip_vs_truth"] / max(r["rel_oracle_vs_truth"], 1e-30)
    rows.append(r)
    msg = f"[full-depth parity] {name}: hip vs oracle {r['psnr_hip_vs_oracle_db']:.1f} dB (rel {r['rel_hip_vs_oracle']:.2e})"
    if alt is not None:
        msg += f"; oracle reversed-K vs oracle {r['psnr_oracle_reordered_vs_oracle_db']:.1f} dB"
    if truth is not None:
        msg += (f"; vs fp32 truth: hip {r['rel_hip_vs_truth']:.2e} ({r['psnr_hip_vs_truth_db']:.1f} dB), "
                f"oracle bf16 {r['rel_oracle_vs_truth']:.2e} ({r['psnr_oracle_vs_truth_db']:.1f} dB)")
    print(msg, flush=True)
    return r


def slabs(proc, tag, S, H):
    """(K [H, S, 128] post-norm / post-RoPE, V [H, S, 128]) from a processor's slabs (V^T slab un-permuted)."""
    k_slab, vt_slab, skv = proc.caches[tag]
    assert skv == S, (skv, S)
    r = torch.arange(S)
    pos = (r & ~12) | ((r & 4) << 1) | ((r & 8) >> 1)
    k = k_slab[:S].cpu().view(S, H, 128).transpose(0, 1)
    v = vt_slab.cpu()[:, pos].view(H, 128, S).permute(0, 2, 1)
    return k, v


def oracle_kv(w, prefix, cache, heads, T, rope_k, double):
    """Oracle raw cache -> what attention consumes: RMSNorm + RoPE on K (inplace.py:760-794), V as is."""
    a = prefix + ".attn."
    k = cache.k.view(1, -1, heads, 128).transpose(1, 2)
    v = cache.v.view(1, -1, heads, 128).transpose(1, 2)
    k = O.rms_norm(k, w[a + "norm_k.weight"])
    cos, sin = rope_k
    if double:
        cos, sin = cos[T:], sin[T:]
    return O.apply_rope(k, cos, sin)[0], v[0]


def _gpu_weights(cfg, seed, w_std=0.02, host=True, **stats):
    """Weights drawn on the device (12-20 B parameters: seconds instead of minutes) -> (device dict, host copy)."""
    wd = synth.make_flux_weights(cfg, seed=seed, dtype=torch.bfloat16, device="cuda", w_std=w_std, **stats)
    wc = {k: v.cpu() for k, v in wd.items()} if host else None
    return wd, wc


def weights_fingerprint(wd, n=8):
    """A fixture's guard against another RNG stream: the bit patterns of the first / last `n` tensors (name order) summed as integers, plus
    the tensor count and the parameter count."""
    names = sorted(wd)
    pick = names[:n] + names[-n:]
    sums = [int(wd[k].contiguous().view(torch.int16).to(torch.int64).sum().item()) for k in pick]
    return dict(tensors=len(names), params=int(sum(v.numel() for v in wd.values())), picked=pick, sums=sums)


FIXTURE_ROWS_STORE, FIXTURE_ROWS_REGION = 64, 32      # slab rows kept per tensor: 24 heads x 128 x 2 B = 6 KB per row


def _box(h, w, r0, r1, c0, c1):
    box = torch.zeros(h, w, dtype=torch.bool)
    box[r0:r1, c0:c1] = True
    e = torch.nonzero(box.flatten()).squeeze(1)
    u = torch.nonzero(~box.flatten()).squeeze(1)
    return box, e, u


# ---------------------------------------------------------------------------------------------------------------------
# (a) full width, full depth: one FULL store step + one REGION step
# ---------------------------------------------------------------------------------------------------------------------
def full_width(family="flux", grid=16, T=64, truth=True, depth=None, alt=False, save_fixture=None, load_fixture=None):
    """save_fixture = path: additionally write the ORACLE side of every comparison (both velocities in full, a seeded subset of the
    last layer's slab rows) + a fingerprint of the device-drawn weights as an .npz; load_fixture = path: take the oracle side from
    such a file instead of running the oracle (the headline shape costs ~5 min of CPU oracle per pass: tests/test_gpu_full_depth.py
    runs the HIP side against the committed file in seconds).  With a fixture the slab comparisons run on the stored row subset."""
    import numpy as np
    from regione_amd import RegionEHelper
    dev = torch.device("cuda", 0)
    t_start = time.time()
    fx = dict(np.load(load_fixture, allow_pickle=False)) if load_fixture else None
    if fx is not None:
        truth = alt = False
    h = w = grid
    L = h * w
    px = grid * 16
    qwen = family == "qwen"
    if qwen:
        from regione_amd.harness import qwen as HQ
        cfg = synth.FluxConfig(**(dict(synth.QWEN) if depth is None else dict(synth.QWEN, n_double=depth)))
    else:
        from regione_amd.harness import flux as H
        cfg = synth.FluxConfig() if depth is None else synth.FluxConfig(n_double=depth[0], n_single=depth[1])
    heads = cfg.heads
    wd, wts = _gpu_weights(cfg, seed=5, host=fx is None)
    fp = weights_fingerprint(wd)
    if fx is not None:
        want = json.loads(str(fx["weights_fingerprint"]))
        if want != fp or [int(v) for v in fx["shape"]] != [grid, T, cfg.n_layers, cfg.d]:
            raise FixtureMismatch(f"{load_fixture} was generated for other weights / another shape: {want} vs {fp}")
    lat, img, prompt, pooled = synth.make_edit_inputs(h, w, T, cfg, seed=9)
    guidance = torch.full([1], 2.5, dtype=torch.float32)
    if qwen:
        pipe = HQ.QwenImageEditPipeline(HQ.QwenImageTransformer2DModel(cfg, dev).load_state_dict_stream(iter(wd.items())))
    else:
        pipe = H.FluxKontextPipeline(H.FluxTransformer2DModel(cfg, dev).load_state_dict_stream(iter(wd.items())))
    del wd
    torch.cuda.empty_cache()
    helper = RegionEHelper(pipe)
    helper.set_params(threshold=0.88)
    helper.enable()
    M = pipe._regione_manager
    if qwen:
        latents, image_latents, latent_ids = pipe.prepare_qwen(img, px, px, lat, None, 28)
        img_shapes = pipe._shapes(px, px)
        M.refresh(latents, image_latents, latent_ids, 2, 8, px, px)
        M.txt_length = T
        text_ids = None
    else:
        latents, image_latents, latent_ids, text_ids, _, _ = pipe.prepare(img, prompt, pooled, px, px, lat, None, 28)
        M.refresh(latents, image_latents, latent_ids, text_ids, 2, 8, px, px)
    ts = pipe.scheduler.timesteps
    prompt_d, pooled_d = prompt.to(dev), pooled.to(dev)

    def hip_forward(x, ids, step):
        M.current_step = step
        t = ts[step].expand(1).to(torch.bfloat16)
        if qwen:
            out = pipe.transformer(hidden_states=x, timestep=t / 1000, encoder_hidden_states=prompt_d, img_shapes=img_shapes,
                                   latent_ids=ids, attention_kwargs={"tag": "cond"}, return_dict=False)[0]
        else:
            out = pipe.transformer(hidden_states=x, timestep=t / 1000, guidance=guidance, pooled_projections=pooled_d,
                                   encoder_hidden_states=prompt_d, txt_ids=text_ids, img_ids=ids,
                                   joint_attention_kwargs={"tag": "cond"}, return_dict=False)[0]
        torch.cuda.synchronize()
        return out

    # ---- the two oracle runs: bf16 (the reference's dtype path) and fp32 (truth) ------------------------------------------
    if qwen:
        ocfg = O.FluxCfg(n_double=cfg.n_double, n_single=0, heads=cfg.heads, head_dim=cfg.head_dim, joint_dim=cfg.joint_dim)
        ids_full = torch.arange(2 * L)
        rope_full = O.qwen_rope([(1, h, w), (1, h, w)], T)
        txt_ids = None
    else:
        ocfg = O.FluxCfg(n_double=cfg.n_double, n_single=cfg.n_single)
        ids_full = synth.flux_latent_ids(h, w)
        txt_ids = torch.zeros(T, 3)
        rope_full = None
    _, ots = O.flow_match_schedule(28, L)
    assert torch.equal(ots, ts.cpu())

    class Run:
        def __init__(self, weights, dtype, fp16_roundtrip):
            self.w, self.dtype, self.rt = weights, dtype, fp16_roundtrip
            self.st = O.RegionState()
            self.st.set_parameters(28, 6, 2, "16", 0.88, 0.03 if qwen else 0.04, True)
            self.st.refresh(img.to(dtype), ids_full, T, h, w)
            self.caches = [O.KVCache() for _ in range(cfg.n_layers)]

        def forward(self, x, ids, step):
            self.st.current_step = step
            if qwen:
                self.st.txt_length = T
            t = ots[step].expand(1).to(self.dtype)
            with torch.no_grad():
                return O.transformer_forward(self.w, ocfg, self.st, self.caches, x.to(self.dtype), prompt.to(self.dtype),
                                             None if qwen else pooled.to(self.dtype), t / 1000, ids, txt_ids,
                                             None if qwen else guidance, fp16_roundtrip=self.rt, rope_full=rope_full)
    ora = Run(wts, torch.bfloat16, True) if fx is None else None
    tru = Run(_Upcast(wts), torch.float32, False) if truth else None
    alt_run = Run(wts, torch.bfloat16, True) if alt else None

    def alt_forward(x, ids, step):
        if alt_run is None:
            return None
        with reversed_k_linears():
            return alt_run.forward(x, ids, step)

    rows, timing = [], {}
    x_full = torch.cat([lat, img], dim=1)
    store = M.warmup_step - 1
    t0 = time.time()
    out_hip = hip_forward(x_full.to(dev), latent_ids, store)
    timing["hip_full_s"] = time.time() - t0
    t0 = time.time()
    out_ora = ora.forward(x_full, ids_full, store) if fx is None else _bf16(fx["vel_full__bf16"])
    timing["oracle_bf16_full_s"] = time.time() - t0
    keep_fx = {}
    if save_fixture:
        keep_fx["vel_full__bf16"] = _bits(out_ora[:, :L])
    t0 = time.time()
    out_tru = tru.forward(x_full, ids_full, store) if truth else None
    timing["oracle_fp32_full_s"] = time.time() - t0
    out_alt = alt_forward(x_full, ids_full, store)
    assert out_hip.shape == (1, 2 * L, 64) and out_ora.shape[1] in (L, 2 * L)
    print(f"[full-depth parity] {family} timing so far: {timing}", flush=True)
    compare(f"{family} full-step velocity ({cfg.n_layers} blocks, d={cfg.d})", out_hip[:, :L], out_ora[:, :L],
            out_tru[:, :L] if truth else None, rows, alt=out_alt[:, :L] if alt else None)

    rope_k = rope_full if qwen else O.flux_pos_embed(torch.cat((txt_ids, ids_full), 0), ocfg.axes_dim)
    S = T + 2 * L
    if qwen:
        last_proc, last_prefix, double = pipe.transformer.transformer_blocks[-1].attn.processor, \
            f"transformer_blocks.{cfg.n_double - 1}", True
    else:
        last_proc, last_prefix, double = pipe.transformer.single_transformer_blocks[-1].attn.processor, \
            f"single_transformer_blocks.{cfg.n_single - 1}", False
    lo = T if double else 0

    def kv_triplet():
        k_hip, v_hip = slabs(last_proc, "cond", S, heads)
        if fx is not None:
            return (k_hip[:, lo:], None, None), (v_hip[:, lo:], None, None)
        k_o, v_o = oracle_kv(wts, last_prefix, ora.caches[-1], heads, T, rope_k, double)
        if truth:
            k_t, v_t = oracle_kv(_Upcast(wts), last_prefix, tru.caches[-1], heads, T, rope_k, double)
        else:
            k_t = v_t = None
        return (k_hip[:, lo:], k_o, k_t), (v_hip[:, lo:], v_o, v_t)
    (kh, ko, kt), (vh, vo, vt) = kv_triplet()
    if fx is not None or save_fixture:      # the stored subset of slab rows (seeded; the same draw on both sides)
        sub = torch.randperm(S - lo, generator=torch.Generator().manual_seed(11))[:FIXTURE_ROWS_STORE].sort().values
    if fx is not None:
        assert torch.equal(sub, torch.from_numpy(fx["rows_store"]))
        compare(f"{family} last-layer K slab after store ({sub.numel()} fixture rows)", kh[:, sub], _bf16(fx["k_store__bf16"]), None, rows)
        compare(f"{family} last-layer V^T slab after store ({sub.numel()} fixture rows)", vh[:, sub], _bf16(fx["v_store__bf16"]), None, rows)
    else:
        compare(f"{family} last-layer K slab after store", kh, ko, kt, rows)
        compare(f"{family} last-layer V^T slab after store", vh, vo, vt, rows)
        if save_fixture:
            keep_fx.update(rows_store=sub.numpy(), k_store__bf16=_bits(ko[:, sub]), v_store__bf16=_bits(vo[:, sub]))
    k_store, v_store = kh.clone(), vh.clone()

    # ---- REGION step: K_e = (grid/2)^2 edited tokens, partial K/V update -----------------------------------------------------
    q = grid // 4
    box, e, u = _box(h, w, q, h - q, q, w - q)
    M.set_partition(e.unsqueeze(0).to(dev), u.unsqueeze(0).to(dev), box.flatten().to(torch.uint8).to(dev))
    for r in (ora, tru, alt_run):
        if r is not None:
            r.st.edited_ids, r.st.unedited_ids = e.unsqueeze(0), u.unsqueeze(0)
    lat_e = torch.randn(1, e.numel(), 64, generator=torch.Generator().manual_seed(77)).to(torch.bfloat16)
    t0 = time.time()
    out_hip = hip_forward(lat_e.to(dev), latent_ids[e], M.warmup_step)
    timing["hip_region_s"] = time.time() - t0
    t0 = time.time()
    out_ora = ora.forward(lat_e, ids_full[e], M.warmup_step) if fx is None else _bf16(fx["vel_region__bf16"])
    out_tru = tru.forward(lat_e, ids_full[e], M.warmup_step) if truth else None
    timing["oracle_region_s"] = time.time() - t0
    out_alt = alt_forward(lat_e, ids_full[e], M.warmup_step)
    assert out_hip.shape == out_ora.shape == (1, e.numel(), 64)
    compare(f"{family} region-step velocity (K_e={e.numel()})", out_hip, out_ora, out_tru, rows, alt=out_alt)
    (kh, ko, kt), (vh, vo, vt) = kv_triplet()
    rw = (T + e) - lo                                            # rewritten image rows (slab row T + id)
    if fx is not None or save_fixture:
        rws = rw[torch.randperm(rw.numel(), generator=torch.Generator().manual_seed(12))[:FIXTURE_ROWS_REGION].sort().values]
    if fx is not None:
        assert torch.equal(rws, torch.from_numpy(fx["rows_region"]))
        compare(f"{family} last-layer rewritten K rows ({rws.numel()} fixture rows)", kh[:, rws], _bf16(fx["k_region__bf16"]), None, rows)
        compare(f"{family} last-layer rewritten V rows ({rws.numel()} fixture rows)", vh[:, rws], _bf16(fx["v_region__bf16"]), None, rows)
    else:
        compare(f"{family} last-layer rewritten K rows", kh[:, rw], ko[:, rw], kt[:, rw] if truth else None, rows)
        compare(f"{family} last-layer rewritten V rows", vh[:, rw], vo[:, rw], vt[:, rw] if truth else None, rows)
    if save_fixture:
        keep_fx.update(vel_region__bf16=_bits(out_ora), rows_region=rws.numpy(), k_region__bf16=_bits(ko[:, rws]), v_region__bf16=_bits(vo[:, rws]),
                       weights_fingerprint=np.array(json.dumps(fp)), shape=np.array([grid, T, cfg.n_layers, cfg.d]),
                       generator=np.array("tools/parity_full_depth.py --cases %s_headline --save-fixture (oracle/regione_oracle.py in bf16 on the "
                                          "CPU of an MI355X box; weights drawn on the device, seed 5)" % family))
        os.makedirs(os.path.dirname(save_fixture), exist_ok=True)
        np.savez(save_fixture, **keep_fx)
        print(f"[full-depth parity] wrote fixture {save_fixture} ({os.path.getsize(save_fixture) / 1e6:.2f} MB)", flush=True)
    keep = torch.cat([T + u, T + L + torch.arange(L)]) - lo      # rows a region step must not touch
    untouched = bool(torch.equal(kh[:, keep], k_store[:, keep]) and torch.equal(vh[:, keep], v_store[:, keep]))
    print(f"[full-depth parity] {family} last-layer untouched cache rows bit-identical: {untouched}", flush=True)
    del pipe, helper
    import gc
    gc.collect()
    torch.cuda.empty_cache()
    return dict(case=f"{family}_full_width", family=family, blocks=cfg.n_layers, d=cfg.d, heads=heads, grid=[h, w], T=T,
                oracle_side=("fixture " + os.path.relpath(load_fixture, ROOT)) if fx is not None else "run in this process",
                K_e=int(e.numel()), rows=rows, untouched_rows_bit_identical=untouched, timing_s=timing,
                wall_s=round(time.time() - t_start, 1))


# ---------------------------------------------------------------------------------------------------------------------
# (b) full depth, d = 512, all 28 steps through RegionEHelper vs oracle.denoise
# ---------------------------------------------------------------------------------------------------------------------
NARROW = dict(heads=4, head_dim=128, joint_dim=512, pooled_dim=64)


# Statistics of the synthetic trunk (synth.make_flux_weights): AdaLN modulation linears scaled so gates / scales /
# shifts have std 0.1, RMSNorm weights 1.5 +- 0.3 - a trained checkpoint's regime instead of a gate-1, error-amplifying one.  Measured on
# the oracle alone (Qwen, 60 blocks, d = 512, 28 steps, CFG 4; its reversed-K run vs its plain run / its bf16 run vs its fp32 run):
# uncalibrated 39.8 / 25.8 dB; mod 0.2, norms 1.5 +- 0.3: 41.6 / 32.5; mod 0.1, norms 1.5 +- 0.3: 44.2 / 37.4; mod 0.1, norms 1.0 +- 0.1: 46.1 / 38.5
CALIBRATED = dict(mod_scale=0.1, norm_mean=1.5, norm_std=0.3)


def narrow_loop(family="flux", grid=16, T=32, device="cuda", hip=True, alt=True, width="narrow", calibrated=True, truth=False):
    """width = "narrow": d = 512 (4 heads); "full": the trunk's real width (d = 3072, 24 heads) - the whole 28-step loop at full width
    AND depth (round 5; ~1-2 min of CPU oracle per pass for FLUX, twice that for Qwen's two branches)."""
    t_start = time.time()
    # d = 512 matmuls on 544 rows: a 256-thread pool costs more in fork / join than the arithmetic (MI355X host: 320 s for the
    # Qwen case against 21 s on 8 threads); at full width the matmuls are 36 x larger: 64 threads
    threads = torch.get_num_threads()
    torch.set_num_threads(min(threads, 16 if width == "narrow" else 64))
    try:
        return _narrow_loop(family, grid, T, device, hip, alt, t_start, width, calibrated, truth)
    finally:
        torch.set_num_threads(threads)


def _narrow_loop(family, grid, T, device, hip, alt, t_start, width="narrow", calibrated=True, truth=False):
    h = w = grid
    L = h * w
    qwen, s1x = family == "qwen", family == "step1x_v1p2"
    dims = NARROW if width == "narrow" else dict(heads=24, head_dim=128, joint_dim=4096, pooled_dim=768)
    if qwen:
        cfg = synth.FluxConfig(**(dict(synth.QWEN, **dict(NARROW, pooled_dim=768)) if width == "narrow" else dict(synth.QWEN)))
        Tn, scale, thr_cache, fam = 24, 4.0, 0.03, "qwen"
    elif s1x:           # Step1X-Edit v1p2: the FLUX trunk without a guidance embedder, sequential tagged CFG, one K/V cache per tag,
        cfg = synth.FluxConfig(guidance_embeds=False, **dims)        # text lengths 32 / 24 (Step1XEditV1P2/inplace.py:398,416,833,868)
        Tn, scale, thr_cache, fam = 24, 4.0, 0.02, "step1x_v1p2"
    else:
        cfg = synth.FluxConfig(**dims)
        Tn, scale, thr_cache, fam = None, 1.0, 0.04, "flux"
    w_std = cfg.d ** -0.5
    stats = CALIBRATED if calibrated else {}
    if width == "narrow":
        wts = synth.make_flux_weights(cfg, seed=42, dtype=torch.bfloat16, w_std=w_std, **stats)
    else:               # 12-20 B parameters: drawn on the device (seconds), copied to the host for the oracle
        wts_dev, wts = _gpu_weights(cfg, seed=42, w_std=w_std, **stats)
    lat, img0, prompt, pooled = synth.make_edit_inputs(h, w, T, cfg, seed=42, dtype=torch.bfloat16)
    if qwen or s1x:
        _, _, nprompt, npooled = synth.make_edit_inputs(h, w, Tn, cfg, seed=43, dtype=torch.bfloat16)
    if qwen:
        ocfg = O.FluxCfg(n_double=cfg.n_double, n_single=0, heads=cfg.heads, head_dim=cfg.head_dim, joint_dim=cfg.joint_dim)
        ids_full = torch.arange(2 * L)
    else:
        ocfg = O.FluxCfg(n_double=cfg.n_double, n_single=cfg.n_single, **dims)
        ids_full = synth.flux_latent_ids(h, w)
    threshold = 0.5

    def oracle_run(img, trace, fp32=False):
        """fp32 = the "truth" run: the same (bf16-valued) weights and inputs computed in fp32, no fp16 round trip."""
        W = _Upcast(wts) if fp32 else wts
        cv = (lambda t: None if t is None else t.float()) if fp32 else (lambda t: t)
        st = O.RegionState()
        st.set_parameters(28, 6, 2, "16", threshold, thr_cache, True)

        def mk(pe, pp, Tt):
            caches = [O.KVCache() for _ in range(cfg.n_layers)]
            rope = O.qwen_rope([(1, h, w), (1, h, w)], Tt) if qwen else None
            pe, pp = cv(pe), cv(pp)

            def model(x, t, ids):
                st.txt_length = Tt
                tsd = t.expand(x.shape[0]).to(x.dtype)
                if qwen:
                    return O.transformer_forward(W, ocfg, st, caches, x, pe, None, tsd / 1000, ids, None, None, rope_full=rope,
                                                 fp16_roundtrip=not fp32)
                return O.transformer_forward(W, ocfg, st, caches, x, pe, pp, tsd / 1000, ids, torch.zeros(Tt, 3),
                                             None if s1x else torch.full([1], 2.5, dtype=torch.float32), fp16_roundtrip=not fp32)
            return model
        with torch.no_grad():
            if qwen or s1x:
                out = O.denoise(mk(prompt, pooled, T), st, cv(lat), cv(img), ids_full, T, h, w, family=fam, trace=trace,
                                neg_model_fn=mk(nprompt, npooled, Tn), true_cfg_scale=scale)
            else:
                out = O.denoise(mk(prompt, pooled, T), st, cv(lat), cv(img), ids_full, T, h, w, trace=trace)
        return out, st

    # pass 1 (arbitrary condition): the one-step estimate at step warmup-1 -> craft a condition with a compact region
    t0 = time.time()
    tr1 = {}
    oracle_run(img0, tr1)
    sig, _ = O.flow_match_schedule(28, L)
    est = tr1["latents"][4].float() + (sig[-1] - sig[5]) * tr1["noise_pred"][5].float()
    box = torch.zeros(h, w, dtype=torch.bool)
    box[4:12, 3:11] = True
    g = torch.Generator().manual_seed(3)
    cond = est + 0.35 * torch.randn(est.shape, generator=g) * est.std()
    cond[0, box.reshape(-1)] = torch.randn(int(box.sum()), 64, generator=g)
    img = cond.to(torch.bfloat16)
    tr_o = {}
    ref, st = oracle_run(img, tr_o)
    t_oracle = time.time() - t0
    res = dict(case=f"{family}_{width}_28_steps", family=family, blocks=cfg.n_layers, d=cfg.d, grid=[h, w], T=T,
               oracle_plan="".join(tr_o["kind"]), oracle_K_e=int(st.edited_ids.shape[1]), oracle_s=round(t_oracle, 1))
    if alt:                                 # the reference arithmetic under another summation order: its own spread
        tr_a = {}
        with reversed_k_linears():
            ref_a, st_a = oracle_run(img, tr_a)
        res.update(oracle_reordered_ids_equal=bool(torch.equal(st_a.edited_ids, st.edited_ids)),
                   oracle_reordered_plan="".join(tr_a["kind"]),
                   psnr_oracle_reordered_vs_oracle_db=round(O.psnr(ref_a, ref), 2), rel_oracle_reordered_vs_oracle=rel(ref_a, ref))
        print(f"[full-depth parity] {family} {cfg.n_layers} blocks d={cfg.d}, 28 steps: oracle with reversed-K linears vs oracle "
              f"{res['psnr_oracle_reordered_vs_oracle_db']:.1f} dB (ids equal: {res['oracle_reordered_ids_equal']})", flush=True)
    ref_t = None
    if truth:                               # the fp32 run of the same edit: both bf16 executions' distance to it
        tr_t = {}
        ref_t, st_t = oracle_run(img, tr_t, fp32=True)
        res.update(truth_plan="".join(tr_t["kind"]), truth_ids_equal=bool(torch.equal(st_t.edited_ids, st.edited_ids)),
                   psnr_oracle_vs_truth_db=round(O.psnr(ref.float(), ref_t), 2), rel_oracle_vs_truth=rel(ref, ref_t))
        if alt:
            res.update(psnr_oracle_reordered_vs_truth_db=round(O.psnr(ref_a.float(), ref_t), 2))
        print(f"[full-depth parity] {family} {cfg.n_layers} blocks d={cfg.d}, 28 steps: oracle bf16 vs its fp32 run "
              f"{res['psnr_oracle_vs_truth_db']:.1f} dB (rel {res['rel_oracle_vs_truth']:.3f}; ids equal: {res['truth_ids_equal']})", flush=True)
    res["calibrated_statistics"] = dict(CALIBRATED) if calibrated else None
    if not hip:
        return res
    from regione_amd import RegionEHelper
    src = wts if width == "narrow" else wts_dev
    if qwen:
        from regione_amd.harness import qwen as HQ
        pipe = HQ.QwenImageEditPipeline(HQ.QwenImageTransformer2DModel(cfg, device).load_state_dict_stream(iter(src.items())))
    elif s1x:
        from regione_amd.harness import step1x as HS
        pipe = HS.Step1XEditPipelineV1P2(HS.Step1XEditTransformer2DModel(cfg, device).load_state_dict_stream(iter(src.items())))
    else:
        from regione_amd.harness import flux as H
        pipe = H.FluxKontextPipeline(H.FluxTransformer2DModel(cfg, device).load_state_dict_stream(iter(src.items())))
    if width != "narrow":
        del wts_dev, src
        torch.cuda.empty_cache()
    helper = RegionEHelper(pipe)
    helper.set_params(threshold=threshold)
    helper.enable()
    tr_h = {}
    if qwen:
        out = pipe(image=img.to(device), prompt_embeds=prompt.to(device), negative_prompt_embeds=nprompt.to(device),
                   height=h * 16, width=w * 16, latents=lat.to(device), true_cfg_scale=scale, return_dict=False, trace=tr_h)[0]
    elif s1x:
        out = pipe(image=img.to(device), prompt_embeds=prompt.to(device), pooled_prompt_embeds=pooled.to(device),
                   negative_prompt_embeds=nprompt.to(device), negative_pooled_prompt_embeds=npooled.to(device), height=h * 16,
                   width=w * 16, latents=lat.to(device), true_cfg_scale=scale, return_dict=False, trace=tr_h)[0]
    else:
        out = pipe(image=img.to(device), prompt_embeds=prompt.to(device), pooled_prompt_embeds=pooled.to(device),
                   height=h * 16, width=w * 16, latents=lat.to(device), guidance_scale=2.5, return_dict=False, trace=tr_h)[0]
    out = out.cpu()
    Mg = pipe._regione_manager
    ids_equal = bool(torch.equal(Mg.edited_ids.cpu(), st.edited_ids))
    per_step = [round(O.psnr(a.cpu(), b), 1) if a.shape == b.shape else None for a, b in zip(tr_h["latents"], tr_o["latents"])]
    if ref_t is not None:
        res.update(psnr_hip_vs_truth_db=round(O.psnr(out.float(), ref_t), 2), rel_hip_vs_truth=rel(out, ref_t))
    res.update(hip_plan="".join(tr_h["kind"]), ids_bit_exact=ids_equal, hip_K_e=int(Mg.edited_ids.shape[1]),
               psnr_final_db=round(O.psnr(out, ref), 2), rel_final=rel(out, ref), psnr_per_step_db=per_step,
               wall_s=round(time.time() - t_start, 1))
    if "branches" in tr_h and "branches" in tr_o:
        # attribution of the combined velocity's distance: each branch's velocity against the oracle's at
        # every computed step, next to the combined one - the CFG combine `neg + s (pos - neg)` (norm-preserving for Qwen) multiplies
        # the two branches' (independent) differences by ~sqrt(s^2 + (s - 1)^2): 5 x = 14 dB at s = 4
        br = []
        for i in sorted(set(tr_h["branches"]) & set(tr_o["branches"])):
            (ph, nh), (po, no) = tr_h["branches"][i], tr_o["branches"][i]
            if ph.shape != po.shape:
                continue
            br.append(dict(step=i, kind=tr_h["kind"][i], psnr_cond_db=round(O.psnr(ph.cpu(), po), 2), psnr_uncond_db=round(O.psnr(nh.cpu(), no), 2),
                           psnr_combined_db=round(O.psnr(tr_h["noise_pred"][i].cpu(), tr_o["noise_pred"][i]), 2),
                           psnr_input_latents_db=(round(O.psnr(tr_h["latents"][i - 1].cpu(), tr_o["latents"][i - 1]), 2) if i else None)))
        worst = min(br, key=lambda r: r["psnr_combined_db"]) if br else None
        res.update(branch_velocities_per_step=br, worst_combined_step=worst, cfg_scale=scale,
                   cfg_amplification_expected_db=round(20 * __import__("math").log10((scale ** 2 + (scale - 1) ** 2) ** 0.5), 1))
        if worst:
            print(f"[full-depth parity] {family}: worst combined velocity at step {worst['step']} ({worst['kind']}): combined "
                  f"{worst['psnr_combined_db']} dB, cond {worst['psnr_cond_db']} dB, uncond {worst['psnr_uncond_db']} dB", flush=True)
    print(f"[full-depth parity] {family} {cfg.n_layers} blocks d={cfg.d}, 28 steps: plan {res['hip_plan']} "
          f"({'==' if res['hip_plan'] == res['oracle_plan'] else '!='} oracle), K_e {res['hip_K_e']} ids "
          f"{'bit-exact' if ids_equal else 'DIFFER'}, final latents {res['psnr_final_db']:.1f} dB "
          f"(min over steps {min(p for p in per_step if p is not None):.1f})", flush=True)
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r06_parity_full_depth.json"))
    ap.add_argument("--cases", default="flux_loop,qwen_loop,step1x_v1p2_loop,flux_width,qwen_width",
                    help="<family>_loop (d = 512, 28 steps) | <family>_fullloop (d = 3072, 28 steps, 16 x 16 grid) | <family>_width (d = 3072, "
                         "16 x 16 grid, one FULL + one REGION step) | <family>_headline (d = 3072, 64 x 64 grid, T = 512, K_e = 1024: the "
                         "bench's shape, one FULL + one REGION step; ~10 min of CPU oracle)")
    ap.add_argument("--no-truth", action="store_true", help="skip the fp32 oracle run of the full-width cases")
    ap.add_argument("--no-alt", action="store_true", help="skip the reversed-K oracle run of the full-width cases")
    ap.add_argument("--uncalibrated", action="store_true", help="<family>_loop / _fullloop: the round-1..5 statistics (modulation linears unscaled)")
    ap.add_argument("--save-fixture", default=None, help="<family>_headline: write the oracle side as this .npz (tests/golden/headline_<family>.npz)")
    ns = ap.parse_args()
    from regione_amd import build as _build
    report = dict(host_threads=torch.get_num_threads(), csrc_sha16=_build.csrc_hash(), cases=[])
    for c in ns.cases.split(","):
        fam, kind = c.rsplit("_", 1)
        if kind == "loop":
            r = narrow_loop(fam, truth=not ns.no_truth, calibrated=not ns.uncalibrated)
        elif kind == "fullloop":
            r = narrow_loop(fam, width="full", truth=not ns.no_truth, calibrated=not ns.uncalibrated)
        elif kind == "headline":
            r = full_width(fam, grid=64, T=512, truth=False, alt=False, save_fixture=ns.save_fixture)
            r["case"] = f"{fam}_headline_shape"
        else:
            r = full_width(fam, truth=not ns.no_truth, alt=not ns.no_alt)
        report["cases"].append(r)
        with open(ns.out, "w") as f:                       # after every case: a long run cut short keeps what it finished
            json.dump(report, f, indent=1)
    os.makedirs(os.path.dirname(ns.out), exist_ok=True)
    with open(ns.out, "w") as f:
        json.dump(report, f, indent=1)
    print("wrote", ns.out)


if __name__ == "__main__":
    main()
