"""-m gpu: the masked MMDiT block at FLUX.1-Kontext's REAL dimensions against the oracle.

One double-stream + one single-stream block at d = 3072, 24 heads x 128, d_ff = 12288, T = 512 text rows, L = L_c = 4096
(S = 8704 rows), bf16 - the same tensors on both sides:

  * FULL step with K/V store (reference inplace.py:721-725): trunk output, and every row of each layer's K slab and
    V^T slab against the oracle's raw cache pushed through the oracle's RMSNorm + RoPE (the engine caches K post-norm /
    post-RoPE and V transposed - DESIGN.md section 2);
  * REGION step (inplace.py:727-750; K_e = 1024 edited tokens, partial K/V update with the fp16 round trip of
    fused_kernels.py:80): trunk output for the T + K_e computed rows, the rewritten cache rows, and the untouched
    cache rows bit-for-bit unchanged.

Tolerance (bf16 arithmetic on both sides, different GEMM accumulation orders): PSNR >= 40 dB and relative L2 error
< 1e-2 on every compared tensor.  The oracle runs torch-CPU bf16 eager, i.e. the dtype path the reference itself runs.
"""
import pytest
import torch

from oracle import regione_oracle as O
from regione_amd import synth

# FLUX (the headline family) and the Qwen / Step1X twins (47 / 34 s of CPU oracle) all run under -m gpu


def _rel(a, b):
    a, b = a.float(), b.float()
    return float((a - b).norm() / b.norm())


def _check(name, got, ref, min_psnr=40.0, max_rel=1e-2):
    p, r = O.psnr(got.float().cpu(), ref.float().cpu()), _rel(got.cpu(), ref.cpu())
    print(f"[full-dims parity] {name}: PSNR {p:.1f} dB, rel L2 {r:.2e}")
    assert p >= min_psnr and r < max_rel, f"{name}: PSNR {p:.1f} dB, rel {r:.2e}"
    return p, r


def _slabs(proc, tag, S, H):
    """(K [H, S, 128] post-norm/post-RoPE, V [H, S, 128]) from a processor's slabs (V^T slab un-permuted)."""
    k_slab, vt_slab, skv = proc.caches[tag]
    assert skv == S
    r = torch.arange(S)
    pos = (r & ~12) | ((r & 4) << 1) | ((r & 8) >> 1)
    k = k_slab[:S].cpu().view(S, H, 128).transpose(0, 1)
    v = vt_slab.cpu()[:, pos].view(H, 128, S).permute(0, 2, 1)
    return k, v


def _oracle_kv(w, prefix, cache, heads, T, rope_k, double):
    """Oracle raw cache -> what attention consumes: RMSNorm + RoPE on K (inplace.py:760-794), V as is.  Double-stream
    caches hold the image rows only (the text K/V are recomputed every step), single-stream ones all rows."""
    a = prefix + ".attn."
    k = cache.k.view(1, -1, heads, 128).transpose(1, 2)
    v = cache.v.view(1, -1, heads, 128).transpose(1, 2)
    k = O.rms_norm(k, w[a + "norm_k.weight"])
    cos, sin = rope_k
    if double:
        cos, sin = cos[T:], sin[T:]
    return O.apply_rope(k, cos, sin)[0], v[0]


@pytest.mark.gpu
def test_flux_full_dims_double_and_single_block_full_store_then_region_update():
    from regione_amd import RegionEHelper
    from regione_amd.harness import flux as H
    dev = torch.device("cuda", 0)
    torch.set_num_threads(max(1, torch.get_num_threads()))
    cfg = synth.FluxConfig(n_double=1, n_single=1)
    assert cfg.d == 3072 and cfg.heads == 24
    wts = synth.make_flux_weights(cfg, seed=5, dtype=torch.bfloat16)
    h = w = 64
    T, L, heads = 512, 64 * 64, cfg.heads
    S = T + 2 * L
    lat, img, prompt, pooled = synth.make_edit_inputs(h, w, T, cfg, seed=9)
    guidance = torch.full([1], 2.5, dtype=torch.float32)

    # ---- HIP engine through the reference's hook points ------------------------------------------------
    tr = H.FluxTransformer2DModel(cfg, dev).load_state_dict(wts)
    pipe = H.FluxKontextPipeline(tr)
    helper = RegionEHelper(pipe)
    helper.set_params(threshold=0.88)
    helper.enable()
    M = pipe._regione_manager
    latents, image_latents, latent_ids, text_ids, _, _ = pipe.prepare(img, prompt, pooled, 1024, 1024, lat, None, 28)
    M.refresh(latents, image_latents, latent_ids, text_ids, 2, 8, 1024, 1024)
    ts = pipe.scheduler.timesteps
    prompt_d, pooled_d = prompt.to(dev), pooled.to(dev)

    def hip_forward(x, ids, step):
        M.current_step = step
        t = ts[step].expand(1).to(torch.bfloat16)
        out = pipe.transformer(hidden_states=x, timestep=t / 1000, guidance=guidance, pooled_projections=pooled_d,
                               encoder_hidden_states=prompt_d, txt_ids=text_ids, img_ids=ids,
                               joint_attention_kwargs={"tag": "cond"}, return_dict=False)[0]
        torch.cuda.synchronize()
        return out

    # ---- oracle state ------------------------------------------------------------------------------------
    ocfg = O.FluxCfg(n_double=1, n_single=1)
    st = O.RegionState()
    st.set_parameters(28, 6, 2, "16", 0.88, 0.04, True)
    ids_full = synth.flux_latent_ids(h, w)
    st.refresh(img, ids_full, T, h, w)
    caches = [O.KVCache(), O.KVCache()]
    txt_ids = torch.zeros(T, 3)
    _, ots = O.flow_match_schedule(28, L)
    assert torch.equal(ots, ts.cpu())

    def oracle_forward(x, ids, step):
        st.current_step = step
        t = ots[step].expand(1).to(torch.bfloat16)
        with torch.no_grad():
            return O.transformer_forward(wts, ocfg, st, caches, x, prompt, pooled, t / 1000, ids, txt_ids, guidance)

    # ---- FULL step + store (step warmup-1) -----------------------------------------------------------------
    x_full = torch.cat([lat, img], dim=1)
    store = M.warmup_step - 1
    out_hip = hip_forward(x_full.to(dev), latent_ids, store)
    out_ref = oracle_forward(x_full, ids_full, store)
    assert out_hip.shape == out_ref.shape == (1, 2 * L, 64)
    _check("full-step output", out_hip, out_ref)
    rope_k = O.flux_pos_embed(torch.cat((txt_ids, ids_full), 0), ocfg.axes_dim)
    procs = [pipe.transformer.transformer_blocks[0].attn.processor, pipe.transformer.single_transformer_blocks[0].attn.processor]
    prefixes = ["transformer_blocks.0", "single_transformer_blocks.0"]
    stored = []
    for proc, prefix, cache, double in zip(procs, prefixes, caches, (True, False)):
        k_hip, v_hip = _slabs(proc, "cond", S, heads)
        k_ref, v_ref = _oracle_kv(wts, prefix, cache, heads, T, rope_k, double)
        lo = T if double else 0                      # the oracle's double-stream cache holds image rows only
        _check(prefix + " K slab (store)", k_hip[:, lo:], k_ref)
        _check(prefix + " V^T slab (store)", v_hip[:, lo:], v_ref)
        stored.append((k_hip.clone(), v_hip.clone()))

    # ---- REGION step (step warmup): K_e = 1024 edited tokens, partial K/V update -------------------------------
    box = torch.zeros(h, w, dtype=torch.bool)
    box[16:48, 16:48] = True
    e = torch.nonzero(box.flatten()).squeeze(1)
    u = torch.nonzero(~box.flatten()).squeeze(1)
    assert e.numel() == 1024
    M.set_partition(e.unsqueeze(0).to(dev), u.unsqueeze(0).to(dev), box.flatten().to(torch.uint8).to(dev))
    st.edited_ids, st.unedited_ids = e.unsqueeze(0), u.unsqueeze(0)
    g = torch.Generator().manual_seed(77)
    lat_e = torch.randn(1, e.numel(), 64, generator=g).to(torch.bfloat16)          # fresh edited-token latents
    ids_e = ids_full[e]
    out_hip = hip_forward(lat_e.to(dev), latent_ids[e], M.warmup_step)
    out_ref = oracle_forward(lat_e, ids_e, st.warmup_step)
    assert out_hip.shape == out_ref.shape == (1, e.numel(), 64)
    _check("region-step output", out_hip, out_ref)
    for proc, prefix, cache, double, (k0, v0) in zip(procs, prefixes, caches, (True, False), stored):
        k_hip, v_hip = _slabs(proc, "cond", S, heads)
        k_ref, v_ref = _oracle_kv(wts, prefix, cache, heads, T, rope_k, double)
        lo = T if double else 0
        _check(prefix + " K slab (after update)", k_hip[:, lo:], k_ref)
        _check(prefix + " V^T slab (after update)", v_hip[:, lo:], v_ref)
        rows = T + e                                                     # rewritten image rows
        _check(prefix + " rewritten K rows", k_hip[:, rows], k_ref[:, rows - lo])
        _check(prefix + " rewritten V rows", v_hip[:, rows], v_ref[:, rows - lo])
        # rows the region step must not touch: condition-image rows and unedited noise rows stay bit-identical
        keep = torch.cat([T + u, T + L + torch.arange(L)])
        assert torch.equal(k_hip[:, keep], k0[:, keep]) and torch.equal(v_hip[:, keep], v0[:, keep]), prefix
        assert not torch.equal(k_hip[:, rows], k0[:, rows])


# ---------------------------------------------------------------------------------------------------------------------
# the other two trunks at their REAL dimensions (round 3): Qwen-Image-Edit (joint width 3584, txt_norm, polar rotary
# table, 1-D ids, two tagged caches with text lengths 512 / 384) and Step1X-Edit (no guidance embedder, B = 2 batched CFG)
# ---------------------------------------------------------------------------------------------------------------------
def _box_partition(h, w, dev):
    box = torch.zeros(h, w, dtype=torch.bool)
    box[16:48, 16:48] = True
    e = torch.nonzero(box.flatten()).squeeze(1)
    u = torch.nonzero(~box.flatten()).squeeze(1)
    return box, e, u


def _compare_branch_caches(label, procs, prefixes, doubles, caches, tag, wts, heads, T, L, rope_k, e=None, u=None, stored=None):
    """Every layer's K / V^T slab of CFG branch `tag` against the oracle's raw cache of that branch pushed through the oracle's
    RMSNorm + RoPE.  With `stored` (the slabs after the FULL step): the region step's rewritten rows are compared on their
    own and every other image row must be bit-identical to what the FULL step stored."""
    S = T + 2 * L
    kept = []
    for proc, prefix, cache, double in zip(procs, prefixes, caches, doubles):
        k_hip, v_hip = _slabs(proc, tag, S, heads)
        k_ref, v_ref = _oracle_kv(wts, prefix, cache, heads, T, rope_k, double)
        lo = T if double else 0
        _check(f"{label} {prefix} K slab", k_hip[:, lo:], k_ref)
        _check(f"{label} {prefix} V^T slab", v_hip[:, lo:], v_ref)
        if stored is not None:
            k0, v0 = stored[len(kept)]
            rows = T + e
            _check(f"{label} {prefix} rewritten K rows", k_hip[:, rows], k_ref[:, rows - lo])
            _check(f"{label} {prefix} rewritten V rows", v_hip[:, rows], v_ref[:, rows - lo])
            keep = torch.cat([T + u, T + L + torch.arange(L)])
            assert torch.equal(k_hip[:, keep], k0[:, keep]) and torch.equal(v_hip[:, keep], v0[:, keep]), (label, prefix)
            assert not torch.equal(k_hip[:, rows], k0[:, rows])
        kept.append((k_hip.clone(), v_hip.clone()))
    return kept


@pytest.mark.gpu
def test_qwen_full_dims_double_block_two_tagged_caches_full_store_then_region_update():
    """Qwen-Image-Edit's block at d = 3072, 24 x 128, joint width 3584, L = L_c = 4096, text lengths 512 (cond) / 384
    (uncond): the reference's forward + tagged two-cache processor (QwenImageEdit/inplace.py:462-571, :737-890) on the HIP
    engine against the oracle's Qwen mode - FULL step with store, then a REGION step (K_e = 1024), per CFG branch."""
    from regione_amd import RegionEHelper
    from regione_amd.harness import qwen as HQ
    dev = torch.device("cuda", 0)
    cfg = synth.FluxConfig(**dict(synth.QWEN, n_double=2))      # two blocks: the first one runs through the batched-branch code
    assert cfg.d == 3072 and cfg.joint_dim == 3584 and cfg.txt_norm
    wts = synth.make_flux_weights(cfg, seed=6, dtype=torch.bfloat16)
    h = w = 64
    L, heads = h * w, cfg.heads
    Ts = {"cond": 512, "uncond": 384}
    lat, img, prompt, _ = synth.make_edit_inputs(h, w, Ts["cond"], cfg, seed=9)
    _, _, nprompt, _ = synth.make_edit_inputs(h, w, Ts["uncond"], cfg, seed=10)
    embeds = {"cond": prompt, "uncond": nprompt}

    pipe = HQ.QwenImageEditPipeline(HQ.QwenImageTransformer2DModel(cfg, dev).load_state_dict(wts))
    helper = RegionEHelper(pipe)
    helper.set_params(threshold=0.88)
    helper.enable()
    M = pipe._regione_manager
    latents, image_latents, latent_ids = pipe.prepare_qwen(img, 1024, 1024, lat, None, 28)
    img_shapes = pipe._shapes(1024, 1024)
    M.refresh(latents, image_latents, latent_ids, 2, 8, 1024, 1024)
    M.txt_length = Ts["cond"]
    ts = pipe.scheduler.timesteps
    embeds_d = {k: v.to(dev) for k, v in embeds.items()}

    def hip_forward(x, ids, step, tag):
        M.current_step = step
        t = ts[step].expand(1).to(torch.bfloat16)
        out = pipe.transformer(hidden_states=x, timestep=t / 1000, encoder_hidden_states=embeds_d[tag], img_shapes=img_shapes,
                               latent_ids=ids, attention_kwargs={"tag": tag}, return_dict=False)[0]
        torch.cuda.synchronize()
        return out

    ocfg = O.FluxCfg(n_double=2, n_single=0, heads=cfg.heads, head_dim=cfg.head_dim, joint_dim=cfg.joint_dim)
    st = O.RegionState()
    st.set_parameters(28, 6, 2, "16", 0.88, 0.03, True)
    ids_full = torch.arange(2 * L)
    st.refresh(img, ids_full, Ts["cond"], h, w)
    caches = {k: [O.KVCache(), O.KVCache()] for k in Ts}
    ropes = {k: O.qwen_rope([(1, h, w), (1, h, w)], T) for k, T in Ts.items()}
    _, ots = O.flow_match_schedule(28, L)
    assert torch.equal(ots, ts.cpu())

    def oracle_forward(x, ids, step, tag):
        st.current_step, st.txt_length = step, Ts[tag]
        t = ots[step].expand(1).to(torch.bfloat16)
        with torch.no_grad():
            return O.transformer_forward(wts, ocfg, st, caches[tag], x, embeds[tag], None, t / 1000, ids, None, None,
                                         rope_full=ropes[tag])

    procs = [b.attn.processor for b in pipe.transformer.transformer_blocks]
    prefixes, doubles = ["transformer_blocks.0", "transformer_blocks.1"], (True, True)
    x_full = torch.cat([lat, img], dim=1)
    store = M.warmup_step - 1
    stored = {}
    for tag in ("cond", "uncond"):
        out_hip, out_ref = hip_forward(x_full.to(dev), latent_ids, store, tag), oracle_forward(x_full, ids_full, store, tag)
        assert out_hip.shape == out_ref.shape == (1, 2 * L, 64)
        _check(f"qwen {tag} full-step output", out_hip, out_ref)
        stored[tag] = _compare_branch_caches(f"qwen {tag} (store)", procs, prefixes, doubles, caches[tag], tag, wts, heads,
                                             Ts[tag], L, ropes[tag])
    box, e, u = _box_partition(h, w, dev)
    assert e.numel() == 1024
    M.set_partition(e.unsqueeze(0).to(dev), u.unsqueeze(0).to(dev), box.flatten().to(torch.uint8).to(dev))
    st.edited_ids, st.unedited_ids = e.unsqueeze(0), u.unsqueeze(0)
    lat_e = torch.randn(1, e.numel(), 64, generator=torch.Generator().manual_seed(77)).to(torch.bfloat16)
    seq = {}
    for tag in ("cond", "uncond"):
        out_hip = hip_forward(lat_e.to(dev), latent_ids[e], M.warmup_step, tag)
        out_ref = oracle_forward(lat_e, ids_full[e], st.warmup_step, tag)
        assert out_hip.shape == out_ref.shape == (1, e.numel(), 64)
        _check(f"qwen {tag} region-step output", out_hip, out_ref)
        _compare_branch_caches(f"qwen {tag} (update)", procs, prefixes, doubles, caches[tag], tag, wts, heads, Ts[tag], L,
                               ropes[tag], e=e, u=u, stored=stored[tag])
        seq[tag] = (out_hip.clone(), out_ref)
    # the same region step with BOTH branches as one batched pass (the pipelines' default): 2944 rows per launch,
    # per-branch K / V^T caches and rotary tables in the grouped Q/K/V epilogue - against the oracle and the two-forward outputs
    from regione_amd import dist as D
    M.current_step = M.warmup_step
    t = ts[M.warmup_step].expand(1).to(torch.bfloat16)

    def fwd(tag):
        return pipe.transformer(hidden_states=lat_e.to(dev), timestep=t / 1000, encoder_hidden_states=embeds_d[tag],
                                img_shapes=img_shapes, latent_ids=latent_ids[e], attention_kwargs={"tag": tag}, return_dict=False)[0]
    both = D.run_cfg_branches(None, lambda: fwd("cond"), lambda: fwd("uncond"), batch_on=pipe.transformer)
    torch.cuda.synchronize()
    for tag, got in zip(("cond", "uncond"), both):
        _check(f"qwen {tag} region-step output, batched pass", got, seq[tag][1])
        _check(f"qwen {tag} batched pass vs two forwards", got, seq[tag][0], min_psnr=55.0, max_rel=5e-3)
        _compare_branch_caches(f"qwen {tag} (update, batched)", procs, prefixes, doubles, caches[tag], tag, wts, heads, Ts[tag], L,
                               ropes[tag], e=e, u=u, stored=stored[tag])


@pytest.mark.gpu
def test_step1x_full_dims_double_and_single_block_batched_cfg_full_store_then_region_update():
    """Step1X-Edit's trunk (FLUX blocks, temb = time_embed + vec_embed(y), no guidance embedder) at d = 3072 with the
    reference's B = 2 batched CFG forward (Step1XEdit/inplace.py:381-399, :460-578): both batch rows of the HIP engine's
    forward against the oracle run per branch - FULL step with store, then a REGION step (K_e = 1024)."""
    from regione_amd import RegionEHelper
    from regione_amd.harness import step1x as HS
    dev = torch.device("cuda", 0)
    cfg = synth.FluxConfig(n_double=1, n_single=1, guidance_embeds=False)
    wts = synth.make_flux_weights(cfg, seed=5, dtype=torch.bfloat16)
    h = w = 64
    T, L, heads = 512, h * w, cfg.heads
    lat, img, prompt, y = synth.make_edit_inputs(h, w, T, cfg, seed=9)
    _, _, nprompt, ny = synth.make_edit_inputs(h, w, T, cfg, seed=10)
    embeds, pooled = {"cond": prompt, "uncond": nprompt}, {"cond": y, "uncond": ny}

    pipe = HS.Step1XEditPipeline(HS.Step1XEditTransformer2DModel(cfg, dev).load_state_dict(wts))
    helper = RegionEHelper(pipe)
    helper.set_params(threshold=0.88)
    helper.enable()
    M = pipe._regione_manager
    latents, image_latents, latent_ids, text_ids, _, _ = pipe.prepare(img, prompt, y, 1024, 1024, lat, None, 28)
    M.refresh(latents, image_latents, latent_ids, text_ids, 2, 8, 1024, 1024)
    ts = pipe.scheduler.timesteps
    tr = pipe.transformer
    tr.set_vec((y.to(dev), ny.to(dev)))
    pe = torch.cat((prompt, nprompt), 0).to(dev)

    def hip_forward(x, ids, step):
        M.current_step = step
        t = ts[step].expand(2).to(torch.bfloat16)
        out = tr(hidden_states=torch.cat((x, x), 0), timestep=t / 1000, guidance=None, encoder_hidden_states=pe,
                 prompt_embeds_mask=None, txt_ids=text_ids, img_ids=ids, return_dict=False)[0]
        torch.cuda.synchronize()
        return out

    ocfg = O.FluxCfg(n_double=1, n_single=1)
    st = O.RegionState()
    st.set_parameters(28, 6, 2, "16", 0.88, 0.02, True)
    ids_full = synth.flux_latent_ids(h, w)
    st.refresh(img, ids_full, T, h, w)
    caches = {k: [O.KVCache(), O.KVCache()] for k in embeds}
    txt_ids = torch.zeros(T, 3)
    _, ots = O.flow_match_schedule(28, L)

    def oracle_forward(x, ids, step, tag):
        st.current_step = step
        t = ots[step].expand(1).to(torch.bfloat16)
        with torch.no_grad():
            return O.transformer_forward(wts, ocfg, st, caches[tag], x, embeds[tag], pooled[tag], t / 1000, ids, txt_ids, None)

    rope_k = O.flux_pos_embed(torch.cat((txt_ids, ids_full), 0), ocfg.axes_dim)
    procs = [tr.transformer_blocks[0].attn.processor, tr.single_transformer_blocks[0].attn.processor]
    prefixes, doubles = ["transformer_blocks.0", "single_transformer_blocks.0"], (True, False)
    x_full = torch.cat([lat, img], dim=1)
    store = M.warmup_step - 1
    out_hip = hip_forward(x_full.to(dev), latent_ids, store)
    assert out_hip.shape == (2, 2 * L, 64)
    stored = {}
    for b, tag in enumerate(("cond", "uncond")):
        _check(f"step1x {tag} full-step output", out_hip[b:b + 1], oracle_forward(x_full, ids_full, store, tag))
        stored[tag] = _compare_branch_caches(f"step1x {tag} (store)", procs, prefixes, doubles, caches[tag], tag, wts, heads, T,
                                             L, rope_k)
    box, e, u = _box_partition(h, w, dev)
    M.set_partition(e.unsqueeze(0).to(dev), u.unsqueeze(0).to(dev), box.flatten().to(torch.uint8).to(dev))
    st.edited_ids, st.unedited_ids = e.unsqueeze(0), u.unsqueeze(0)
    lat_e = torch.randn(1, e.numel(), 64, generator=torch.Generator().manual_seed(77)).to(torch.bfloat16)
    out_hip = hip_forward(lat_e.to(dev), latent_ids[e], M.warmup_step)
    assert out_hip.shape == (2, e.numel(), 64)
    for b, tag in enumerate(("cond", "uncond")):
        _check(f"step1x {tag} region-step output", out_hip[b:b + 1], oracle_forward(lat_e, ids_full[e], st.warmup_step, tag))
        _compare_branch_caches(f"step1x {tag} (update)", procs, prefixes, doubles, caches[tag], tag, wts, heads, T, L, rope_k,
                               e=e, u=u, stored=stored[tag])
