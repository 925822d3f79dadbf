"""-m gpu: the patched pipeline end to end on the HIP kernels.

* loop fixtures (reference __call__ with an elementwise fake transformer): BIT-EXACT latents and ids;
* toy MMDiT (reference __call__ + forward + processors around [EXT] blocks): mask bit-exact, latents
  PSNR >= 40 dB vs the reference fixture (BASELINE.json north_star tolerance)."""
import numpy as np
import pytest
import torch

from oracle import regione_oracle as O
from regione_amd import RegionEHelper, synth
from regione_amd.harness import flux as H
from tests.test_host_logic import FakeTransformer

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["loop_bf16_32", "loop_f32_16", "loop_bf16_50x83"])
def test_loop_fixture_bit_exact_on_gpu(golden, name):
    g = golden(name)
    h, w = g["h"], g["w"]
    dt = torch.bfloat16 if g["bf16"] else torch.float32
    L = h * w
    lat, img, _, _ = synth.make_edit_inputs(h, w, 8, synth.FluxConfig(), seed=g["seed"], dtype=dt)
    tgt = synth.region_target(h, w, tuple(int(x) for x in g["box"]), img, seed=g["tseed"], ramp=g["ramp"])
    tr = FakeTransformer(torch.cat([tgt, img[0].float()], 0), w, L, device="cuda")
    pipe = H.FluxKontextPipeline(tr)
    helper = RegionEHelper(pipe)
    helper.set_params(threshold=g["threshold"], cache_threshold=g["cache_threshold"], refresh_step=str(g["refresh_step"]))
    helper.enable()
    trace = {}
    out = pipe(image=img, prompt_embeds=torch.zeros(1, 8, 4), pooled_prompt_embeds=torch.zeros(1, 4), height=h * 16,
               width=w * 16, latents=lat, return_dict=False, trace=trace)[0]
    assert "".join(trace["kind"]) == "".join(g["kinds"].tolist())
    M = pipe._regione_manager
    assert torch.equal(M.edited_ids.cpu().squeeze(0).int(), g["edited_ids"].squeeze(0))     # mask bit-exact
    assert torch.equal(M.unedited_ids.cpu().squeeze(0).int(), g["unedited_ids"].squeeze(0))
    assert [x.shape[1] for x in trace["latents"]] == g["len"].tolist()
    got = np.array([float(x.double().sum()) for x in trace["latents"]])
    assert np.array_equal(got, g["lat_sum"].numpy())
    for i in range(28):
        if f"lat{i}" in g:
            assert torch.equal(trace["latents"][i].cpu(), g[f"lat{i}"]), i
    assert torch.equal(out.cpu(), g["final"])


@pytest.mark.parametrize("name", ["loop_plan_64", "loop_plan_128"])
def test_loop_at_baseline_lengths_bit_exact_checksums_on_gpu(golden, name):
    """The reference loop at the BASELINE token counts (L = 4096 for 1024^2, L = 16384 for 2048^2) on the HIP
    region kernels: plan, sequence lengths, id count / id checksum and every step's latent checksum are those of
    the reference run (bit-exact path, so the checksums are compared with ==)."""
    g = golden(name)
    h, w = g["h"], g["w"]
    L = h * w
    lat, img, _, _ = synth.make_edit_inputs(h, w, 8, synth.FluxConfig(), seed=g["seed"], dtype=torch.bfloat16)
    tgt = synth.region_target(h, w, tuple(int(x) for x in g["box"]), img, seed=g["tseed"], ramp=g["ramp"])
    tr = FakeTransformer(torch.cat([tgt, img[0].float()], 0), w, L, device="cuda")
    pipe = H.FluxKontextPipeline(tr)
    helper = RegionEHelper(pipe)
    helper.set_params(threshold=g["threshold"], cache_threshold=g["cache_threshold"], refresh_step=str(g["refresh_step"]))
    helper.enable()
    trace = {}
    pipe(image=img, prompt_embeds=torch.zeros(1, 8, 4), pooled_prompt_embeds=torch.zeros(1, 4), height=h * 16,
         width=w * 16, latents=lat, return_dict=False, trace=trace)
    assert "".join(trace["kind"]) == "".join(g["kinds"].tolist())
    ids = pipe._regione_manager.edited_ids
    assert ids.numel() == g["n_edited"] and int(ids.sum()) == g["edited_ids_sum"]
    assert [x.shape[1] for x in trace["latents"]] == g["len"].tolist()
    assert np.array_equal(np.array([float(x.double().sum()) for x in trace["latents"]]), g["lat_sum"].numpy())
    assert np.array_equal(np.array([float(x.double().sum()) for x in trace["noise_pred"]]), g["np_sum"].numpy())


def _toy_pipe(wts, cfg):
    tr = H.FluxTransformer2DModel(cfg, "cuda").load_state_dict(wts)
    return H.FluxKontextPipeline(tr)


def test_toy_forward_matches_oracle_full_step(golden):
    """One full-token forward of the HIP engine vs the oracle's transformer_forward (same weights)."""
    g = golden("toy_bf16")
    h, w, T = g["h"], g["w"], g["T"]
    cfg = synth.FluxConfig(**synth.TOY)
    wts = synth.make_flux_weights(cfg, seed=42, dtype=torch.bfloat16, w_std=g["w_std"])
    lat, _, prompt, pooled = synth.make_edit_inputs(h, w, T, cfg, seed=g["seed"], dtype=torch.bfloat16)
    img = g["image_latents"]
    ids = synth.flux_latent_ids(h, w)
    st = O.RegionState()
    st.set_parameters(28, 6, 2, "16", 0.5, 0.04, True)
    st.refresh(img, ids, T, h, w)
    _, ts = O.flow_match_schedule(28, h * w)
    x = torch.cat([lat, img], 1)
    tstep = ts[0].expand(1).to(torch.bfloat16) / 1000
    guidance = torch.full([1], 2.5)
    with torch.no_grad():
        ref = O.transformer_forward(wts, O.FluxCfg(**synth.TOY), st, [O.KVCache() for _ in range(cfg.n_layers)], x,
                                    prompt, pooled, tstep, ids, torch.zeros(T, 3), guidance)
    pipe = _toy_pipe(wts, cfg)
    out = pipe.transformer(hidden_states=x.cuda(), timestep=tstep, guidance=guidance, pooled_projections=pooled.cuda(),
                           encoder_hidden_states=prompt.cuda(), txt_ids=torch.zeros(T, 3), img_ids=ids,
                           return_dict=False)[0].cpu()
    assert torch.isfinite(ref.float()).all() and ref.shape == out.shape
    err = (out.float() - ref.float()).norm() / ref.float().norm()
    assert float(err) < 2e-2, float(err)
    assert O.psnr(out, ref) > 40.0


def test_toy_mmdit_regione_vs_reference_fixture(golden):
    g = golden("toy_bf16")
    h, w, T = g["h"], g["w"], g["T"]
    cfg = synth.FluxConfig(**synth.TOY)
    wts = synth.make_flux_weights(cfg, seed=42, dtype=torch.bfloat16, w_std=g["w_std"])
    assert float(sum(v.double().abs().sum() for v in wts.values())) == g["weight_abs_sum"], "torch RNG drift"
    lat, _, prompt, pooled = synth.make_edit_inputs(h, w, T, cfg, seed=g["seed"], dtype=torch.bfloat16)
    img = g["image_latents"]
    pipe = _toy_pipe(wts, cfg)
    helper = RegionEHelper(pipe)
    helper.set_params(threshold=g["threshold"])
    helper.enable()
    trace = {}
    out = pipe(image=img.cuda(), prompt_embeds=prompt.cuda(), pooled_prompt_embeds=pooled.cuda(), height=h * 16,
               width=w * 16, latents=lat.cuda(), guidance_scale=2.5, return_dict=False, trace=trace)[0].cpu()
    assert "".join(trace["kind"]) == "".join(g["kinds"].tolist())
    M = pipe._regione_manager
    assert torch.equal(M.edited_ids.cpu().squeeze(0).int(), g["edited_ids"].squeeze(0)), "edited-token mask must be bit-exact"
    for i in (0, 5, 6, 14, 15, 27):
        assert O.psnr(trace["noise_pred"][i].cpu(), g[f"np{i}"]) > 35.0, i
        assert O.psnr(trace["latents"][i].cpu(), g[f"lat{i}"]) > 40.0, i
    assert O.psnr(out, g["final"]) >= 40.0
    # vanilla (full-token) loop on the same engine: RegionE output stays close to it
    helper.disable()
    van = pipe(image=img.cuda(), prompt_embeds=prompt.cuda(), pooled_prompt_embeds=pooled.cuda(), height=h * 16,
               width=w * 16, latents=lat.cuda(), guidance_scale=2.5, return_dict=False)[0].cpu()
    assert torch.isfinite(van.float()).all()


def test_toy_true_cfg_strict_reference_and_per_branch_caches(golden):
    """FLUX true-CFG on the HIP engine: strict_reference=True (one cache shared by both branches,
    quirk A-4) must match the reference fixture; the default (one cache per branch tag) must run and
    differ only slightly (it is the intentional fix)."""
    g = golden("toy_bf16_cfg")
    h, w, T = g["h"], g["w"], g["T"]
    cfg = synth.FluxConfig(**synth.TOY)
    wts = synth.make_flux_weights(cfg, seed=42, dtype=torch.bfloat16, w_std=g["w_std"])
    lat, _, prompt, pooled = synth.make_edit_inputs(h, w, T, cfg, seed=g["seed"], dtype=torch.bfloat16)
    _, _, nprompt, npooled = synth.make_edit_inputs(h, w, T, cfg, seed=g["nseed"], dtype=torch.bfloat16)
    img = g["image_latents"]
    outs = {}
    for strict in (True, False):
        pipe = _toy_pipe(wts, cfg)
        helper = RegionEHelper(pipe)
        helper.set_params(threshold=g["threshold"], strict_reference=strict)
        helper.enable()
        trace = {}
        outs[strict] = pipe(image=img.cuda(), prompt_embeds=prompt.cuda(), pooled_prompt_embeds=pooled.cuda(),
                            height=h * 16, width=w * 16, latents=lat.cuda(), guidance_scale=2.5,
                            true_cfg_scale=g["true_cfg_scale"], negative_prompt_embeds=nprompt.cuda(),
                            negative_pooled_prompt_embeds=npooled.cuda(), return_dict=False, trace=trace)[0].cpu()
        assert "".join(trace["kind"]) == "".join(g["kinds"].tolist())
        if strict:
            assert torch.equal(pipe._regione_manager.edited_ids.cpu().squeeze(0).int(), g["edited_ids"].squeeze(0))
            for i in (0, 5, 6, 15, 27):
                assert O.psnr(trace["noise_pred"][i].cpu(), g[f"np{i}"]) > 35.0, i
            ncache = {len(b.attn.processor.caches) for b in pipe.transformer.transformer_blocks}
            assert ncache == {1}
        else:
            ncache = {len(b.attn.processor.caches) for b in pipe.transformer.single_transformer_blocks}
            assert ncache == {2}                                   # cond + uncond
    assert O.psnr(outs[True], g["final"]) >= 40.0
    assert torch.isfinite(outs[False].float()).all() and O.psnr(outs[False], g["final"]) > 20.0


@pytest.mark.parametrize("name", ["s1x_loop_bf16_32", "s1x_loop_f32_16"])
def test_step1x_loop_fixture_bit_exact_on_gpu(golden, name):
    """Step1X-Edit patch set (batched CFG via per-branch passes, norm-rescaled CFG kernel, Step1X gamma)."""
    from tests.test_host_logic import step1x_case
    g = golden(name)
    pipe, out, trace = step1x_case(g, device="cuda")
    assert "".join(trace["kind"]) == "".join(g["kinds"].tolist())
    assert torch.equal(pipe._regione_manager.edited_ids.cpu().squeeze(0).int(), g["edited_ids"].squeeze(0))
    for i in range(28):
        if f"lat{i}" in g:
            a, b = trace["latents"][i].cpu(), g[f"lat{i}"]
            if g["bf16"]:
                # bit for bit: the rescaled CFG follows torch-CPU's row-norm tree and its bf16 cast of the pow exponent
                assert torch.equal(a, b), (i, float((a != b).float().mean()))
            else:
                # fp32 rows: torch's CPU pow is Sleef's 1-ulp powf, the kernel rounds a double pow once - they differ in the
                # last bit of the norm factor on a few per cent of the rows
                assert O.psnr(a, b) > 120.0, i
    if g["bf16"]:
        assert torch.equal(out.cpu(), g["final"])
    else:
        assert O.psnr(out.cpu(), g["final"]) > 120.0


def test_step1x_toy_mmdit_vs_oracle():
    """Step1X-Edit-shaped engine (no guidance embedder, batched CFG, per-branch K/V caches) end to end
    against the oracle with the same weights: edited ids exact, latents PSNR >= 40 dB."""
    from regione_amd.harness import step1x as HS
    cfg = synth.FluxConfig(guidance_embeds=False, **synth.TOY)
    h = w = 16
    T = 32
    wts = synth.make_flux_weights(cfg, seed=5, dtype=torch.bfloat16, w_std=0.05)
    lat, img, prompt, y = synth.make_edit_inputs(h, w, T, cfg, seed=9, dtype=torch.bfloat16)
    _, _, nprompt, ny = synth.make_edit_inputs(h, w, T, cfg, seed=10, dtype=torch.bfloat16)
    pipe = HS.Step1XEditPipeline(HS.Step1XEditTransformer2DModel(cfg, "cuda").load_state_dict(wts))
    helper = RegionEHelper(pipe)
    helper.set_params(threshold=0.1)
    helper.enable()
    out = pipe(image=img.cuda(), prompt_embeds=prompt.cuda(), pooled_prompt_embeds=y.cuda(),
               negative_prompt_embeds=nprompt.cuda(), negative_pooled_prompt_embeds=ny.cuda(), height=h * 16,
               width=w * 16, latents=lat.cuda(), true_cfg_scale=4.0, return_dict=False)[0].cpu()
    st = O.RegionState()
    st.set_parameters(28, 6, 2, "16", 0.1, 0.02, True)
    ocfg = O.FluxCfg(**synth.TOY)
    txt_ids = torch.zeros(T, 3)

    def mk(pe, pp):
        caches = [O.KVCache() for _ in range(cfg.n_layers)]          # one cache set per branch (batched rows)
        def model(x, t, img_ids):
            ts = t.expand(x.shape[0]).to(x.dtype)
            return O.transformer_forward(wts, ocfg, st, caches, x, pe, pp, ts / 1000, img_ids, txt_ids, None)
        return model
    with torch.no_grad():
        ref = O.denoise(mk(prompt, y), st, lat, img, synth.flux_latent_ids(h, w), T, h, w, family="step1x",
                        neg_model_fn=mk(nprompt, ny), true_cfg_scale=4.0)
    assert torch.equal(pipe._regione_manager.edited_ids.cpu(), st.edited_ids)
    assert O.psnr(out, ref) >= 40.0


def test_step1x_second_edit_with_another_prompt_does_not_reuse_the_first_prompts_conditioning():
    """Advisor round 5 (high): the stacked (cond, uncond) prompt embeddings were cached on the pipeline keyed on the inputs' ADDRESSES;
    the caching allocator hands the same address to the next call's same-shaped embeddings, so a second edit with another prompt ran
    on the first prompt's conditioning.  Two edits on ONE pipeline object, the first call's embeddings freed in between (so their
    addresses are recycled), must equal the same edits on fresh pipeline objects."""
    import gc
    from regione_amd.harness import step1x as HS
    cfg = synth.FluxConfig(guidance_embeds=False, **synth.TOY)
    h = w = 16
    T = 32
    wts = synth.make_flux_weights(cfg, seed=5, dtype=torch.bfloat16, w_std=0.05)
    lat, img, _, _ = synth.make_edit_inputs(h, w, T, cfg, seed=9, dtype=torch.bfloat16)

    def mk():
        pipe = HS.Step1XEditPipeline(HS.Step1XEditTransformer2DModel(cfg, "cuda").load_state_dict(wts))
        helper = RegionEHelper(pipe)
        helper.set_params(threshold=0.1)
        helper.enable()
        return pipe

    def edit(pipe, seed):
        _, _, prompt, y = synth.make_edit_inputs(h, w, T, cfg, seed=seed, dtype=torch.bfloat16)
        _, _, nprompt, ny = synth.make_edit_inputs(h, w, T, cfg, seed=seed + 100, dtype=torch.bfloat16)
        pe, npe = prompt.cuda(), nprompt.cuda()
        ptrs = (pe.data_ptr(), npe.data_ptr())
        out = pipe(image=img.cuda(), prompt_embeds=pe, pooled_prompt_embeds=y.cuda(), negative_prompt_embeds=npe,
                   negative_pooled_prompt_embeds=ny.cuda(), height=h * 16, width=w * 16, latents=lat.cuda(), true_cfg_scale=4.0,
                   return_dict=False)[0].cpu()
        del pe, npe
        gc.collect()
        return out, ptrs
    shared = mk()
    a, ptr_a = edit(shared, 21)
    b, ptr_b = edit(shared, 22)
    a_fresh, _ = edit(mk(), 21)
    b_fresh, _ = edit(mk(), 22)
    assert torch.equal(a, a_fresh)
    assert torch.equal(b, b_fresh), "the second edit did not see its own prompt"
    assert not torch.equal(a, b)
    print("[prompt cache] second call's embeddings reused the first call's addresses:", ptr_a == ptr_b)


def test_step1x_v1p2_toy_mmdit_vs_oracle_different_text_lengths():
    """Sequential tagged CFG with T_cond != T_uncond (per-text-length selection rows / RoPE tables and
    per-tag K/V caches, Step1XEditV1P2/inplace.py:833,868) on the HIP engine vs the oracle."""
    from regione_amd.harness import step1x as HS
    cfg = synth.FluxConfig(guidance_embeds=False, **synth.TOY)
    h = w = 16
    Tp, Tn = 32, 24
    wts = synth.make_flux_weights(cfg, seed=5, dtype=torch.bfloat16, w_std=0.05)
    lat, img, prompt, y = synth.make_edit_inputs(h, w, Tp, cfg, seed=9, dtype=torch.bfloat16)
    _, _, nprompt, ny = synth.make_edit_inputs(h, w, Tn, cfg, seed=10, dtype=torch.bfloat16)
    pipe = HS.Step1XEditPipelineV1P2(HS.Step1XEditTransformer2DModel(cfg, "cuda").load_state_dict(wts))
    helper = RegionEHelper(pipe)
    helper.set_params(threshold=0.1)
    helper.enable()
    out = pipe(image=img.cuda(), prompt_embeds=prompt.cuda(), pooled_prompt_embeds=y.cuda(),
               negative_prompt_embeds=nprompt.cuda(), negative_pooled_prompt_embeds=ny.cuda(), height=h * 16,
               width=w * 16, latents=lat.cuda(), true_cfg_scale=4.0, return_dict=False)[0].cpu()
    st = O.RegionState()
    st.set_parameters(28, 6, 2, "16", 0.1, 0.02, True)
    ocfg = O.FluxCfg(**synth.TOY)

    def mk(pe, pp, T):
        caches = [O.KVCache() for _ in range(cfg.n_layers)]
        def model(x, t, img_ids):
            st.txt_length = T                                    # txt_length / neg_txt_length of the branch
            ts = t.expand(x.shape[0]).to(x.dtype)
            return O.transformer_forward(wts, ocfg, st, caches, x, pe, pp, ts / 1000, img_ids, torch.zeros(T, 3), None)
        return model
    with torch.no_grad():
        ref = O.denoise(mk(prompt, y, Tp), st, lat, img, synth.flux_latent_ids(h, w), Tp, h, w, family="step1x_v1p2",
                        neg_model_fn=mk(nprompt, ny, Tn), true_cfg_scale=4.0)
    assert torch.equal(pipe._regione_manager.edited_ids.cpu(), st.edited_ids)
    assert O.psnr(out, ref) >= 40.0
    # vanilla v1p2 loop runs too
    helper.disable()
    van = pipe(image=img.cuda(), prompt_embeds=prompt.cuda(), pooled_prompt_embeds=y.cuda(),
               negative_prompt_embeds=nprompt.cuda(), negative_pooled_prompt_embeds=ny.cuda(), height=h * 16,
               width=w * 16, latents=lat.cuda(), true_cfg_scale=4.0, return_dict=False)[0]
    assert torch.isfinite(van.float()).all()


def test_qwen_toy_mmdit_vs_oracle(golden):
    """Qwen-Image-Edit-shaped engine (double-stream only, txt_norm, timestep-only conditioning, Qwen rotary
    table, sequential tagged CFG with different text lengths, norm-preserving CFG) vs the oracle, on the condition
    image of the reference fixture (a region by construction: with a random condition the similarities sit inside the
    numerical noise of the threshold and the mask comparison is meaningless)."""
    from regione_amd.harness import qwen as HQ
    cfg = synth.FluxConfig(**synth.QWEN_TOY)
    h = w = 16
    Tp, Tn = 32, 24
    wts = synth.make_flux_weights(cfg, seed=6, dtype=torch.bfloat16, w_std=0.05)
    lat, _, prompt, _ = synth.make_edit_inputs(h, w, Tp, cfg, seed=9, dtype=torch.bfloat16)
    _, _, nprompt, _ = synth.make_edit_inputs(h, w, Tn, cfg, seed=10, dtype=torch.bfloat16)
    img = golden("qwen_toy_bf16")["image_latents"]
    pipe = HQ.QwenImageEditPipeline(HQ.QwenImageTransformer2DModel(cfg, "cuda").load_state_dict(wts))
    helper = RegionEHelper(pipe)
    helper.set_params(threshold=0.5)
    helper.enable()
    out = pipe(image=img.cuda(), prompt_embeds=prompt.cuda(), negative_prompt_embeds=nprompt.cuda(), height=h * 16,
               width=w * 16, latents=lat.cuda(), true_cfg_scale=4.0, return_dict=False)[0].cpu()
    st = O.RegionState()
    st.set_parameters(28, 6, 2, "16", 0.5, 0.03, True)
    ocfg = O.FluxCfg(n_double=cfg.n_double, n_single=0, heads=cfg.heads, head_dim=cfg.head_dim, joint_dim=cfg.joint_dim)
    shapes = [(1, h, w), (1, h, w)]

    def mk(pe, T):
        caches = [O.KVCache() for _ in range(cfg.n_layers)]
        rope = O.qwen_rope(shapes, T)
        def model(x, t, ids):
            st.txt_length = T
            ts = t.expand(x.shape[0]).to(x.dtype)
            return O.transformer_forward(wts, ocfg, st, caches, x, pe, None, ts / 1000, ids, None, None, rope_full=rope)
        return model
    with torch.no_grad():
        ref = O.denoise(mk(prompt, Tp), st, lat, img, torch.arange(2 * h * w), Tp, h, w, family="qwen",
                        neg_model_fn=mk(nprompt, Tn), true_cfg_scale=4.0)
    assert torch.equal(pipe._regione_manager.edited_ids.cpu(), st.edited_ids)
    assert O.psnr(out, ref) >= 40.0
    helper.disable()
    van = pipe(image=img.cuda(), prompt_embeds=prompt.cuda(), negative_prompt_embeds=nprompt.cuda(), height=h * 16,
               width=w * 16, latents=lat.cuda(), true_cfg_scale=4.0, return_dict=False)[0]
    assert torch.isfinite(van.float()).all()


@pytest.mark.parametrize("name", ["qwen_loop_bf16_32", "qwen_loop_f32_16"])
def test_qwen_loop_fixture_on_gpu(golden, name):
    from tests.test_host_logic import qwen_case
    g = golden(name)
    pipe, out, trace = qwen_case(g, device="cuda")
    assert "".join(trace["kind"]) == "".join(g["kinds"].tolist())
    assert torch.equal(pipe._regione_manager.edited_ids.cpu().squeeze(0).int(), g["edited_ids"].squeeze(0))
    assert torch.equal(out.cpu(), g["final"])             # norm-preserving CFG on torch-CPU's row-norm trees: bit for bit
    for i in range(28):
        if f"lat{i}" in g:
            assert torch.equal(trace["latents"][i].cpu(), g[f"lat{i}"]), i


def test_fit_gamma_tool_and_non_28_step_run_on_gpu():
    """tools/fit_gamma.py on the toy engine at N = 20 -> a 19-entry table -> RegionE with num_inference_steps = 20
    (extension) runs, caches at least one step, and matches the oracle loop driven by the same table (ids exact,
    latents >= 40 dB)."""
    from tools.fit_gamma import fit
    cfg = synth.FluxConfig(**synth.TOY)
    h = w = 16
    T, N = 32, 20
    wts = synth.make_flux_weights(cfg, seed=5, dtype=torch.bfloat16, w_std=0.05)
    pipe = _toy_pipe({k: v.cuda() for k, v in wts.items()}, cfg)

    def call(seed):
        lat, img, prompt, pooled = [t.cuda() for t in synth.make_edit_inputs(h, w, T, cfg, seed=50 + seed)]
        pipe(image=img, prompt_embeds=prompt, pooled_prompt_embeds=pooled, height=h * 16, width=w * 16, latents=lat,
             num_inference_steps=N, return_dict=False)
    gamma, ratio = fit(pipe, call, N, samples=2)
    assert gamma.shape == (N - 1,) and gamma.dtype == torch.float16 and torch.isfinite(gamma.float()).all()
    lat, img, prompt, pooled = synth.make_edit_inputs(h, w, T, cfg, seed=9, dtype=torch.bfloat16)
    helper = RegionEHelper(pipe)
    kw = dict(warmup_step=4, post_step=2, refresh_step="10", threshold=0.1, cache_threshold=0.5)
    helper.set_params(num_inference_steps=N, gamma=gamma, **kw)
    helper.enable()
    trace = {}
    out = pipe(image=img.cuda(), prompt_embeds=prompt.cuda(), pooled_prompt_embeds=pooled.cuda(), height=h * 16, width=w * 16,
               latents=lat.cuda(), num_inference_steps=N, return_dict=False, trace=trace)[0].cpu()
    st = O.RegionState()
    st.set_parameters(N, 4, 2, "10", 0.1, 0.5, True, gamma=gamma)
    ocfg = O.FluxCfg(**synth.TOY)
    caches = [O.KVCache() for _ in range(cfg.n_layers)]
    txt_ids = torch.zeros(T, 3)

    def model(x, t, img_ids):
        ts = t.expand(x.shape[0]).to(x.dtype)
        guidance = torch.full([1], 2.5, dtype=torch.float32).expand(x.shape[0])
        return O.transformer_forward(wts, ocfg, st, caches, x, prompt, pooled, ts / 1000, img_ids, txt_ids, guidance)
    tr = {}
    with torch.no_grad():
        ref = O.denoise(model, st, lat, img, synth.flux_latent_ids(h, w), T, h, w, trace=tr)
    assert len(trace["kind"]) == N and "".join(trace["kind"]) == "".join(tr["kind"])
    assert torch.equal(pipe._regione_manager.edited_ids.cpu(), st.edited_ids)
    assert O.psnr(out, ref) >= 40.0


@pytest.mark.parametrize("name", ["toy_bf16_all", "toy_bf16_none"])
def test_toy_mmdit_partition_edge_cases_vs_reference_fixture(golden, name):
    """K_e = L (every token edited: the region path on the full token set) and K_e = 0 (no token edited: region steps carry
    the text rows only, zero-row image problems in every launch) on the HIP engine against the fixtures the REFERENCE's own
    __call__ produced for these thresholds: same plan, same lengths, ids exact, latents >= 40 dB."""
    g = golden(name)
    h, w, T = g["h"], g["w"], g["T"]
    cfg = synth.FluxConfig(**synth.TOY)
    wts = synth.make_flux_weights(cfg, seed=42, dtype=torch.bfloat16, w_std=g["w_std"])
    lat, _, prompt, pooled = synth.make_edit_inputs(h, w, T, cfg, seed=g["seed"], dtype=torch.bfloat16)
    img = golden("toy_bf16")["image_latents"]
    pipe = _toy_pipe(wts, cfg)
    helper = RegionEHelper(pipe)
    helper.set_params(threshold=g["threshold"])
    helper.enable()
    trace = {}
    out = pipe(image=img.cuda(), prompt_embeds=prompt.cuda(), pooled_prompt_embeds=pooled.cuda(), height=h * 16, width=w * 16,
               latents=lat.cuda(), guidance_scale=2.5, return_dict=False, trace=trace)[0].cpu()
    assert "".join(trace["kind"]) == "".join(g["kinds"].tolist())
    M = pipe._regione_manager
    assert M.edited_ids.shape[1] == (h * w if name.endswith("all") else 0)
    assert [x.shape[1] for x in trace["latents"]] == g["len"].tolist()
    for i in (5, 6, 15, 27):
        if trace["latents"][i].shape[1]:
            assert O.psnr(trace["latents"][i].cpu(), g[f"lat{i}"]) > 40.0, i
    assert torch.isfinite(out.float()).all() and O.psnr(out, g["final"]) >= 40.0


def test_edit_driver_timing_protocol(tmp_path):
    """tools/edit_driver.py: the reference drivers' protocol (jsonl items, 3 warm-ups, synchronised wall-clock per item,
    time_consuming.json with the reference's keys) on the toy engine."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    items = tmp_path / "data.jsonl"
    items.write_text("\n".join(json.dumps({"instruction": f"edit number {i}", "key": f"synthetic/item_{i}"}) for i in range(3)))
    out = tmp_path / "result"
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "edit_driver.py"), "--image_path", str(items), "--use_regione",
                        "--compare", "--toy", "--size", "256", "--threshold", "0.1", "--erosion_dilation", "--output_dir", str(out)],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    rep = json.load(open(out / "time_consuming.json"))
    assert rep["num_item"] == 3 and len(rep["time_consuming_list"]) == 3 and rep["ave_time_consuming"] > 0
    assert set(rep["latent_psnr_vs_full_token_db"]) == {f"synthetic/item_{i}" for i in range(3)}
    assert all(os.path.exists(out / f"item_{i}.latent.pt") for i in range(3))


@pytest.mark.parametrize("family", ["flux", "step1x_v1p2"])
def test_edit_driver_hosted_end_to_end_stage_timing(tmp_path, family):
    """tools/edit_driver.py --pipeline_factory: the reference drivers' END-TO-END protocol on a stock pipeline object
    (`RegionEHelper(pipe).enable(); pipe(image=, prompt=)`), reporting encode / loop / decode wall-clock per item and on
    average, RegionE and full-token through the same hosted call."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    items = tmp_path / "data.jsonl"
    items.write_text("\n".join(json.dumps({"instruction": f"edit number {i}", "key": f"synthetic/item_{i}"}) for i in range(2)))
    out = tmp_path / "result"
    env = dict(os.environ, PYTHONPATH=root + os.pathsep + os.path.join(root, "tests") + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "edit_driver.py"), "--image_path", str(items), "--use_regione",
                        "--compare", "--size", "256", "--threshold", "0.5", "--erosion_dilation", "--output_dir", str(out),
                        "--pipeline_factory", f"tests.host_standins:make_{family}"], capture_output=True, text=True, timeout=900, env=env,
                       cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    rep = json.load(open(out / "time_consuming.json"))
    assert rep["num_item"] == 2 and len(rep["stages_per_item"]) == 2
    for st in [rep["stages"], rep["full_token"]["stages"]] + rep["stages_per_item"]:
        assert set(st) == {"encode_s", "loop_s", "decode_s"} and st["loop_s"] > 0 and st["encode_s"] >= 0 and st["decode_s"] >= 0
    assert rep["ave_time_consuming"] >= rep["stages"]["loop_s"] and rep["speedup"] > 0 and rep["loop_speedup"] > 0


def test_qwen_toy_mmdit_vs_reference_fixture(golden):
    """The Qwen-Image-Edit patch set on the HIP engine against the fixture produced by the REFERENCE's own Qwen
    __call__ + transformer forward + two-cache tagged processors (tests/golden/qwen_toy_bf16.npz): plan and edited ids
    exact, velocities / latents / final output within the north-star tolerance."""
    from regione_amd.harness import qwen as HQ
    g = golden("qwen_toy_bf16")
    h, w, T, Tn = g["h"], g["w"], g["T"], g["Tn"]
    cfg = synth.FluxConfig(**synth.QWEN_TOY)
    wts = synth.make_flux_weights(cfg, seed=g["wseed"], dtype=torch.bfloat16, w_std=g["w_std"])
    assert float(sum(v.double().abs().sum() for v in wts.values())) == g["weight_abs_sum"], "torch RNG drift"
    lat, _, prompt, _ = synth.make_edit_inputs(h, w, T, cfg, seed=g["seed"], dtype=torch.bfloat16)
    _, _, nprompt, _ = synth.make_edit_inputs(h, w, Tn, cfg, seed=g["nseed"], dtype=torch.bfloat16)
    img = g["image_latents"]
    pipe = HQ.QwenImageEditPipeline(HQ.QwenImageTransformer2DModel(cfg, "cuda").load_state_dict(wts))
    helper = RegionEHelper(pipe)
    helper.set_params(threshold=g["threshold"], cache_threshold=g["cache_threshold"])
    helper.enable()
    trace = {}
    out = pipe(image=img.cuda(), prompt_embeds=prompt.cuda(), negative_prompt_embeds=nprompt.cuda(), height=h * 16,
               width=w * 16, latents=lat.cuda(), true_cfg_scale=g["true_cfg_scale"], return_dict=False, trace=trace)[0].cpu()
    assert "".join(trace["kind"]) == "".join(g["kinds"].tolist())
    assert torch.equal(pipe._regione_manager.edited_ids.cpu().squeeze(0).int(), g["edited_ids"].squeeze(0))
    for i in (0, 5, 6, 15, 27):
        assert O.psnr(trace["noise_pred"][i].cpu(), g[f"np{i}"]) > 35.0, i
        assert O.psnr(trace["latents"][i].cpu(), g[f"lat{i}"]) > 40.0, i
    assert O.psnr(out, g["final"]) >= 40.0


@pytest.mark.parametrize("name,v1p2", [("s1x_toy_bf16", False), ("s1xv2_toy_bf16", True)])
def test_step1x_toy_mmdit_vs_reference_fixture(golden, name, v1p2):
    """Step1X-Edit v1p1 (batched CFG) / v1p2 (tagged sequential CFG, text lengths 32 / 24) patch sets on the HIP engine
    against fixtures produced by the REFERENCE's own __call__ + transformer forward + attention processors: plan and
    edited ids exact, velocities / latents / final output within the north-star tolerance."""
    from regione_amd.harness import step1x as HS
    g = golden(name)
    h, w, T, Tn = g["h"], g["w"], g["T"], g["Tn"]
    cfg = synth.FluxConfig(guidance_embeds=False, **synth.TOY)
    wts = synth.make_flux_weights(cfg, seed=g["wseed"], dtype=torch.bfloat16, w_std=g["w_std"])
    assert float(sum(v.double().abs().sum() for v in wts.values())) == g["weight_abs_sum"], "torch RNG drift"
    lat, _, prompt, y = synth.make_edit_inputs(h, w, T, cfg, seed=g["seed"], dtype=torch.bfloat16)
    _, _, nprompt, ny = synth.make_edit_inputs(h, w, Tn, cfg, seed=g["nseed"], dtype=torch.bfloat16)
    img = g["image_latents"]
    tr_model = HS.Step1XEditTransformer2DModel(cfg, "cuda").load_state_dict(wts)
    pipe = HS.Step1XEditPipelineV1P2(tr_model) if v1p2 else HS.Step1XEditPipeline(tr_model)
    helper = RegionEHelper(pipe)
    helper.set_params(threshold=g["threshold"], cache_threshold=g["cache_threshold"])
    helper.enable()
    trace = {}
    out = pipe(image=img.cuda(), prompt_embeds=prompt.cuda(), pooled_prompt_embeds=y.cuda(), negative_prompt_embeds=nprompt.cuda(),
               negative_pooled_prompt_embeds=ny.cuda(), height=h * 16, width=w * 16, latents=lat.cuda(),
               true_cfg_scale=g["true_cfg_scale"], return_dict=False, trace=trace)[0].cpu()
    assert "".join(trace["kind"]) == "".join(g["kinds"].tolist())
    assert torch.equal(pipe._regione_manager.edited_ids.cpu().squeeze(0).int(), g["edited_ids"].squeeze(0))
    for i in (0, 5, 6, 15, 27):
        assert O.psnr(trace["noise_pred"][i].cpu(), g[f"np{i}"]) > 35.0, i
        assert O.psnr(trace["latents"][i].cpu(), g[f"lat{i}"]) > 40.0, i
    assert O.psnr(out, g["final"]) >= 40.0


def test_pipeline_reuse_no_state_leak_between_edits():
    """One engine, many edits: different images / region sizes / latent grids, RegionE toggled off and on in between.
    Re-running the first edit at the end must reproduce its first result BIT FOR BIT (no stale K/V cache rows, ids,
    modulation tables or workspace contents survive from the edits in between)."""
    cfg = synth.FluxConfig(**synth.TOY)
    wts = synth.make_flux_weights(cfg, seed=5, dtype=torch.bfloat16, w_std=0.05)
    pipe = _toy_pipe({k: v.cuda() for k, v in wts.items()}, cfg)
    helper = RegionEHelper(pipe)
    helper.set_params(threshold=0.3)

    def edit(h, w, seed, T=32):
        lat, img, prompt, pooled = [t.cuda() for t in synth.make_edit_inputs(h, w, T, cfg, seed=seed, dtype=torch.bfloat16)]
        out = pipe(image=img, prompt_embeds=prompt, pooled_prompt_embeds=pooled, height=h * 16, width=w * 16, latents=lat,
                   guidance_scale=2.5, return_dict=False)[0]
        M = getattr(pipe, "_regione_manager", None)
        return out.clone(), (None if M is None or M.edited_ids is None else M.edited_ids.clone())
    helper.enable()
    first, ids_first = edit(16, 16, 1)
    edit(16, 16, 2)                       # another image, another region
    edit(24, 16, 3, T=40)                 # another latent grid and text length (workspace / cache slabs grow)
    helper.disable()
    van1, _ = edit(16, 16, 1)             # full-token run of image 1
    helper.enable()
    edit(12, 12, 4)                       # smaller again
    again, ids_again = edit(16, 16, 1)
    assert torch.equal(ids_first, ids_again) and torch.equal(first, again)
    helper.disable()
    van2, _ = edit(16, 16, 1)
    assert torch.equal(van1, van2) and torch.isfinite(first.float()).all()


@pytest.mark.parametrize("h,w,T", [(13, 11, 19), (7, 30, 5), (33, 17, 77)])
def test_ragged_sizes_forward_and_regione_run(h, w, T):
    """Token counts that are multiples of nothing (GEMM / attention / cache-scatter tails): one full forward vs the
    oracle (>= 40 dB), then a whole RegionE edit with a constructed region: finite, the reference-logic step plan, the
    compacted length inside the region-aware stage."""
    cfg = synth.FluxConfig(**synth.TOY)
    wts = synth.make_flux_weights(cfg, seed=11, dtype=torch.bfloat16, w_std=0.05)
    lat, img, prompt, pooled = synth.make_edit_inputs(h, w, T, cfg, seed=21, dtype=torch.bfloat16)
    ids = synth.flux_latent_ids(h, w)
    L = h * w
    st = O.RegionState()
    st.set_parameters(28, 6, 2, "16", 0.5, 0.04, True)
    st.refresh(img, ids, T, h, w)
    _, ts = O.flow_match_schedule(28, L)
    x = torch.cat([lat, img], 1)
    tstep = ts[3].expand(1).to(torch.bfloat16) / 1000
    guidance = torch.full([1], 2.5)
    ocfg = O.FluxCfg(**synth.TOY)
    with torch.no_grad():
        ref = O.transformer_forward(wts, ocfg, st, [O.KVCache() for _ in range(cfg.n_layers)], x, prompt, pooled, tstep, ids,
                                    torch.zeros(T, 3), guidance)
    pipe = _toy_pipe(wts, cfg)
    out = pipe.transformer(hidden_states=x.cuda(), timestep=tstep, guidance=guidance, pooled_projections=pooled.cuda(),
                           encoder_hidden_states=prompt.cuda(), txt_ids=torch.zeros(T, 3), img_ids=ids, return_dict=False)[0].cpu()
    assert O.psnr(out, ref) > 40.0
    # whole edit with a region fixed by construction (velocity substitution at step warmup-1, as bench.py does)
    import bench as B
    box = (h // 4, h // 4 + max(h // 3, 3), w // 4, w // 4 + max(w // 3, 3))
    helper = RegionEHelper(pipe)
    helper.set_params(threshold=0.5)
    helper.enable()
    B.install_region_injection(pipe, h, w, box, img[0:1].cuda(), seed=7)
    trace = {}
    got = pipe(image=img.cuda(), prompt_embeds=prompt.cuda(), pooled_prompt_embeds=pooled.cuda(), height=h * 16, width=w * 16,
               latents=lat.cuda(), guidance_scale=2.5, return_dict=False, trace=trace)[0].cpu()
    assert torch.isfinite(got.float()).all() and got.shape == (1, L, 64)
    kinds = "".join(trace["kind"])
    assert kinds == "".join(O.derive_schedule(L, "flux", 6, 2, "16", 0.04)).replace("S", "F")
    K = pipe._regione_manager.edited_ids.numel()
    assert 0 < K < L and [x_.shape[1] for x_ in trace["latents"]][6] == K


@pytest.mark.parametrize("threshold,expect", [(-2.0, "none"), (2.0, "all")])
def test_degenerate_partitions_nothing_or_everything_edited(threshold, expect):
    """K_e = 0 (no token below the threshold) and K_e = L (every token) must run: region steps
    then work on the text rows only / on all noise tokens against the cached condition tokens."""
    cfg = synth.FluxConfig(**synth.TOY)
    h = w = 16
    T, L = 32, 256
    wts = synth.make_flux_weights(cfg, seed=5, dtype=torch.bfloat16, w_std=0.05)
    pipe = _toy_pipe({k: v.cuda() for k, v in wts.items()}, cfg)
    lat, img, prompt, pooled = [t.cuda() for t in synth.make_edit_inputs(h, w, T, cfg, seed=3, dtype=torch.bfloat16)]
    helper = RegionEHelper(pipe)
    helper.set_params(threshold=threshold)
    helper.enable()
    trace = {}
    out = pipe(image=img, prompt_embeds=prompt, pooled_prompt_embeds=pooled, height=h * 16, width=w * 16, latents=lat,
               guidance_scale=2.5, return_dict=False, trace=trace)[0]
    K = pipe._regione_manager.edited_ids.numel()
    assert K == (0 if expect == "none" else L)
    assert torch.isfinite(out.float()).all() and out.shape == (1, L, 64)
    assert "".join(trace["kind"]) == "".join(O.derive_schedule(L, "flux", 6, 2, "16", 0.04)).replace("S", "F")


@pytest.mark.parametrize("family", ["flux", "step1x_v1p2", "qwen", "step1x_v1p2_fp8"])
def test_last_block_skips_rows_nothing_reads_same_result(family, monkeypatch):
    """harness/flux.py `out_rows`: the pipelines read only `[:, :latents.size(1)]` of a forward, so in a full step the last
    single block computes queries / MLP / attention / proj_out (and norm_out / proj_out) for the latent rows only.  Same
    latents and ids as with every row computed like the reference does (harness.flux.SKIP_UNREAD_ROWS = False), RegionE on and off; a direct
    transformer call without the hint still returns every row."""
    from regione_amd.harness import step1x as HS
    h = w = 16
    if family == "flux":
        cfg = synth.FluxConfig(**synth.TOY)
        wts = synth.make_flux_weights(cfg, seed=5, dtype=torch.bfloat16, w_std=0.05)
        pipe = _toy_pipe({k: v.cuda() for k, v in wts.items()}, cfg)
    elif family == "qwen":
        from regione_amd.harness import qwen as HQ
        cfg = synth.FluxConfig(**synth.QWEN_TOY)
        wts = synth.make_flux_weights(cfg, seed=6, dtype=torch.bfloat16, w_std=0.05)
        pipe = HQ.QwenImageEditPipeline(HQ.QwenImageTransformer2DModel(cfg, "cuda").load_state_dict(wts))
    else:
        cfg = synth.FluxConfig(guidance_embeds=False, **synth.TOY)
        wts = synth.make_flux_weights(cfg, seed=5, dtype=torch.bfloat16, w_std=0.05)
        pipe = HS.Step1XEditPipelineV1P2(HS.Step1XEditTransformer2DModel(cfg, "cuda").load_state_dict(wts))
        if family.endswith("fp8"):           # quantised trunk: the weight-row slices of the skipping path carry their scales (ops.wrows)
            pipe.transformer.quantize_fp8_()
    cu = lambda t: t.cuda() if t is not None else None
    lat, img, prompt, y = [cu(t) for t in synth.make_edit_inputs(h, w, 32, cfg, seed=9, dtype=torch.bfloat16)]
    _, _, nprompt, ny = [cu(t) for t in synth.make_edit_inputs(h, w, 24, cfg, seed=10, dtype=torch.bfloat16)]
    blk = torch.zeros(h, w, dtype=torch.bool)
    blk[4:10, 5:12] = True                                   # a region by construction: condition == start latents outside it
    img = lat.clone()
    img[0, blk.flatten().cuda()] = -lat[0, blk.flatten().cuda()]
    kw = dict(image=img, prompt_embeds=prompt, height=h * 16, width=w * 16, latents=lat, return_dict=False)
    if family == "flux":
        kw.update(pooled_prompt_embeds=y, guidance_scale=2.5)
    elif family == "qwen":
        kw.update(negative_prompt_embeds=nprompt, true_cfg_scale=4.0)
    else:
        kw.update(pooled_prompt_embeds=y, negative_prompt_embeds=nprompt, negative_pooled_prompt_embeds=ny, true_cfg_scale=4.0)
    helper = RegionEHelper(pipe)
    helper.set_params(threshold=0.5)
    res = {}
    for skip in (True, False):
        monkeypatch.setattr(H, "SKIP_UNREAD_ROWS", skip)
        van = pipe(**kw)[0].clone()
        helper.enable()
        reg = pipe(**kw)[0].clone()
        ids = pipe._regione_manager.edited_ids.clone()
        helper.disable()
        res[skip] = (van, reg, ids)
    assert torch.equal(res[True][2], res[False][2])
    assert torch.equal(res[True][0], res[False][0]) and torch.equal(res[True][1], res[False][1])
    assert 0 < res[True][2].numel() < h * w
    if family == "flux":
        monkeypatch.setattr(H, "SKIP_UNREAD_ROWS", True)
        tr = pipe.transformer
        x = torch.cat([lat, img], 1)
        a = dict(hidden_states=x, timestep=torch.tensor([0.5], dtype=torch.bfloat16), guidance=torch.tensor([2.5]),
                 pooled_projections=y, encoder_hidden_states=prompt, txt_ids=torch.zeros(32, 3), img_ids=synth.flux_latent_ids(h, w),
                 return_dict=False)
        full = tr(**a)[0]
        tr.out_rows_hint = h * w
        part = tr(**a)[0]
        assert full.shape[1] == 2 * h * w and part.shape[1] == h * w and torch.equal(full[:, : h * w], part)
        assert tr(**a)[0].shape[1] == 2 * h * w          # the hint is one-shot


def test_gpu_eager_scalars_switch_rounds_the_decay_ratio_like_torch_device_kernels():
    """`cache * ratio` with a 0-dim fp32 DEVICE tensor (what the reference computes when it runs on a GPU, inplace.py:318):
    torch casts the ratio to the bf16 of the cache first.  `set_params(gpu_eager_scalars=True)` follows that; the default
    keeps the fp32 scalar like torch's CPU kernels (what the CPU-generated fixtures pin).  Checked on every cache-served
    step of a toy edit against torch's own device arithmetic."""
    from regione_amd import RegionEHelper, synth, ops
    from regione_amd.FluxKontext.inplace import gamma
    cfg = synth.FluxConfig(**synth.TOY)
    wts = synth.make_flux_weights(cfg, seed=42, dtype=torch.bfloat16, w_std=0.05)
    lat, img, prompt, pooled = synth.make_edit_inputs(16, 16, 24, cfg, seed=5, dtype=torch.bfloat16)
    pipe = _toy_pipe(wts, cfg)
    helper = RegionEHelper(pipe)
    kw = dict(image=img.cuda(), prompt_embeds=prompt.cuda(), pooled_prompt_embeds=pooled.cuda(), height=256, width=256,
              latents=lat.cuda(), guidance_scale=2.5, return_dict=False)
    differ = 0
    for flag in (False, True):
        helper.set_params(threshold=0.1, gpu_eager_scalars=flag)
        helper.enable()
        trace = {}
        pipe(trace=trace, **kw)
        M = pipe._regione_manager
        ts = pipe.scheduler.timesteps.float().cpu()
        kinds, last, checked = trace["kind"], None, 0
        for i, k in enumerate(kinds):
            v = trace["noise_pred"][i]
            if k == "C":
                ratio = gamma[i - 1] * (1 + (ts[i] - ts[i - 1]) / 1000)          # fp16 table x fp32 -> fp32, as in the reference
                src = last if last.shape[1] == v.shape[1] else ops.gather_rows(last, M.edited_ids)
                dev = src * ratio.to(torch.float32).cuda()                          # torch device kernel: 0-dim fp32 -> bf16 first
                host = (src.float() * float(ratio)).to(torch.bfloat16)              # fp32 scalar kept (CPU kernel semantics)
                assert torch.equal(v, dev if flag else host), (flag, i)
                differ += int(not torch.equal(dev, host))
                checked += 1
            last = v if k != "C" else (last if last.shape[1] == v.shape[1] else ops.gather_rows(last, M.edited_ids))
        assert checked == kinds.count("C") > 0
        helper.disable()
    assert differ > 0                                                               # the two conventions really do differ


@pytest.mark.parametrize("mode", ["1", "2"])
def test_cfg_branches_on_two_streams_bit_identical_to_sequential(golden, mode, monkeypatch):
    """The uncond forward of a two-forward CFG step may run on a side stream next to the cond forward (region steps by
    default, RGN_BRANCH_STREAMS=2: every computed step): same launches, same arguments, one activation workspace and one
    split-K / KV-split scratch per stream -> the whole 28-step edit is bit-identical to the sequential run (mode 0), trace
    included, for the tagged-CFG families and for FLUX true CFG."""
    from regione_amd.harness import qwen as HQ
    monkeypatch.setenv("RGN_BATCH_BRANCHES", "0")            # two forwards per step (the batched pass, round 3, has its own test)
    cfg = synth.FluxConfig(**synth.QWEN_TOY)
    h = w = 16
    wts = synth.make_flux_weights(cfg, seed=6, dtype=torch.bfloat16, w_std=0.05)
    lat, _, prompt, _ = synth.make_edit_inputs(h, w, 32, cfg, seed=9, dtype=torch.bfloat16)
    _, _, nprompt, _ = synth.make_edit_inputs(h, w, 24, cfg, seed=10, dtype=torch.bfloat16)
    img = golden("qwen_toy_bf16")["image_latents"]
    pipe = HQ.QwenImageEditPipeline(HQ.QwenImageTransformer2DModel(cfg, "cuda").load_state_dict(wts))
    helper = RegionEHelper(pipe)
    helper.set_params(threshold=0.5)
    helper.enable()
    kw = dict(image=img.cuda(), prompt_embeds=prompt.cuda(), negative_prompt_embeds=nprompt.cuda(), height=h * 16, width=w * 16,
              latents=lat.cuda(), true_cfg_scale=4.0, return_dict=False)
    runs = {}
    for m in ("0", mode, "0", mode):
        monkeypatch.setenv("RGN_BRANCH_STREAMS", m)
        trace = {}
        out = pipe(trace=trace, **kw)[0]
        torch.cuda.synchronize()
        runs.setdefault(m, []).append((out.clone(), [x.clone() for x in trace["noise_pred"]], "".join(trace["kind"])))
    a, b = runs["0"][0], runs[mode][0]
    assert a[2] == b[2] and "R" in a[2]
    assert torch.equal(a[0], b[0]) and all(torch.equal(x, y) for x, y in zip(a[1], b[1]))
    assert torch.equal(runs[mode][0][0], runs[mode][1][0]) and torch.equal(runs["0"][0][0], runs["0"][1][0])
    # FLUX true CFG (per-branch caches): same property
    fcfg = synth.FluxConfig(**synth.TOY)
    fw = synth.make_flux_weights(fcfg, seed=42, dtype=torch.bfloat16, w_std=0.05)
    flat, fimg, fp, fpool = synth.make_edit_inputs(16, 16, 24, fcfg, seed=5, dtype=torch.bfloat16)
    _, _, fn, fnpool = synth.make_edit_inputs(16, 16, 24, fcfg, seed=6, dtype=torch.bfloat16)
    fpipe = _toy_pipe(fw, fcfg)
    fh = RegionEHelper(fpipe)
    fh.set_params(threshold=0.1)
    fh.enable()
    fkw = dict(image=fimg.cuda(), prompt_embeds=fp.cuda(), pooled_prompt_embeds=fpool.cuda(), negative_prompt_embeds=fn.cuda(),
               negative_pooled_prompt_embeds=fnpool.cuda(), true_cfg_scale=6.0, height=256, width=256, latents=flat.cuda(),
               guidance_scale=2.5, return_dict=False)
    outs = []
    for m in ("0", mode):
        monkeypatch.setenv("RGN_BRANCH_STREAMS", m)
        outs.append(fpipe(**fkw)[0].clone())
        torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[1])



@pytest.mark.parametrize("family", ["flux_true_cfg", "step1x", "step1x_v1p2", "qwen"])
def test_cfg_branches_batched_through_one_pass_bit_identical_to_two_forwards(family, golden, monkeypatch):
    """Round 3: the cond / uncond forwards of a computed step run as ONE batched pass (the reference's B = 2 forward,
    Step1XEdit/inplace.py:381-399; rows of different branches never meet in a Linear): activations [text_0 ; image_0 ;
    text_1 ; image_1], one launch per projection over both branches (rgn_gemm_group, per-branch Q/K/V epilogue descriptors,
    K / V^T caches, rotary tables, AdaLN vectors), attention per branch.  At these dimensions no launch takes a split-K
    path, so every row sees the same arithmetic as in a forward of its own: the whole 28-step edit - every noise_pred of the
    trace, the ids, the final latents - is BIT-IDENTICAL to the two-forward run (RGN_BATCH_BRANCHES=0), RegionE on and off."""
    from regione_amd.harness import qwen as HQ, step1x as HS
    h = w = 16
    cu = lambda t: t.cuda() if t is not None else None
    if family == "qwen":
        cfg = synth.FluxConfig(**synth.QWEN_TOY)
        wts = synth.make_flux_weights(cfg, seed=6, dtype=torch.bfloat16, w_std=0.05)
        pipe = HQ.QwenImageEditPipeline(HQ.QwenImageTransformer2DModel(cfg, "cuda").load_state_dict(wts))
        img = golden("qwen_toy_bf16")["image_latents"]
    elif family == "flux_true_cfg":
        cfg = synth.FluxConfig(**synth.TOY)
        wts = synth.make_flux_weights(cfg, seed=42, dtype=torch.bfloat16, w_std=0.05)
        pipe = _toy_pipe(wts, cfg)
        img = golden("toy_bf16")["image_latents"]
    else:
        cfg = synth.FluxConfig(guidance_embeds=False, **synth.TOY)
        wts = synth.make_flux_weights(cfg, seed=5, dtype=torch.bfloat16, w_std=0.05)
        cls = HS.Step1XEditPipelineV1P2 if family.endswith("v1p2") else HS.Step1XEditPipeline
        pipe = cls(HS.Step1XEditTransformer2DModel(cfg, "cuda").load_state_dict(wts))
        img = golden("s1xv2_toy_bf16" if family.endswith("v1p2") else "s1x_toy_bf16")["image_latents"]
    Tn = 32 if family in ("step1x", "flux_true_cfg") else 24          # v1p1's batch of two and FLUX share the text length
    lat, _, prompt, y = [cu(t) for t in synth.make_edit_inputs(h, w, 32, cfg, seed=9 if family != "flux_true_cfg" else 42, dtype=torch.bfloat16)]
    _, _, nprompt, ny = [cu(t) for t in synth.make_edit_inputs(h, w, Tn, cfg, seed=10, dtype=torch.bfloat16)]
    kw = dict(image=cu(img), prompt_embeds=prompt, negative_prompt_embeds=nprompt, height=h * 16, width=w * 16, latents=lat,
              true_cfg_scale=4.0, return_dict=False)
    if family != "qwen":
        kw.update(pooled_prompt_embeds=y, negative_pooled_prompt_embeds=ny)
    if family == "flux_true_cfg":
        kw.update(guidance_scale=2.5)
    helper = RegionEHelper(pipe)
    helper.set_params(threshold=0.5)
    monkeypatch.setenv("RGN_BRANCH_STREAMS", "0")
    res = {}
    # "1": the batched pass (second branch's attention on a side stream, the default), "1s": the batched pass on one stream,
    # "0": two forwards
    for batched in ("1", "1s", "0"):
        monkeypatch.setenv("RGN_BATCH_BRANCHES", batched[0])
        monkeypatch.setattr(H, "ATTN_BRANCH_STREAMS", batched != "1s")
        van = pipe(**kw)[0].clone()
        helper.enable()
        trace = {}
        reg = pipe(trace=trace, **kw)[0].clone()
        torch.cuda.synchronize()
        res[batched] = (van, reg, pipe._regione_manager.edited_ids.clone(), [x.clone() for x in trace["noise_pred"]], "".join(trace["kind"]))
        helper.disable()
    a, b, c = res["1"], res["0"], res["1s"]
    assert torch.equal(a[1], c[1]) and torch.equal(a[0], c[0]) and all(torch.equal(x, y) for x, y in zip(a[3], c[3]))
    assert a[4] == b[4] and "R" in a[4] and "F" in a[4]
    assert torch.equal(a[2], b[2]) and a[2].numel() > 0
    assert all(torch.equal(x, y) for x, y in zip(a[3], b[3])), "a noise_pred of the batched pass differs from the two-forward run"
    assert torch.equal(a[1], b[1]) and torch.equal(a[0], b[0])
    assert torch.isfinite(a[1].float()).all()


def test_attention_score_bound_is_loaded_eagerly_and_follows_in_place_weight_changes():
    """Advisor finding (round 3): `Attention.score_bound()` - the caller-side guarantee the bounded softmax runs on - is computed for
    every block when the weights are loaded (one device read for the trunk, none inside a forward) and is keyed on the norm
    weights' (data_ptr, _version): an in-place change after first use (LoRA merge, .copy_) is seen, not silently violated."""
    import math
    cfg = synth.FluxConfig(**synth.TOY)
    wts = synth.make_flux_weights(cfg, seed=42, dtype=torch.bfloat16, w_std=0.05)
    tr = H.FluxTransformer2DModel(cfg, "cuda").load_state_dict(wts)
    for blk in list(tr.transformer_blocks) + list(tr.single_transformer_blocks):
        assert blk.attn.__dict__.get("_score_bound") is not None          # eager: nothing left for the first forward
    a = tr.transformer_blocks[0].attn
    b0 = a.score_bound()
    wq = max(float(a.norm_q.float().abs().max()), float(a.norm_added_q.float().abs().max()))
    wk = max(float(a.norm_k.float().abs().max()), float(a.norm_added_k.float().abs().max()))
    assert b0 == pytest.approx(1.05 * a.head_dim * wq * wk / math.sqrt(a.head_dim), rel=1e-6)
    assert a.score_bound() == b0
    a.norm_k.mul_(3.0)                                                     # in place: same storage, new version
    assert a.score_bound() == pytest.approx(3.0 * b0, rel=2e-2) or a.score_bound() > 1.5 * b0
    s = tr.single_transformer_blocks[0].attn
    s.norm_q = (s.norm_q * 2).contiguous()                                 # swapped tensor: new data_ptr
    assert s.score_bound() > 1.5 * 1.05 * s.head_dim * 1.0 / math.sqrt(s.head_dim) * 0.5
