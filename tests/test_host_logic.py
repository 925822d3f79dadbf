"""CPU: the product's HOST logic (RegionEHelper, warp/unwarp, denoise loop control flow, AVD decision,
scheduler branch selection, manager state machine) against the reference-generated fixtures.

The device kernels are replaced *inside this test only* by oracle stand-ins (monkeypatched
regione_amd.ops.*): this exercises the Python host side without a GPU; it is not a product fallback."""
import numpy as np
import pytest
import torch

from oracle import regione_oracle as O
from regione_amd import RegionEHelper, ops, synth
from regione_amd.FluxKontext import inplace as fk
from regione_amd.harness import flux as H


@pytest.fixture()
def cpu_ops(monkeypatch):
    def arp(sample, mo, cond, dt_final, thr, h, w, ed=True, want_sim=False):
        est = sample.to(torch.float32)
        if mo is not None:
            est = est + torch.tensor(dt_final) * mo
        e, u, raw, final = O.token_selector(est, cond, thr, h, w, ed)
        return e, u, torch.from_numpy(final.copy()), torch.from_numpy(raw.copy()), None

    def euler(sample, v, dt, mask=None, dt_direct=0.0):
        s = sample.to(torch.float32)
        a = (s + torch.tensor(dt) * v)
        if mask is not None:
            b = (s + torch.tensor(dt_direct) * v)
            a = torch.where(mask.bool()[None, :, None], a, b)
        return a.to(v.dtype)

    def avd(cache, ratio, ids=None, round_ratio=False):
        c = O.ids_gather(cache, ids) if ids is not None else cache
        return c * torch.tensor(ratio)

    def cfg(pos, neg, scale, mode=0, power=0.4):
        fam = {0: "flux", 1: "step1x", 2: "qwen"}[mode]
        return O.cfg_combine(fam, pos, neg, scale, t=torch.tensor(1e9), power=power)

    # the product reaches its kernels through torch.ops.regione_mi (regione_amd.torch_ops.R): stand in for that namespace
    import types
    from regione_amd import torch_ops
    monkeypatch.setattr(torch_ops, "R", types.SimpleNamespace(
        cfg_combine=cfg, arp_partition=lambda *a: arp(*a)[:3], split_euler_step=euler, avd_apply=avd,
        gather_rows=lambda x, ids: O.ids_gather(x, ids) if x.dim() == 3 else x[ids.reshape(-1)],
        scatter_rows_=lambda s, ids, d: O.ids_scatter(s, ids, d)))
    monkeypatch.setattr(fk, "ids_gather", lambda x, ids, *a, **k: O.ids_gather(x, ids))
    # `selection` of inplace.py:732-733 (rgn_sel_rows on the device)
    monkeypatch.setattr(ops, "sel_rows", lambda ids, T: torch.cat((torch.arange(T), ids.reshape(-1) + T)))


class FakeTransformer:
    """Elementwise stand-in (same function the fixture generator gave the reference loop)."""

    def __init__(self, target_full, w_tok, L, device="cpu"):
        self.cfg_model = synth.FluxConfig()
        self.device = torch.device(device)
        self.transformer_blocks, self.single_transformer_blocks = [], []
        self.target, self.w_tok, self.L = target_full.to(device), w_tok, L

    def __call__(self, hidden_states=None, timestep=None, img_ids=None, **kw):
        tok = (img_ids[:, 0] * self.L + img_ids[:, 1] * self.w_tok + img_ids[:, 2]).long().to(self.device)
        n = hidden_states.shape[1]
        k = float(1.0 / timestep.float()[0].item())
        return (((hidden_states.float() - self.target[tok[:n]][None]) * k).to(hidden_states.dtype),)


def test_helper_surface_defaults_and_asserts(capsys):
    pipe = H.FluxKontextPipeline(FakeTransformer(torch.zeros(2, 64), 1, 1))
    h = RegionEHelper(pipe)
    assert h.name == "FluxKontextPipeline"
    assert h.config == {"num_inference_steps": 28, "warmup_step": 6, "post_step": 2, "refresh_step": "16",
                        "threshold": 0.93, "cache_threshold": 0.04, "erosion_dilation": True}
    h.set_params(threshold=0.88, refresh_step="12,20")
    assert "set_params" in capsys.readouterr().out
    assert h.config["threshold"] == 0.88 and h.config["warmup_step"] == 6
    with pytest.raises(AssertionError):
        h.set_params(num_inference_steps=50)
    h.enable()
    assert pipe.__class__.__name__ == "RegionEFluxKontextPipeline" and h.pipeline is pipe
    assert isinstance(pipe.scheduler, fk.RegionEFlowMatchEulerDiscreteScheduler)
    assert pipe._regione_manager.refresh_step == [12, 20, 27]
    h.disable()
    assert pipe.__class__ is H.FluxKontextPipeline and type(pipe.scheduler) is H.FlowMatchEulerDiscreteScheduler
    # parameter validation of Manager.set_parameters (utils.py:390-402)
    for bad in (dict(refresh_step="7"), dict(refresh_step="16,17"), dict(refresh_step="26"), dict(warmup_step=0)):
        hb = RegionEHelper(H.FluxKontextPipeline(FakeTransformer(torch.zeros(2, 64), 1, 1)))
        hb.set_params(**bad)
        with pytest.raises(AssertionError):
            hb.enable()
    with pytest.raises(KeyError):
        RegionEHelper(object())
    with pytest.raises(ValueError):
        H.FlowMatchEulerDiscreteScheduler().step(None, 3, None)


@pytest.mark.parametrize("name", ["loop_bf16_32", "loop_f32_16", "loop_bf16_50x83"])
def test_product_loop_matches_reference_trace(golden, cpu_ops, name):
    g = golden(name)
    h, w = g["h"], g["w"]
    dt = torch.bfloat16 if g["bf16"] else torch.float32
    L = h * w
    lat, img, _, _ = synth.make_edit_inputs(h, w, 8, synth.FluxConfig(), seed=g["seed"], dtype=dt)
    tgt = synth.region_target(h, w, tuple(int(x) for x in g["box"]), img, seed=g["tseed"], ramp=g["ramp"])
    tr = FakeTransformer(torch.cat([tgt, img[0].float()], 0), w, L)
    pipe = H.FluxKontextPipeline(tr)
    helper = RegionEHelper(pipe)
    helper.set_params(threshold=g["threshold"], cache_threshold=g["cache_threshold"], refresh_step=str(g["refresh_step"]))
    helper.enable()
    trace = {}
    out = pipe(image=img, prompt_embeds=torch.zeros(1, 8, 4), pooled_prompt_embeds=torch.zeros(1, 4), height=h * 16,
               width=w * 16, latents=lat, return_dict=False, trace=trace)[0]
    assert "".join(trace["kind"]) == "".join(g["kinds"].tolist())
    assert [x.shape[1] for x in trace["latents"]] == g["len"].tolist()
    assert [(-1 if p is None else p) for p in trace["prev_refresh"]] == g["prev_refresh"].tolist()
    M = pipe._regione_manager
    assert torch.equal(M.edited_ids.squeeze(0).int(), g["edited_ids"].squeeze(0))
    assert np.array_equal(np.array([float(x.double().sum()) for x in trace["latents"]]), g["lat_sum"].numpy())
    assert torch.equal(out, g["final"])
    # vanilla loop still works after disable()
    helper.disable()
    out2 = pipe(image=img, prompt_embeds=torch.zeros(1, 8, 4), pooled_prompt_embeds=torch.zeros(1, 4), height=h * 16,
                width=w * 16, latents=lat, return_dict=False)[0]
    st = O.RegionState()
    ref = O.denoise(lambda x, t, ids: tr(hidden_states=x, timestep=(t.expand(1).to(x.dtype) / 1000), img_ids=ids)[0],
                    st, lat, img, synth.flux_latent_ids(h, w), 8, h, w, regione=False)
    assert torch.equal(out2, ref)


def test_avd_decision_is_data_independent_plan(golden):
    """The host-side decision reproduces the oracle's derived plan for all three sequence lengths."""
    for L, thr in ((1024, 0.02), (4096, 0.04), (16384, 0.02)):
        plan = O.derive_schedule(L, "flux", 6, 2, "16", thr)
        M = fk.FluxKontextManager()
        M.set_parameters(dict(num_inference_steps=28, warmup_step=6, post_step=2, refresh_step="16", threshold=0.9,
                              cache_threshold=thr, erosion_dilation=True))
        M.refresh(torch.zeros(1, L, 1), torch.zeros(1, L, 1), torch.zeros(2 * L, 3), torch.zeros(4, 3), 2, 8, 16, 16)
        _, ts = O.flow_match_schedule(28, L)
        avd, got = fk.AvdState(), []
        for i in range(28):
            hit, _ = fk.avd_decide(M, avd, i, ts)
            got.append("C" if hit else ("F" if M.is_full_input_step() else "R"))
            cur = M.current_step
            if cur == M.warmup_step - 1:
                M.prev_refresh_step = M.refresh_step_real_time.pop(0) - 1
            elif M.prev_refresh_step is not None and cur == M.prev_refresh_step and M.refresh_step_real_time:
                M.next_refresh_step = M.refresh_step_real_time.pop(0) - 1
            M.current_step += 1
            c = M.current_step
            if c == M.inference_step - M.post_step:
                M.prev_refresh_step = None
            elif M.prev_refresh_step is not None and c == M.prev_refresh_step + 1 and c != M.warmup_step:
                M.prev_refresh_step = M.next_refresh_step
        assert "".join(got) == "".join(plan).replace("S", "F")
    assert "".join(O.derive_schedule(4096)).count("C") == 14      # 9 F, 5 R, 14 C


class FakeTransformerB2(FakeTransformer):
    """Batch-2 stand-in for Step1X's batched CFG forward (row 0 cond, row 1 uncond)."""

    def __init__(self, tpos_full, tneg_full, w_tok, L, device="cpu"):
        super().__init__(tpos_full, w_tok, L, device)
        self.cfg_model = synth.FluxConfig(guidance_embeds=False)
        self.t2 = (tpos_full.to(device), tneg_full.to(device))

    def __call__(self, hidden_states=None, timestep=None, img_ids=None, **kw):
        tok = (img_ids[:, 0] * self.L + img_ids[:, 1] * self.w_tok + img_ids[:, 2]).long().to(self.device)
        n = hidden_states.shape[1]
        k = float(1.0 / timestep.float()[0].item())
        outs = [((hidden_states[b:b + 1].float() - self.t2[b][tok[:n]][None]) * k).to(hidden_states.dtype)
                for b in range(hidden_states.shape[0])]
        return (torch.cat(outs, 0),)


def step1x_case(g, device="cpu"):
    from regione_amd.harness import step1x as HS
    h, w = g["h"], g["w"]
    dt = torch.bfloat16 if g["bf16"] else torch.float32
    L = h * w
    lat, img, _, _ = synth.make_edit_inputs(h, w, 8, synth.FluxConfig(), seed=g["seed"], dtype=dt)
    tpos = synth.region_target(h, w, tuple(int(x) for x in g["box"]), img, seed=g["tseed"], ramp=g["ramp"])
    tneg = tpos + 0.05 * torch.randn(tpos.shape, generator=torch.Generator().manual_seed(g["nseed"]))
    cond = img[0].float()
    tr = FakeTransformerB2(torch.cat([tpos, cond], 0), torch.cat([tneg, cond], 0), w, L, device)
    pipe = HS.Step1XEditPipeline(tr)
    helper = RegionEHelper(pipe)
    assert helper.config["threshold"] == 0.88 and helper.config["cache_threshold"] == 0.02      # tool/RegionE.py:3
    helper.enable()
    trace = {}
    dummy = torch.zeros(1, 8, 4).to(dt)
    out = pipe(image=img, prompt_embeds=dummy, negative_prompt_embeds=dummy, height=h * 16, width=w * 16, latents=lat,
               true_cfg_scale=g["true_cfg_scale"], return_dict=False, trace=trace)[0]
    return pipe, out, trace


@pytest.mark.parametrize("name", ["s1x_loop_bf16_32", "s1x_loop_f32_16"])
def test_step1x_product_loop_matches_reference_trace(golden, cpu_ops, name):
    g = golden(name)
    pipe, out, trace = step1x_case(g)
    assert pipe.__class__.__name__ == "RegionEStep1XEditPipeline"
    assert "".join(trace["kind"]) == "".join(g["kinds"].tolist())
    assert torch.equal(pipe._regione_manager.edited_ids.squeeze(0).int(), g["edited_ids"].squeeze(0))
    assert [x.shape[1] for x in trace["latents"]] == g["len"].tolist()
    assert np.array_equal(np.array([float(x.double().sum()) for x in trace["latents"]]), g["lat_sum"].numpy())
    assert torch.equal(out, g["final"])


class FakeTransformerTagged(FakeTransformerB2):
    """Sequential-CFG stand-in (Step1X-v1p2): branch = joint_attention_kwargs['tag']."""

    def __call__(self, hidden_states=None, timestep=None, img_ids=None, joint_attention_kwargs=None, **kw):
        tgt = self.t2[0 if joint_attention_kwargs["tag"] == "cond" else 1]
        tok = (img_ids[:, 0] * self.L + img_ids[:, 1] * self.w_tok + img_ids[:, 2]).long().to(self.device)
        n = hidden_states.shape[1]
        k = float(1.0 / timestep.float()[0].item())
        return (((hidden_states.float() - tgt[tok[:n]][None]) * k).to(hidden_states.dtype),)


def step1x_v1p2_case(g, device="cpu"):
    from regione_amd.harness import step1x as HS
    h, w = g["h"], g["w"]
    dt = torch.bfloat16 if g["bf16"] else torch.float32
    L = h * w
    lat, img, _, _ = synth.make_edit_inputs(h, w, 8, synth.FluxConfig(), seed=g["seed"], dtype=dt)
    tpos = synth.region_target(h, w, tuple(int(x) for x in g["box"]), img, seed=g["tseed"], ramp=g["ramp"])
    tneg = tpos + 0.05 * torch.randn(tpos.shape, generator=torch.Generator().manual_seed(g["nseed"]))
    cond = img[0].float()
    tr = FakeTransformerTagged(torch.cat([tpos, cond], 0), torch.cat([tneg, cond], 0), w, L, device)
    pipe = HS.Step1XEditPipelineV1P2(tr)
    helper = RegionEHelper(pipe)
    assert helper.name == "Step1XEditPipelineV1P2" and helper.config["threshold"] == 0.88
    helper.enable()
    trace = {}
    out = pipe(image=img, prompt_embeds=torch.zeros(1, g["txt_len"], 4).to(dt),
               negative_prompt_embeds=torch.zeros(1, g["neg_txt_len"], 4).to(dt), height=h * 16, width=w * 16,
               latents=lat, true_cfg_scale=g["true_cfg_scale"], return_dict=False, trace=trace)[0]
    return pipe, out, trace


def test_step1x_v1p2_product_loop_matches_reference_trace(golden, cpu_ops):
    g = golden("s1xv2_loop_bf16_32")
    pipe, out, trace = step1x_v1p2_case(g)
    assert "".join(trace["kind"]) == "".join(g["kinds"].tolist())
    M = pipe._regione_manager
    assert M.txt_length == 8 and M.neg_txt_length == 5
    assert torch.equal(M.edited_ids.squeeze(0).int(), g["edited_ids"].squeeze(0))
    assert np.array_equal(np.array([float(x.double().sum()) for x in trace["latents"]]), g["lat_sum"].numpy())
    assert torch.equal(out, g["final"])


class FakeTransformerQwen(FakeTransformerB2):
    """Qwen stand-in: 1-D latent ids, branch = attention_kwargs['tag']."""

    def __init__(self, tpos_full, tneg_full, device="cpu"):
        super().__init__(tpos_full, tneg_full, 1, 1, device)
        self.cfg_model = synth.FluxConfig(**synth.QWEN_TOY)

    def __call__(self, hidden_states=None, timestep=None, latent_ids=None, attention_kwargs=None, **kw):
        tgt = self.t2[0 if attention_kwargs["tag"] == "cond" else 1]
        n = hidden_states.shape[1]
        k = float(1.0 / timestep.float()[0].item())
        return (((hidden_states.float() - tgt[latent_ids[:n].long().to(self.device)][None]) * k).to(hidden_states.dtype),)


def qwen_case(g, device="cpu", plus=False):
    from regione_amd.harness import qwen as HQ
    h, w = g["h"], g["w"]
    dt = torch.bfloat16 if g["bf16"] else torch.float32
    lat, img, _, _ = synth.make_edit_inputs(h, w, 8, synth.FluxConfig(), seed=g["seed"], dtype=dt)
    tpos = synth.region_target(h, w, tuple(int(x) for x in g["box"]), img, seed=g["tseed"], ramp=g["ramp"])
    tneg = tpos + 0.05 * torch.randn(tpos.shape, generator=torch.Generator().manual_seed(g["nseed"]))
    cond = img[0].float()
    tr = FakeTransformerQwen(torch.cat([tpos, cond], 0), torch.cat([tneg, cond], 0), device)
    pipe = (HQ.QwenImageEditPlusPipeline if plus else HQ.QwenImageEditPipeline)(tr)
    helper = RegionEHelper(pipe)
    assert helper.config["threshold"] == 0.80 and helper.config["cache_threshold"] == 0.03      # tool/RegionE.py:5-6
    helper.enable()
    trace = {}
    out = pipe(image=img, prompt_embeds=torch.zeros(1, g["txt_len"], 4).to(dt),
               negative_prompt_embeds=torch.zeros(1, g["neg_txt_len"], 4).to(dt), height=h * 16, width=w * 16,
               latents=lat, true_cfg_scale=g["true_cfg_scale"], return_dict=False, trace=trace)[0]
    return pipe, out, trace


@pytest.mark.parametrize("name", ["qwen_loop_bf16_32", "qwen_loop_f32_16"])
def test_qwen_product_loop_matches_reference_trace(golden, cpu_ops, name):
    g = golden(name)
    pipe, out, trace = qwen_case(g)
    assert pipe.__class__.__name__ == "RegionEQwenImageEditPipeline"
    assert "".join(trace["kind"]) == "".join(g["kinds"].tolist())
    assert torch.equal(pipe._regione_manager.edited_ids.squeeze(0).int(), g["edited_ids"].squeeze(0))
    assert np.array_equal(np.array([float(x.double().sum()) for x in trace["latents"]]), g["lat_sum"].numpy())
    assert torch.equal(out, g["final"])


def test_qwen_plus_dispatch_and_gamma(golden, cpu_ops):
    from regione_amd.QwenImageEditPlus import inplace as qp
    pipe, out, trace = qwen_case(golden("qwen_loop_f32_16"), plus=True)
    assert pipe.__class__.__name__ == "RegionEQwenImageEditPlusPipeline"
    assert float(pipe.gamma[0]) == float(qp.gamma[0]) != float(pipe.__class__.__mro__[1].gamma[0])
    assert torch.isfinite(out).all()


def test_family_gamma_tables_match_reference_fixture(golden):
    """Every patch set ships the reference's fitted AVD decay table bit for bit (fp16), tests/golden/avd.npz."""
    import importlib
    g = golden("avd")
    for fam, key in (("FluxKontext", "gamma"), ("Step1XEdit", "gamma_step1x"), ("Step1XEditV1P2", "gamma_step1x_v1p2"),
                     ("QwenImageEdit", "gamma_qwen"), ("QwenImageEditPlus", "gamma_qwen_plus")):
        mod = importlib.import_module(f"regione_amd.{fam}.inplace")
        assert mod.gamma.dtype == torch.float16 and torch.equal(mod.gamma, g[key]), fam


def test_num_inference_steps_extension_with_caller_gamma(golden, cpu_ops):
    """Extension: N != 28 is accepted only together with a decay table of N-1
    entries; the default behaviour (assert 28, tool/RegionE.py:44, utils.py:391) is untouched.  The N = 50 loop of the
    patched pipeline equals the oracle's loop with the same table, bit for bit, plan included."""
    from regione_amd.tool.RegionE import resample_gamma
    g = golden("loop_bf16_32")
    h, w = g["h"], g["w"]
    L = h * w
    lat, img, _, _ = synth.make_edit_inputs(h, w, 8, synth.FluxConfig(), seed=g["seed"], dtype=torch.bfloat16)
    tgt = synth.region_target(h, w, tuple(int(x) for x in g["box"]), img, seed=g["tseed"], ramp=g["ramp"])
    pipe = H.FluxKontextPipeline(FakeTransformer(torch.cat([tgt, img[0].float()], 0), w, L))
    helper = RegionEHelper(pipe)
    with pytest.raises(AssertionError):
        helper.set_params(num_inference_steps=50)
    helper.set_params(num_inference_steps=50, gamma="resample", warmup_step=10, post_step=4, refresh_step="28",
                      threshold=0.93, cache_threshold=0.04)
    table = helper.config["gamma"]
    assert table.dtype == torch.float16 and table.numel() == 49
    assert torch.equal(resample_gamma(O.GAMMA["flux"], 28), torch.tensor(O.GAMMA["flux"], dtype=torch.float16))   # identity at N = 28
    helper.enable()
    trace = {}
    out = pipe(image=img, prompt_embeds=torch.zeros(1, 8, 4), pooled_prompt_embeds=torch.zeros(1, 4), height=h * 16,
               width=w * 16, latents=lat, num_inference_steps=50, return_dict=False, trace=trace)[0]
    st = O.RegionState()
    st.set_parameters(50, 10, 4, "28", 0.93, 0.04, True, gamma=table)
    tr = {}

    def fn(x, t, img_ids):
        tok = (img_ids[:, 0] * L + img_ids[:, 1] * w + img_ids[:, 2]).long()
        k = float(1.0 / (t.expand(x.shape[0]).to(x.dtype) / 1000).float()[0].item())
        return ((x.float() - torch.cat([tgt, img[0].float()], 0)[tok[:x.shape[1]]][None]) * k).to(x.dtype)
    ref = O.denoise(fn, st, lat, img, synth.flux_latent_ids(h, w), 8, h, w, trace=tr)
    kinds = "".join(trace["kind"])
    assert len(kinds) == 50 and kinds == "".join(tr["kind"]) and "C" in kinds and "R" in kinds
    assert kinds == "".join(O.derive_schedule(L, "flux", 10, 4, "28", 0.04, n=50, gamma=table)).replace("S", "F")
    assert torch.equal(pipe._regione_manager.edited_ids, st.edited_ids)
    assert torch.equal(out, ref)


def test_overlay_tool_matches_reference_arithmetic(tmp_path):
    """tools/overlay.py: edited token id -> (id // W_tok, id % W_tok) cell, nearest x16 upsampling, white at alpha 160
    composited like PIL.Image.alpha_composite (src/Step1X-Edit-v1p2/inplace.py:456-497)."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import overlay as OV
    from PIL import Image
    h, w = 800, 1328                                     # the 50 x 83 token grid of demo_0.png
    ids = torch.tensor([[0, 82, 83 * 49 + 82, 83 * 10 + 7]])
    m = OV.token_ids_to_mask(ids.numpy(), h, w)
    assert m.shape == (800, 1328) and int(m.sum()) == 4 * 256
    assert m[:16, :16].all() and m[:16, 82 * 16:].all() and m[49 * 16:, 82 * 16:].all() and m[160:176, 112:128].all()
    ref = torch.zeros(1, 1, 50, 83)
    ref[:, :, ids[0] // 83, ids[0] % 83] = 1           # the reference's token_ids2hw_mask
    ref = torch.nn.functional.interpolate(ref, scale_factor=16, mode="nearest")[0, 0].numpy().astype(np.uint8)
    assert np.array_equal(m, ref)
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
    img[..., 3] = 255
    layer = np.stack([m * 255] * 3 + [m * 160], -1).astype(np.uint8)
    want = np.asarray(Image.alpha_composite(Image.fromarray(img, "RGBA"), Image.fromarray(layer, "RGBA")))
    assert np.array_equal(OV.overlay_rgba(img, m), want)
    out = tmp_path / "o.png"
    OV.save_overlay(str(out), ids.numpy(), h, w)
    assert np.array_equal(np.asarray(Image.open(out)) > 0, m.astype(bool))
    with pytest.raises(ValueError):
        OV.token_ids_to_mask([50 * 83], h, w)


def test_scheduler_config_keys_honoured_or_refused():
    """Advisor finding (round 1): a host scheduler config must not be silently truncated.  shift_terminal (Qwen-Image-Edit
    ships 0.02), time_shift_type and invert_sigmas are implemented; karras / exponential / beta sigmas, stochastic sampling
    and unknown keys are refused; the default schedule is still the oracle's, bit for bit."""
    from regione_amd.harness import flux as H
    sig = np.linspace(1.0, 1 / 28, 28)
    mu = H.calculate_shift(4096)
    s = H.FlowMatchEulerDiscreteScheduler()
    s.set_timesteps(sigmas=sig, mu=mu)
    osig, ots = O.flow_match_schedule(28, 4096)
    assert torch.equal(s.sigmas, osig) and torch.equal(s.timesteps, ots)
    q = H.FlowMatchEulerDiscreteScheduler.from_config(dict(shift_terminal=0.02, _class_name="FlowMatchEulerDiscreteScheduler"))
    q.set_timesteps(sigmas=sig, mu=mu)
    assert abs(float(q.sigmas[-2]) - 0.02) < 1e-6 and float(q.sigmas[0]) == 1.0 and float(q.sigmas[-1]) == 0.0
    assert bool((q.sigmas[:-1] <= s.sigmas[:-1] + 1e-6).all())           # the tail is stretched DOWN to the terminal value
    lin = H.FlowMatchEulerDiscreteScheduler(time_shift_type="linear")
    lin.set_timesteps(sigmas=sig, mu=mu)
    assert not torch.equal(lin.sigmas, s.sigmas)
    for bad in (dict(use_karras_sigmas=True), dict(stochastic_sampling=True), dict(some_new_key=1), dict(time_shift_type="cubic")):
        with pytest.raises(ValueError):
            H.FlowMatchEulerDiscreteScheduler(**bad)


def test_reversed_k_oracle_is_the_same_arithmetic_in_another_summation_order():
    """tools/parity_full_depth.py's yardstick (the oracle with every Linear summed in the opposite order along K): on the toy trunk
    the two oracle runs agree to fp32 summation noise in fp32 and to bf16 rounding in bf16, differ bit-wise somewhere, and the
    context manager puts the oracle's own `_lin` back."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import parity_full_depth as P
    from oracle import regione_oracle as O
    from regione_amd import synth
    cfg = synth.FluxConfig(**synth.TOY)
    h = w = 8
    T = 16
    orig = O._lin
    outs = {}
    for dt in (torch.float32, torch.bfloat16):
        wts = synth.make_flux_weights(cfg, seed=3, dtype=dt, w_std=0.05)
        lat, img, prompt, pooled = synth.make_edit_inputs(h, w, T, cfg, seed=4, dtype=dt)
        ids = synth.flux_latent_ids(h, w)

        def fwd():
            st = O.RegionState()
            st.set_parameters(28, 6, 2, "16", 0.5, 0.04, True)
            st.refresh(img, ids, T, h, w)
            with torch.no_grad():
                return O.transformer_forward(wts, O.FluxCfg(**synth.TOY), st, [O.KVCache() for _ in range(cfg.n_layers)],
                                             torch.cat([lat, img], 1), prompt, pooled, torch.full([1], 0.5, dtype=dt), ids,
                                             torch.zeros(T, 3), torch.full([1], 2.5))
        a = fwd()
        with P.reversed_k_linears():
            assert O._lin is not orig
            b = fwd()
        assert O._lin is orig
        outs[dt] = (a, b)
    a, b = outs[torch.float32]
    assert not torch.equal(a, b) and float((a - b).abs().max() / a.abs().max()) < 1e-4
    a, b = outs[torch.bfloat16]
    assert O.psnr(b, a) > 40.0
