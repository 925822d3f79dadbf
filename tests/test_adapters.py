"""Host-pipeline adapter (regione_amd/adapters.py): a stock-pipeline-shaped object goes on the HIP engine.

Host stand-ins: the `torch.nn` module trees of tests/host_trunks.py (host parameter naming, their own vanilla CPU forward and
stock attention processors) + a minimal host pipeline with the method surface the reference's `__call__` uses
(RegionE/FluxKontext/inplace.py:112-240, :396-410).  diffusers itself is not installed in this image.

The GPU cases compare the adopted HIP engine with the HOST TRUNK'S OWN forward (torch-CPU bf16 eager on the nn.Module, host
parameter names): two independent implementations, so a key map that satisfies the shape check but wires a weight to the
wrong place (img_mlp <-> txt_mlp, to_out <-> to_add_out ...) fails - the last case shows that it does.
"""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from regione_amd import adapters as A, synth  # noqa: E402


def _host_names(sd, family):
    """engine (FLUX-layout) names -> the family's real host names (inverse of adapters._KEYMAPS)."""
    inv = {"flux": (), "step1x": (("time_text_embed.timestep_embedder.", "time_embed."), ("time_text_embed.text_embedder.", "vec_embed.")),
           "qwen": (("x_embedder.", "img_in."), ("context_embedder.", "txt_in."), (".norm1.linear.", ".img_mod.1."),
                    (".norm1_context.linear.", ".txt_mod.1."), (".ff.", ".img_mlp."), (".ff_context.", ".txt_mlp."))}[family]
    out = {}
    for k, v in sd.items():
        for a, b in inv:
            if k.startswith("norm_out."):
                break
            k = k.replace(a, b)
        out[k] = v
    return out


class _Tr:
    def __init__(self, sd, axes=(16, 56, 56)):
        self._sd, self.config = sd, {"axes_dims_rope": axes}

    def state_dict(self):
        return self._sd


def _fake_pipe(cls_name, sd):
    return type(cls_name, (), {})(), sd


@pytest.mark.parametrize("family,cls,cfgkw", [
    ("flux", "FluxKontextPipeline", dict(synth.TOY)),
    ("step1x", "Step1XEditPipelineV1P2", dict(synth.TOY, guidance_embeds=False)),
    ("qwen", "QwenImageEditPlusPipeline", dict(synth.QWEN_TOY)),
])
def test_key_maps_and_config_inference(family, cls, cfgkw):
    cfg = synth.FluxConfig(**cfgkw)
    host = {k: torch.empty(s, device="meta") for k, s in _host_names(synth.flux_param_shapes(cfg), family).items()}
    if family == "qwen":
        assert "transformer_blocks.0.img_mod.1.weight" in host and "img_in.weight" in host and "transformer_blocks.1.txt_mlp.net.2.bias" in host
    shapes = {A.map_key(k, family): tuple(v.shape) for k, v in host.items()}
    got = A.infer_config(shapes, (16, 56, 56))
    assert got == cfg
    assert shapes == {k: tuple(s) for k, s in synth.flux_param_shapes(cfg).items()}


def test_public_dimensions_inferred_from_shapes():
    for kw in ({}, dict(synth.QWEN), dict(guidance_embeds=False)):
        cfg = synth.FluxConfig(**kw)
        assert A.infer_config(synth.flux_param_shapes(cfg), cfg.axes_dim) == cfg


def test_foreign_layout_is_refused_before_anything_is_built():
    cfg = synth.FluxConfig(**synth.TOY)
    sd = {k: torch.empty(s, device="meta") for k, s in synth.flux_param_shapes(cfg).items()}
    pipe = type("FluxKontextPipeline", (), {})()
    pipe.scheduler = None
    bad = dict(sd)
    del bad["transformer_blocks.1.attn.to_v.bias"]
    pipe.transformer = _Tr(bad)
    with pytest.raises(KeyError, match="missing"):
        A.adopt_engine(pipe)
    bad = dict(sd, **{"controlnet_x_embedder.weight": torch.empty(4, 4, device="meta")})
    pipe.transformer = _Tr(bad)
    with pytest.raises(KeyError, match="unexpected"):
        A.adopt_engine(pipe)
    bad = dict(sd, **{"single_transformer_blocks.0.proj_out.weight": torch.empty(256, 999, device="meta")})
    pipe.transformer = _Tr(bad)
    with pytest.raises(KeyError, match="shape mismatch"):
        A.adopt_engine(pipe)
    pipe.transformer = _Tr(sd, axes=(16, 24, 24))
    with pytest.raises(ValueError, match="rotary axes"):
        A.adopt_engine(pipe)
    other = type("StableDiffusionPipeline", (), {})()
    with pytest.raises(NotImplementedError):
        A.adopt_engine(other)
    with pytest.raises(NotImplementedError):
        A.adopt(other)


# ------------------------------------------------------------------------------------------------ GPU
from host_standins import stub_trunk as _stub_trunk  # noqa: E402
from oracle import regione_oracle as O  # noqa: E402  (PSNR helper only)


def _host_forward(family, mod, lat, img, prompt, y, h, w, t):
    """The host trunk's OWN vanilla forward on the CPU (tests/host_trunks.py), the way its stock pipeline calls it."""
    x = torch.cat([lat, img], dim=1)
    T = prompt.shape[1]
    ts = t.expand(1).to(torch.bfloat16) / 1000
    with torch.no_grad():
        if family == "flux":
            return mod(hidden_states=x, encoder_hidden_states=prompt, pooled_projections=y, timestep=ts,
                       img_ids=synth.flux_latent_ids(h, w), txt_ids=torch.zeros(T, 3), guidance=torch.full([1], 2.5),
                       return_dict=False)[0]
        if family == "step1x":
            mod.set_vec(prompt, y)                               # the stub connector hands back the vector registered for a prompt
            return mod(hidden_states=x, encoder_hidden_states=prompt, prompt_embeds_mask=None, timestep=ts,
                       img_ids=synth.flux_latent_ids(h, w), txt_ids=torch.zeros(T, 3), return_dict=False)[0]
        return mod(hidden_states=x, encoder_hidden_states=prompt, timestep=ts, img_shapes=[[(1, h, w), (1, h, w)]],
                   txt_seq_lens=[T], return_dict=False)[0]


def _engine_forward(family, eng, lat, img, prompt, y, h, w, t):
    tr = eng.transformer
    x = torch.cat([lat, img], dim=1).cuda()
    T = prompt.shape[1]
    ts = t.expand(1).to(torch.bfloat16) / 1000
    if family == "flux":
        return tr(hidden_states=x, encoder_hidden_states=prompt.cuda(), pooled_projections=y.cuda(), timestep=ts,
                  img_ids=synth.flux_latent_ids(h, w), txt_ids=torch.zeros(T, 3), guidance=torch.full([1], 2.5),
                  return_dict=False)[0]
    if family == "step1x":
        tr.set_vec((y.cuda(),))
        return tr(hidden_states=x, encoder_hidden_states=prompt.cuda(), prompt_embeds_mask=None, timestep=ts, guidance=None,
                  img_ids=synth.flux_latent_ids(h, w), txt_ids=torch.zeros(T, 3), return_dict=False)[0]
    return tr(hidden_states=x, encoder_hidden_states=prompt.cuda(), timestep=ts, img_shapes=[[(1, h, w), (1, h, w)]],
              latent_ids=torch.arange(2 * h * w), return_dict=False)[0]


_CASES = [("flux", "FluxKontextPipeline"), ("step1x", "Step1XEditPipeline"), ("step1x", "Step1XEditPipelineV1P2"),
          ("qwen", "QwenImageEditPipeline")]


@pytest.mark.gpu
@pytest.mark.parametrize("family,cls", _CASES)
def test_adopted_engine_matches_the_host_trunks_own_forward(family, cls):
    """nn.Module trunk (host naming, stock processors) -> adopt_engine -> HIP forward, against the SAME module's own CPU
    forward: >= 40 dB at three timesteps.  Cross-implementation: the host side never touches regione_amd."""
    mod = _stub_trunk(family)
    pipe = type(cls, (), {"vae_scale_factor": 8})()
    pipe.transformer, pipe.scheduler = mod, None
    eng = A.adopt_engine(pipe)
    assert type(eng).__name__ == cls
    cfg = eng.transformer.cfg_model
    h = w = 16
    lat, img, prompt, y = synth.make_edit_inputs(h, w, 32, cfg, seed=9, dtype=torch.bfloat16)
    for t in (torch.tensor(1000.0), torch.tensor(612.0), torch.tensor(87.0)):
        ref = _host_forward(family, mod, lat, img, prompt, y, h, w, t)
        got = _engine_forward(family, eng, lat, img, prompt, y, h, w, t).cpu()
        assert got.shape == ref.shape == (1, 2 * h * w, cfg.in_channels)
        p = O.psnr(got.float(), ref.float())
        print(f"[adapter x-check] {cls} t={float(t):.0f}: {p:.1f} dB vs the host trunk's own forward")
        assert torch.isfinite(got.float()).all() and p >= 40.0, (cls, float(t), p)


@pytest.mark.gpu
@pytest.mark.parametrize("family,cls,swap", [
    ("qwen", "QwenImageEditPipeline", (".img_mlp.", ".txt_mlp.")),
    ("flux", "FluxKontextPipeline", (".attn.to_out.0.", ".attn.to_add_out.")),
    ("step1x", "Step1XEditPipeline", (".ff.net.2.", ".ff_context.net.2.")),
])
def test_cross_check_catches_a_miswired_key_map(family, cls, swap):
    """The failure the engine-vs-engine comparison could not see: a host state dict whose `swap` weights trade places has
    the right key set and shapes (adopt_engine accepts it), and the cross-check against the host forward rejects it."""
    mod = _stub_trunk(family)

    class _Swapped:
        config = mod.config
        pos_embed = mod.pos_embed

        def state_dict(self):
            sd, a, b = mod.state_dict(), swap[0], swap[1]
            return {(k.replace(a, b) if a in k else k.replace(b, a) if b in k else k): v for k, v in sd.items()}
    pipe = type(cls, (), {"vae_scale_factor": 8})()
    pipe.transformer, pipe.scheduler = _Swapped(), None
    eng = A.adopt_engine(pipe)                                          # same names, same shapes: accepted
    cfg = eng.transformer.cfg_model
    h = w = 16
    lat, img, prompt, y = synth.make_edit_inputs(h, w, 32, cfg, seed=9, dtype=torch.bfloat16)
    t = torch.tensor(612.0)
    ref = _host_forward(family, mod, lat, img, prompt, y, h, w, t)
    got = _engine_forward(family, eng, lat, img, prompt, y, h, w, t).cpu()
    assert O.psnr(got.float(), ref.float()) < 30.0


@pytest.mark.gpu
@pytest.mark.parametrize("family,cls", _CASES)
def test_adopted_engine_runs_the_pipeline_loop(family, cls):
    """The adopted engine's latent-level pipeline call (vanilla loop, 4 steps) on the adopted weights: finite, non-trivial."""
    mod = _stub_trunk(family)
    pipe = type(cls, (), {"vae_scale_factor": 8})()
    pipe.transformer, pipe.scheduler = mod, None
    eng = A.adopt_engine(pipe)
    cfg = eng.transformer.cfg_model
    h = w = 16
    lat, img, prompt, y = synth.make_edit_inputs(h, w, 32, cfg, seed=9, dtype=torch.bfloat16)
    kw = dict(image=img.cuda(), prompt_embeds=prompt.cuda(), height=h * 16, width=w * 16, latents=lat.cuda(), return_dict=False,
              num_inference_steps=4)
    if family != "qwen":
        kw.update(pooled_prompt_embeds=y.cuda())
    if family != "flux":
        kw.update(negative_prompt_embeds=prompt.cuda().flip(1), true_cfg_scale=3.0)
        if family == "step1x":
            kw.update(negative_pooled_prompt_embeds=y.cuda().flip(1))
    a = eng(**kw)[0]
    assert torch.isfinite(a.float()).all() and a.float().abs().max() > 0


class _ImageProcessor:
    def get_default_height_width(self, img):
        return img.shape[-2], img.shape[-1]

    def resize(self, image, h, w):
        return torch.nn.functional.interpolate(image, size=(h, w), mode="nearest")

    def preprocess(self, image, h, w):
        return image * 2 - 1

    def postprocess(self, image, output_type="pt"):
        return (image / 2 + 0.5).clamp(0, 1)


class _Vae:
    """16-channel 8x 'VAE': average pooling / nearest upsampling of a fixed channel lift (deterministic, invertible enough)."""
    dtype = torch.float32

    class config:
        scaling_factor, shift_factor = 0.36, 0.12

    def encode_pixels(self, image):
        x = torch.nn.functional.avg_pool2d(image, 8)
        return torch.cat([x * (0.5 + 0.1 * i) for i in range(6)], 1)[:, :16]

    def decode(self, lat, return_dict=False):
        return (torch.nn.functional.interpolate(lat[:, :3], scale_factor=8, mode="nearest"),)


def _host_pipeline(trunk):
    calls = []

    class FluxKontextPipeline:
        vae_scale_factor, latent_channels, default_sample_size = 8, 16, 128
        _execution_device = torch.device("cpu")
        image_processor, vae = _ImageProcessor(), _Vae()

        def encode_prompt(self, prompt=None, prompt_2=None, prompt_embeds=None, pooled_prompt_embeds=None, device=None,
                          num_images_per_prompt=1, max_sequence_length=512, lora_scale=None):
            calls.append(("encode_prompt", prompt))
            if prompt_embeds is None:
                g = torch.Generator().manual_seed(sum(map(ord, prompt)))
                prompt_embeds = torch.randn(1, 32, 256, generator=g).to(torch.bfloat16)
                pooled_prompt_embeds = torch.randn(1, 64, generator=g).to(torch.bfloat16)
            return prompt_embeds, pooled_prompt_embeds, torch.zeros(prompt_embeds.shape[1], 3)

        @staticmethod
        def _pack_latents(x):
            b, c, h, w = x.shape
            return x.view(b, c, h // 2, 2, w // 2, 2).permute(0, 2, 4, 1, 3, 5).reshape(b, (h // 2) * (w // 2), c * 4)

        @staticmethod
        def _unpack_latents(latents, height, width, vae_scale_factor):
            calls.append(("unpack", tuple(latents.shape)))
            b, n, c = latents.shape
            h, w = 2 * (height // (vae_scale_factor * 2)), 2 * (width // (vae_scale_factor * 2))
            return latents.view(b, h // 2, w // 2, c // 4, 2, 2).permute(0, 3, 1, 4, 2, 5).reshape(b, c // 4, h, w)

        def prepare_latents(self, image, batch_size, num_channels_latents, height, width, dtype, device, generator, latents=None):
            calls.append(("prepare_latents", height, width, num_channels_latents))
            image_latents = self._pack_latents(self.vae.encode_pixels(image)).to(dtype)
            if latents is None:
                latents = torch.randn(image_latents.shape, generator=generator).to(dtype)
            return latents, image_latents, None, None

        def maybe_free_model_hooks(self):
            calls.append(("free",))
    p = FluxKontextPipeline()
    p.transformer, p.scheduler = trunk, None
    return p, calls


@pytest.mark.gpu
def test_hosted_flux_call_image_and_prompt_in_image_out():
    from regione_amd import RegionEHelper
    host, calls = _host_pipeline(_stub_trunk("flux"))
    hosted = A.adopt(host)
    g = torch.Generator().manual_seed(5)
    picture = torch.rand(1, 3, 256, 256, generator=g)
    picture[:, :, 64:160, 96:192] = 0.0                      # something for the prompt to 'edit'
    # _auto_resize=False: keep the toy 256 x 256 input (the default snaps to Kontext's training resolutions, ~1024^2 area)
    out = hosted(image=picture, prompt="make the square red", generator=torch.Generator().manual_seed(1), output_type="pt",
                 guidance_scale=2.5, _auto_resize=False)
    assert tuple(out.images.shape) == (1, 3, 256, 256) and torch.isfinite(out.images).all()
    assert [c[0] for c in calls] == ["encode_prompt", "prepare_latents", "unpack", "free"]
    assert calls[1][1:] == (256, 256, 16)
    # the loop the hosted call ran == the engine's own latent-level call on the same packed inputs
    lat = hosted(image=picture, prompt="make the square red", generator=torch.Generator().manual_seed(1), output_type="latent",
                 guidance_scale=2.5, _auto_resize=False).images
    pe, pp, _ = host.encode_prompt(prompt="make the square red")
    l0, il, _, _ = host.prepare_latents(picture * 2 - 1, 1, 16, 256, 256, torch.bfloat16, None, torch.Generator().manual_seed(1))
    direct = hosted.engine(image=il.cuda(), prompt_embeds=pe.cuda(), pooled_prompt_embeds=pp.cuda(), height=256, width=256,
                           latents=l0.cuda(), guidance_scale=2.5, return_dict=False)[0]
    assert torch.equal(lat, direct)
    # RegionEHelper on the hosted pipeline patches the engine; the hosted call then runs the region-aware loop
    helper = RegionEHelper(hosted)
    helper.set_params(threshold=0.5)
    helper.enable()
    trace = {}
    reg = hosted(image=picture, prompt="make the square red", generator=torch.Generator().manual_seed(1), output_type="latent",
                 guidance_scale=2.5, trace=trace, _auto_resize=False).images
    assert "".join(trace["kind"]).startswith("FFFFFF") and len(trace["kind"]) == 28
    assert torch.isfinite(reg.float()).all()
    helper.disable()
    again = hosted(image=picture, prompt="make the square red", generator=torch.Generator().manual_seed(1), output_type="latent",
                   guidance_scale=2.5, _auto_resize=False).images
    assert torch.equal(again, lat)
    with pytest.raises(ValueError, match="batch-1"):
        hosted(image=picture, prompt=["a", "b"])
