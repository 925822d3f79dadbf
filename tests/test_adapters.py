"""Host-pipeline adapter (regione_amd/adapters.py): a stock-pipeline-shaped object goes on the HIP engine.

Host stand-ins: the `torch.nn` module trees of tools/ref_stubs.py (the ones the reference itself ran on when the golden
fixtures were made) + a minimal host pipeline with the method surface the reference's `__call__` uses
(RegionE/FluxKontext/inplace.py:112-240, :396-410).  diffusers itself is not installed in this image.
"""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

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
def _stub_trunk(family):
    import ref_stubs as RS
    torch.manual_seed(3)
    mod = {"flux": RS.FluxTransformer2DModel, "step1x": RS.Step1XEditTransformer2DModel, "qwen": RS.QwenImageTransformer2DModel}[family]()
    with torch.no_grad():
        for n, p in mod.named_parameters():
            if p.dim() == 1 and not n.endswith("bias"):
                p.copy_(1.0 + 0.1 * torch.randn_like(p))
            elif p.dim() == 1:
                p.copy_(0.01 * torch.randn_like(p))
            else:
                p.copy_(0.05 * torch.randn_like(p))
    return mod.to(torch.bfloat16)


@pytest.mark.gpu
@pytest.mark.parametrize("family,cls", [("flux", "FluxKontextPipeline"), ("step1x", "Step1XEditPipeline"),
                                        ("step1x", "Step1XEditPipelineV1P2"), ("qwen", "QwenImageEditPipeline")])
def test_adopt_engine_from_host_module_tree(family, cls):
    """nn.Module trunk (host naming) -> engine; same output as the engine loaded from the renamed dict directly."""
    from regione_amd.harness import flux as HF, qwen as HQ, step1x as HS
    mod = _stub_trunk(family)
    pipe = type(cls, (), {"vae_scale_factor": 8})()
    pipe.transformer, pipe.scheduler = mod, None
    eng = A.adopt_engine(pipe)
    assert type(eng).__name__ == cls
    cfg = eng.transformer.cfg_model
    direct_sd = {A.map_key(k, family): v.detach().clone() for k, v in mod.state_dict().items()}
    tr_cls = {"flux": HF.FluxTransformer2DModel, "step1x": HS.Step1XEditTransformer2DModel, "qwen": HQ.QwenImageTransformer2DModel}[family]
    direct = type(eng)(tr_cls(cfg, "cuda").load_state_dict(direct_sd))
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
    a, b = eng(**kw)[0], direct(**kw)[0]
    assert torch.isfinite(a.float()).all() and a.float().abs().max() > 0
    assert torch.equal(a, b)


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
