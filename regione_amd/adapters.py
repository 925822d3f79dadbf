"""Host-pipeline adapter: put a stock diffusers editing pipeline on the HIP engine.

    engine = adopt_engine(pipe)                 # FluxKontext / Step1XEdit(V1P2) / QwenImageEdit(Plus)Pipeline
    hosted = adopt(pipe)                        # FluxKontextPipeline: image + prompt in, image out
    helper = RegionEHelper(hosted); helper.set_params(...); helper.enable()
    image = hosted(image=pil_image, prompt="...", guidance_scale=2.5).images[0]

`adopt_engine` reads the host transformer's `state_dict()` (tensor by tensor, straight to the GPU in bf16), infers the
trunk dimensions from the tensor shapes, maps the family's parameter names onto the engine's FLUX-layout names and
returns the matching `regione_amd.harness` pipeline (latent-level API).  `adopt` additionally keeps the host pipeline for
everything OUTSIDE the denoise loop - image preprocessing, prompt encoders, VAE encode / decode, post-processing - calling
the host's own methods in the order the reference's patched `__call__` does (RegionE/FluxKontext/inplace.py:112-240 and
:396-410), and runs the loop itself (:240-394) on the engine.

diffusers is not installed in the build image: the call sequence follows the reference's copy of the diffusers
`__call__`, and is exercised in tests/test_adapters.py against host-pipeline stand-ins with the same method surface
(tools/ref_stubs.py module trees).  Treat the first run against a real checkpoint as the remaining validation step.
"""
from __future__ import annotations

import re
from typing import Dict, Iterable, Iterator, Optional, Tuple

import torch

from .synth import FluxConfig, flux_param_shapes

# host parameter name -> engine (FLUX-layout) parameter name, per family; applied as ordered regex substitutions
_KEYMAPS: Dict[str, Tuple[Tuple[str, str], ...]] = {
    "flux": (),
    "step1x": ((r"^time_embed\.", "time_text_embed.timestep_embedder."),           # Step1XEditV1P2/inplace.py:590-600
               (r"^vec_embed\.", "time_text_embed.text_embedder.")),
    "qwen": ((r"^img_in\.", "x_embedder."), (r"^txt_in\.", "context_embedder."),    # QwenImageEdit/inplace.py:497-507
             (r"\.img_mod\.1\.", ".norm1.linear."), (r"\.txt_mod\.1\.", ".norm1_context.linear."),
             (r"\.img_mlp\.", ".ff."), (r"\.txt_mlp\.", ".ff_context.")),
}

_FAMILY_OF = {
    "FluxKontextPipeline": "flux",
    "Step1XEditPipeline": "step1x",
    "Step1XEditPipelineV1P2": "step1x",
    "QwenImageEditPipeline": "qwen",
    "QwenImageEditPlusPipeline": "qwen",
}


def map_key(name: str, family: str) -> str:
    for pat, rep in _KEYMAPS[family]:
        name = re.sub(pat, rep, name)
    return name


def infer_config(shapes: Dict[str, Tuple[int, ...]], axes_dim: Optional[Iterable[int]] = None) -> FluxConfig:
    """Trunk dimensions from (engine-named) parameter shapes; raises KeyError / ValueError on a foreign layout."""
    d, in_channels = shapes["x_embedder.weight"]
    joint_dim = shapes["context_embedder.weight"][1]
    n_double = 1 + max((int(m.group(1)) for k in shapes if (m := re.match(r"transformer_blocks\.(\d+)\.attn\.to_q\.weight$", k))),
                       default=-1)
    n_single = 1 + max((int(m.group(1)) for k in shapes if (m := re.match(r"single_transformer_blocks\.(\d+)\.proj_mlp\.weight$", k))),
                       default=-1)
    if n_double == 0:
        raise ValueError("no transformer_blocks.*.attn.to_q.weight in the host state dict: not an MMDiT trunk this engine knows")
    head_dim = shapes["transformer_blocks.0.attn.norm_q.weight"][0]
    if d % head_dim:
        raise ValueError(f"inner dim {d} is not a multiple of head_dim {head_dim}")
    pooled = shapes.get("time_text_embed.text_embedder.linear_1.weight")
    cfg = FluxConfig(in_channels=in_channels, n_double=n_double, n_single=n_single, heads=d // head_dim, head_dim=head_dim,
                     joint_dim=joint_dim, pooled_dim=pooled[1] if pooled else 768,
                     axes_dim=tuple(axes_dim) if axes_dim is not None else (16, 56, 56),
                     mlp_ratio=shapes["transformer_blocks.0.ff.net.0.proj.weight"][0] // d,
                     guidance_embeds="time_text_embed.guidance_embedder.linear_1.weight" in shapes,
                     pooled_embeds=pooled is not None, txt_norm="txt_norm.weight" in shapes)
    if sum(cfg.axes_dim) != head_dim:
        raise ValueError(f"rotary axes {cfg.axes_dim} do not sum to head_dim {head_dim}")
    return cfg


def _axes_dim(transformer):
    c = getattr(transformer, "config", None)
    for key in ("axes_dims_rope", "axes_dim"):
        v = (c.get(key) if hasattr(c, "get") else getattr(c, key, None)) if c is not None else None
        if v is not None:
            return tuple(v)
    pe = getattr(transformer, "pos_embed", None)
    return tuple(pe.axes_dim) if pe is not None and hasattr(pe, "axes_dim") else None


def _stream(sd, family: str, device) -> Iterator[Tuple[str, torch.Tensor]]:
    for k, v in sd.items():
        yield map_key(k, family), v.detach().to(device=device, dtype=torch.bfloat16)


def adopt_engine(pipe, device="cuda", family: Optional[str] = None, ignore_prefixes: Tuple[str, ...] = ("connector.",)):
    """The harness pipeline of the host pipeline's family, its transformer loaded from `pipe.transformer`.
    `ignore_prefixes`: host sub-modules that run in front of the hot path and stay on the host (Step1X-Edit's text
    connector, Step1XEditV1P2/inplace.py:606-609); any other name the trunk layout does not know is an error."""
    from .harness import flux as HF, qwen as HQ, step1x as HS
    name = pipe.__class__.__name__
    family = family or _FAMILY_OF.get(name)
    if family is None:
        raise NotImplementedError(f"no RegionE patch set for pipeline class {name}")
    sd = pipe.transformer.state_dict()
    shapes = {map_key(k, family): tuple(v.shape) for k, v in sd.items()}
    cfg = infer_config(shapes, _axes_dim(pipe.transformer))
    want = flux_param_shapes(cfg)
    extra = [k for k in shapes if k not in want and not k.startswith(tuple(ignore_prefixes))]
    missing = [k for k in want if k not in shapes]
    wrong = [k for k in want if k in shapes and tuple(shapes[k]) != tuple(want[k])]
    if extra or missing or wrong:
        raise KeyError(f"host transformer does not match the {family} trunk layout: missing {missing[:4]}, unexpected {extra[:4]}, "
                       f"shape mismatch {[(k, shapes[k], want[k]) for k in wrong[:4]]}")
    device = torch.device(device)
    if family == "flux":
        tr = HF.FluxTransformer2DModel(cfg, device)
        engine_cls = HF.FluxKontextPipeline
    elif family == "step1x":
        if cfg.guidance_embeds:
            raise ValueError("Step1X-Edit trunk with a guidance embedder: unexpected layout")
        tr = HS.Step1XEditTransformer2DModel(cfg, device)
        engine_cls = HS.Step1XEditPipelineV1P2 if name.endswith("V1P2") else HS.Step1XEditPipeline
    else:
        if cfg.n_single or cfg.pooled_embeds or cfg.guidance_embeds or not cfg.txt_norm:
            raise ValueError("Qwen-Image trunk expected: double-stream blocks only, timestep-only conditioning, txt_norm")
        tr = HQ.QwenImageTransformer2DModel(cfg, device)
        engine_cls = HQ.QwenImageEditPlusPipeline if "Plus" in name else HQ.QwenImageEditPipeline
    tr.load_state_dict_stream((k, v) for k, v in _stream(sd, family, device) if k in want)
    sched_cfg = dict(getattr(pipe.scheduler, "config", {}) or {}) if getattr(pipe, "scheduler", None) is not None else {}
    known = {k: sched_cfg[k] for k in ("num_train_timesteps", "shift", "use_dynamic_shifting", "base_shift", "max_shift",
                                       "base_image_seq_len", "max_image_seq_len") if k in sched_cfg}
    engine = engine_cls(tr, HF.FlowMatchEulerDiscreteScheduler(**known))
    if hasattr(pipe, "vae_scale_factor"):
        engine.vae_scale_factor = pipe.vae_scale_factor
    return engine


class HostedOutput(dict):
    def __init__(self, images):
        super().__init__(images=images)
        self.images = images


class HostedFluxKontextPipeline:
    """Host pipeline for everything outside the loop, HIP engine for the loop.  RegionEHelper(hosted) patches the engine."""

    def __init__(self, host, engine):
        self.host, self._regione_engine = host, engine

    @property
    def engine(self):
        return self._regione_engine

    @torch.no_grad()
    def __call__(self, image=None, prompt=None, prompt_2=None, negative_prompt=None, negative_prompt_2=None,
                 true_cfg_scale: float = 1.0, height=None, width=None, num_inference_steps: int = 28, guidance_scale: float = 3.5,
                 num_images_per_prompt: int = 1, generator=None, latents=None, prompt_embeds=None, pooled_prompt_embeds=None,
                 negative_prompt_embeds=None, negative_pooled_prompt_embeds=None, output_type: str = "pil",
                 return_dict: bool = True, max_sequence_length: int = 512, max_area: int = 1024 ** 2, _auto_resize: bool = True,
                 preferred_resolutions=None, trace=None):
        host, eng = self.host, self._regione_engine
        dev = eng.transformer.device
        multiple_of = host.vae_scale_factor * 2
        # 1. image preprocessing (inplace.py:115-140)
        if image is not None and not (isinstance(image, torch.Tensor) and image.size(1) == host.latent_channels):
            img = image[0] if isinstance(image, list) else image
            image_height, image_width = host.image_processor.get_default_height_width(img)
            if _auto_resize and preferred_resolutions:
                ar = image_width / image_height
                _, image_width, image_height = min((abs(ar - w / h), w, h) for w, h in preferred_resolutions)
            image_width, image_height = image_width // multiple_of * multiple_of, image_height // multiple_of * multiple_of
            image = host.image_processor.resize(image, image_height, image_width)
            image = host.image_processor.preprocess(image, image_height, image_width)
            height, width = image.shape[-2], image.shape[-1]
        else:
            height = height or host.default_sample_size * host.vae_scale_factor
            width = width or host.default_sample_size * host.vae_scale_factor
            ar = width / height
            width = round((max_area * ar) ** 0.5) // multiple_of * multiple_of
            height = round((max_area / ar) ** 0.5) // multiple_of * multiple_of
        # 2./3. prompts (inplace.py:142-208)
        if num_images_per_prompt != 1 or (isinstance(prompt, list) and len(prompt) != 1):
            raise ValueError("the region-aware loop is batch-1 (token_selector squeezes the batch, utils.py:337-343): "
                             "one image per call, shard images across GPUs")
        exec_dev = getattr(host, "_execution_device", dev)
        has_neg = negative_prompt is not None or (negative_prompt_embeds is not None and negative_pooled_prompt_embeds is not None)
        do_true_cfg = true_cfg_scale > 1 and has_neg
        prompt_embeds, pooled_prompt_embeds, _ = host.encode_prompt(
            prompt=prompt, prompt_2=prompt_2, prompt_embeds=prompt_embeds, pooled_prompt_embeds=pooled_prompt_embeds,
            device=exec_dev, num_images_per_prompt=1, max_sequence_length=max_sequence_length, lora_scale=None)
        if do_true_cfg:
            negative_prompt_embeds, negative_pooled_prompt_embeds, _ = host.encode_prompt(
                prompt=negative_prompt, prompt_2=negative_prompt_2, prompt_embeds=negative_prompt_embeds,
                pooled_prompt_embeds=negative_pooled_prompt_embeds, device=exec_dev, num_images_per_prompt=1,
                max_sequence_length=max_sequence_length, lora_scale=None)
        # 4. latents: the host packs noise and the VAE-encoded condition image (inplace.py:210-226)
        latents, image_latents, _, _ = host.prepare_latents(image, 1, eng.transformer.cfg_model.in_channels // 4, height, width,
                                                            prompt_embeds.dtype, exec_dev, generator, latents)
        if image_latents is None:
            raise ValueError("FluxKontext editing needs a condition image")
        bf = lambda t: t.to(device=dev, dtype=torch.bfloat16) if t is not None else None
        # 5.-6. the denoise loop on the engine (inplace.py:228-394)
        kw = dict(image=bf(image_latents), prompt_embeds=bf(prompt_embeds), pooled_prompt_embeds=bf(pooled_prompt_embeds),
                  height=height, width=width, num_inference_steps=num_inference_steps, guidance_scale=guidance_scale,
                  latents=bf(latents), return_dict=False, true_cfg_scale=true_cfg_scale,
                  negative_prompt_embeds=bf(negative_prompt_embeds) if do_true_cfg else None,
                  negative_pooled_prompt_embeds=bf(negative_pooled_prompt_embeds) if do_true_cfg else None)
        if trace is not None:
            kw["trace"] = trace
        latents = eng(**kw)[0]
        # 7. decode (inplace.py:396-410)
        if output_type == "latent":
            out = latents
        else:
            vae = host.vae
            lat = host._unpack_latents(latents.to(vae.dtype if hasattr(vae, "dtype") else latents.dtype), height, width,
                                       host.vae_scale_factor)
            lat = lat / vae.config.scaling_factor + vae.config.shift_factor
            out = host.image_processor.postprocess(vae.decode(lat, return_dict=False)[0], output_type=output_type)
        if hasattr(host, "maybe_free_model_hooks"):
            host.maybe_free_model_hooks()
        return HostedOutput(out) if return_dict else (out,)


def adopt(pipe, device="cuda"):
    """Hosted pipeline (image + prompt in, image out) - FLUX.1 Kontext this round; the CFG families' prompt paths go through
    their VLM encoders and are adopted at the latent level (`adopt_engine`)."""
    if pipe.__class__.__name__ != "FluxKontextPipeline":
        raise NotImplementedError(f"hosted call for {pipe.__class__.__name__} is not built yet: use adopt_engine(pipe) and feed "
                                  "prompt embeddings / packed latents")
    return HostedFluxKontextPipeline(pipe, adopt_engine(pipe, device))
