"""Host-pipeline adapter: a STOCK diffusers editing pipeline on the HIP engine, with the
reference's call shape (RegionE/README.md:85-113):

    helper = RegionEHelper(pipe)                # FluxKontext / Step1XEdit(V1P2) / QwenImageEdit(Plus)Pipeline object
    helper.set_params(...); helper.enable()     # adopts the transformer weights once, patches the engine, swaps pipe.__class__
    image = pipe(image=pil_image, prompt="...").images[0]        # the user keeps calling the pipeline itself
    helper.disable()                            # pipe is the stock pipeline again

Lower levels, for callers that want them:
    engine = adopt_engine(pipe)                 # latent-level HIP pipeline of the host's family (weights adopted)
    hosted = adopt(pipe)                        # wrapper object: hosted(image=..., prompt=...) without touching pipe.__class__

`adopt_engine` reads the host transformer's `state_dict()` (tensor by tensor, straight to the GPU in bf16), infers the
trunk dimensions from the tensor shapes, maps the family's parameter names onto the engine's FLUX-layout names and
returns the matching `regione_amd.harness` pipeline (latent-level API).  `adopt` additionally keeps the host pipeline for
everything OUTSIDE the denoise loop - image preprocessing, prompt encoders, VAE encode / decode, post-processing - calling
the host's own methods in the order the reference's patched `__call__` does (RegionE/FluxKontext/inplace.py:112-240 and
:396-410), and runs the loop itself (:240-394) on the engine.

diffusers is not installed in the build image: the call sequence follows the reference's copy of the diffusers
`__call__`, and is exercised in tests/test_adapters.py against host-pipeline stand-ins with the same method surface
(tests/host_trunks.py module trees, whose own vanilla forward the adopted engine is compared with).  Treat the first run against a real checkpoint as the remaining validation step.
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
    # every key travels: the engine's scheduler implements shift_terminal / time_shift_type / invert_sigmas and REFUSES
    # what it does not implement (karras / exponential / beta sigmas, stochastic sampling, unknown keys)
    engine = engine_cls(tr, HF.FlowMatchEulerDiscreteScheduler(**sched_cfg))
    if hasattr(pipe, "vae_scale_factor"):
        engine.vae_scale_factor = pipe.vae_scale_factor
    return engine


# ---------------------------------------------------------------------------------------------------------------------
# hosted calls: everything OUTSIDE the denoise loop on the host pipeline's own methods, the loop on the engine
# ---------------------------------------------------------------------------------------------------------------------
# Kontext's training resolutions (diffusers FluxKontextPipeline; reference copy RegionE/FluxKontext/utils.py:18-36):
# the default target of `_auto_resize`
PREFERRED_KONTEXT_RESOLUTIONS = [(672, 1568), (688, 1504), (720, 1456), (752, 1392), (800, 1328), (832, 1248), (880, 1184),
                                 (944, 1104), (1024, 1024), (1104, 944), (1184, 880), (1248, 832), (1328, 800), (1392, 752),
                                 (1456, 720), (1504, 688), (1568, 672)]
STEP1X_DEFAULT_NEGATIVE = ("worst quality, wrong limbs, unreasonable limbs, normal quality, low quality, low res, blurry, text, "
                           "watermark, logo, banner, extra digits, cropped, jpeg artifacts, signature, username, error, sketch ,"
                           "duplicate, ugly, monochrome, horror, geometry, mutation, disgusting")    # Step1XEdit/inplace.py:231
QWEN_CONDITION_IMAGE_SIZE, QWEN_VAE_IMAGE_SIZE = 384 * 384, 1024 * 1024                                # QwenImageEditPlus/inplace.py:53-54


class HostedOutput(dict):
    def __init__(self, images, timing=None):
        super().__init__(images=images)
        self.images = images
        self.timing = timing or {}           # wall-clock seconds of the three stages: encode / loop / decode


def _bf(t, dev):
    return t.to(device=dev, dtype=torch.bfloat16) if t is not None else None


def _one_image_only(prompt, num_images_per_prompt):
    if num_images_per_prompt != 1 or (isinstance(prompt, list) and len(prompt) != 1):
        raise ValueError("the region-aware loop is batch-1 (token_selector squeezes the batch, utils.py:337-343): "
                         "one image per call, shard images across GPUs")


# stock-pipeline arguments the hosted calls take no action on, with the value at which ignoring them changes nothing
_IGNORABLE_DEFAULTS = {"max_sequence_length": 512, "num_images_per_prompt": 1, "guidance_scale": None, "ip_adapter_image": None,
                       "ip_adapter_image_embeds": None, "negative_ip_adapter_image": None, "negative_ip_adapter_image_embeds": None,
                       "joint_attention_kwargs": None, "attention_kwargs": None, "prompt_2": None, "negative_prompt_2": None}


def _refuse_unused(fn_name: str, unused: dict):
    """A stock-pipeline argument the hosted loop does not implement must not be dropped silently (advisor finding, round 2:
    the reference honours `joint_attention_kwargs`, IP-adapter inputs ... - FluxKontext/inplace.py:76-110): None or the
    documented default passes, anything else raises."""
    for k, v in unused.items():
        if v is None or (k in _IGNORABLE_DEFAULTS and v == _IGNORABLE_DEFAULTS[k]):
            continue
        known = k in _IGNORABLE_DEFAULTS
        raise (NotImplementedError if known else TypeError)(
            f"{fn_name}: argument {k}={v!r} is {'not implemented by' if known else 'unknown to'} the hosted HIP loop")


def _loop_extras(kw: dict, sigmas, callback_on_step_end, callback_on_step_end_tensor_inputs):
    """`sigmas` (inplace.py:229-242) and `callback_on_step_end` (:376-383) travel to the engine's loop."""
    if sigmas is not None:
        kw["sigmas"] = sigmas
    if callback_on_step_end is not None:
        kw["callback_on_step_end"] = callback_on_step_end
        kw["callback_on_step_end_tensor_inputs"] = tuple(callback_on_step_end_tensor_inputs)
    return kw


_NO_HIP_VAE = object()


def hip_vae_for(host, dev):
    """The host's AutoencoderKL decoder adopted onto the HIP kernels (regione_amd/vae.py) - once per host pipeline,
    kept on it as `_regione_hip_vae`.  None when the host's VAE is not of that layout (Qwen-Image's 3-D causal VAE, a module with a
    post_quant_conv): the host module then decodes, exactly as in the reference (`self.vae.decode`, FluxKontext/inplace.py:396-402).  A VAE
    that HAS the layout but cannot be adopted (unknown parameters, other widths) raises instead of silently running the slower module;
    `pipe._regione_hip_vae = False` before the first call keeps the host module on purpose."""
    cached = host.__dict__.get("_regione_hip_vae", _NO_HIP_VAE)
    if cached is False or cached is None:
        return None
    if cached is not _NO_HIP_VAE:
        return cached
    vae = getattr(host, "vae", None)
    dec = getattr(vae, "decoder", None)
    ok = dec is not None and all(hasattr(dec, n) for n in ("conv_in", "mid_block", "up_blocks", "conv_norm_out", "conv_out")) and \
        getattr(vae, "post_quant_conv", None) is None and hasattr(dec, "state_dict")
    if not ok:
        host._regione_hip_vae = None
        return None
    from . import vae as V
    cfg = getattr(vae, "config", None)
    kw = {}
    for name in ("block_out_channels", "latent_channels", "layers_per_block"):
        v = getattr(cfg, name, None) if cfg is not None else None
        if v is not None:
            kw[name] = tuple(v) if name == "block_out_channels" else int(v)
    host._regione_hip_vae = V.HipVaeDecoder(dec.state_dict(), dev, **kw)
    return host._regione_hip_vae


def hip_vae_encoder_for(host, dev):
    """The host's AutoencoderKL ENCODER on the HIP kernels (regione_amd/vae.py HipVaeEncoder), adopted once per host pipeline
    (`_regione_hip_vae_encoder`); None when the VAE has no such encoder, carries a quant_conv, or `pipe._regione_hip_vae = False`."""
    if host.__dict__.get("_regione_hip_vae", _NO_HIP_VAE) is False:
        return None
    cached = host.__dict__.get("_regione_hip_vae_encoder", _NO_HIP_VAE)
    if cached is not _NO_HIP_VAE:
        return cached
    vae = getattr(host, "vae", None)
    enc = getattr(vae, "encoder", None)
    ok = enc is not None and all(hasattr(enc, n) for n in ("conv_in", "down_blocks", "mid_block", "conv_norm_out", "conv_out")) and \
        getattr(vae, "quant_conv", None) is None and hasattr(enc, "state_dict")
    if not ok:
        host._regione_hip_vae_encoder = None
        return None
    from . import vae as V
    cfg = getattr(vae, "config", None)
    kw = {}
    for name in ("block_out_channels", "latent_channels", "layers_per_block"):
        v = getattr(cfg, name, None) if cfg is not None else None
        if v is not None:
            kw[name] = tuple(v) if name == "block_out_channels" else int(v)
    host._regione_hip_vae_encoder = V.HipVaeEncoder(enc.state_dict(), dev, **kw)
    return host._regione_hip_vae_encoder


class _hip_vae_encode:
    """`with _hip_vae_encode(host, dev): host.prepare_latents(...)` - the host's own `prepare_latents` (resize, `_encode_vae_image`,
    `retrieve_latents`, shift / scale, packing: its code, untouched) runs with `vae.encode` answered by the HIP encoder for single 4-D
    images; anything else (batches, 5-D video-style inputs) falls through to the module's own method.  The binding is undone on exit."""

    def __init__(self, host, dev):
        self.host, self.dev = host, dev

    def __enter__(self):
        self.enc = hip_vae_encoder_for(self.host, self.dev)
        if self.enc is None:
            return self
        vae, enc, dev = self.host.vae, self.enc, self.dev
        self.orig = vae.__dict__.get("encode", _NO_HIP_VAE)
        own = vae.encode

        def encode(x, return_dict=True, **kw):
            if isinstance(x, torch.Tensor) and x.dim() == 4 and x.shape[0] == 1 and x.shape[1] == 3 and not kw:
                out = enc.encode_dist(x.to(dev))
                return out if return_dict else (out.latent_dist,)
            return own(x, return_dict=return_dict, **kw)
        vae.encode = encode
        return self

    def __exit__(self, *a):
        if self.enc is not None:
            if self.orig is _NO_HIP_VAE:
                del self.host.vae.__dict__["encode"]
            else:
                self.host.vae.encode = self.orig
        return False


def _decode_image(host, vae, lat, dev):
    """`vae.decode(lat, return_dict=False)[0]` - on the HIP decoder when the host's VAE is an AutoencoderKL (one image per call)."""
    hv = hip_vae_for(host, dev)
    if hv is not None and lat.dim() == 4 and lat.shape[0] == 1:
        return hv.decode(lat.to(dev))
    return vae.decode(lat, return_dict=False)[0]


class _Clock:
    """Wall-clock of the stages of an edit (end-to-end = encode + loop + decode)."""

    def __init__(self, dev):
        import time
        self.dev, self.t, self.time, self.out = dev, None, time, {}
        self.mark(None)

    def mark(self, name):
        if torch.cuda.is_available():
            torch.cuda.synchronize(self.dev)
        now = self.time.perf_counter()
        if name is not None:
            self.out[name] = now - self.t
        self.t = now


def _hosted_flux(host, eng, image=None, prompt=None, prompt_2=None, negative_prompt=None, negative_prompt_2=None,
                 true_cfg_scale: float = 1.0, height=None, width=None, num_inference_steps: int = 28, guidance_scale: float = 3.5,
                 num_images_per_prompt: int = 1, generator=None, latents=None, prompt_embeds=None, pooled_prompt_embeds=None,
                 negative_prompt_embeds=None, negative_pooled_prompt_embeds=None, output_type: str = "pil",
                 return_dict: bool = True, max_sequence_length: int = 512, max_area: int = 1024 ** 2, _auto_resize: bool = True,
                 preferred_resolutions=None, trace=None, sigmas=None, callback_on_step_end=None,
                 callback_on_step_end_tensor_inputs=("latents",), **unused):
    """FluxKontextPipeline.__call__ around the engine loop (RegionE/FluxKontext/inplace.py:112-240, :396-410)."""
    _refuse_unused("FluxKontextPipeline.__call__", unused)
    dev = eng.transformer.device
    clk = _Clock(dev)
    multiple_of = host.vae_scale_factor * 2
    preferred = PREFERRED_KONTEXT_RESOLUTIONS if preferred_resolutions is None else preferred_resolutions
    # 1. image preprocessing (inplace.py:115-140)
    if image is not None and not (isinstance(image, torch.Tensor) and image.size(1) == host.latent_channels):
        img = image[0] if isinstance(image, list) else image
        image_height, image_width = host.image_processor.get_default_height_width(img)
        if _auto_resize and preferred:
            ar = image_width / image_height
            _, image_width, image_height = min((abs(ar - w / h), w, h) for w, h in preferred)
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
    _one_image_only(prompt, num_images_per_prompt)
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
    with _hip_vae_encode(host, dev):
        latents, image_latents, _, _ = host.prepare_latents(image, 1, eng.transformer.cfg_model.in_channels // 4, height, width,
                                                            prompt_embeds.dtype, exec_dev, generator, latents)
    if image_latents is None:
        raise ValueError("FluxKontext editing needs a condition image")
    clk.mark("encode_s")
    # 5.-6. the denoise loop on the engine (inplace.py:228-394)
    kw = dict(image=_bf(image_latents, dev), prompt_embeds=_bf(prompt_embeds, dev), pooled_prompt_embeds=_bf(pooled_prompt_embeds, dev),
              height=height, width=width, num_inference_steps=num_inference_steps, guidance_scale=guidance_scale,
              latents=_bf(latents, dev), return_dict=False, true_cfg_scale=true_cfg_scale,
              negative_prompt_embeds=_bf(negative_prompt_embeds, dev) if do_true_cfg else None,
              negative_pooled_prompt_embeds=_bf(negative_pooled_prompt_embeds, dev) if do_true_cfg else None)
    if trace is not None:
        kw["trace"] = trace
    latents = eng(**_loop_extras(kw, sigmas, callback_on_step_end, callback_on_step_end_tensor_inputs))[0]
    clk.mark("loop_s")
    # 7. decode (inplace.py:396-410)
    if output_type == "latent":
        out = latents
    else:
        vae = host.vae
        lat = host._unpack_latents(latents.to(vae.dtype if hasattr(vae, "dtype") else latents.dtype), height, width,
                                   host.vae_scale_factor)
        lat = lat / vae.config.scaling_factor + vae.config.shift_factor
        out = host.image_processor.postprocess(_decode_image(host, vae, lat, dev), output_type=output_type)
    if hasattr(host, "maybe_free_model_hooks"):
        host.maybe_free_model_hooks()
    clk.mark("decode_s")
    return HostedOutput(out, clk.out) if return_dict else (out,)


class _HostConnector:
    """Step1X-Edit's text path in front of the trunk - the [EXT] Qwen2 `connector` (token refiner conditioned on the
    TIMESTEP, so it runs once per computed step) and, in v1p2, `text_token_mapping` - stays on the host transformer
    (Step1XEdit/inplace.py:514-516, Step1XEditV1P2/inplace.py:602-609); the engine's `connector` hook calls this object
    and gets (context tokens, pooled vector rows) back.  Masks / text embeddings are bound per CFG branch."""

    def __init__(self, host_tr, dev, masks, text=None):
        self.tr, self.dev, self.masks, self.text = host_tr, dev, masks, text

    def _one(self, enc, timestep, row):
        mask = self.masks[row]
        hdev = enc.device if mask is None else mask.device
        e, y = self.tr.connector(enc.to(hdev), timestep.to(hdev), mask)
        ttm = getattr(self.tr, "text_token_mapping", None)
        if ttm is not None and self.text is not None and self.text[row] is not None:       # v1p2 :606-609
            emb, tmask = self.text[row]
            e = e + ttm(emb) * tmask[:, :, None]
        return _bf(e, self.dev), _bf(y, self.dev)

    def __call__(self, encoder_hidden_states, timestep, prompt_embeds_mask=None, tag=None):
        if tag is not None:                                   # sequential CFG (v1p2): one branch per call
            return self._one(encoder_hidden_states, timestep, 0 if tag == "cond" else 1)
        outs = [self._one(encoder_hidden_states[b:b + 1], timestep[b:b + 1], b) for b in range(encoder_hidden_states.shape[0])]
        return torch.cat([o[0] for o in outs], 0), [o[1] for o in outs]                    # batched CFG (v1p1)


def _hosted_step1x(host, eng, image=None, prompt=None, negative_prompt=None, true_cfg_scale: float = 6.0, height=None, width=None,
                   num_inference_steps: int = 28, guidance_scale: float = 6.0, num_images_per_prompt: int = 1, generator=None,
                   latents=None, prompt_embeds=None, prompt_embeds_mask=None, negative_prompt_embeds=None,
                   negative_prompt_embeds_mask=None, output_type: str = "pil", return_dict: bool = True,
                   timesteps_truncate: float = 0.93, process_norm_power: float = 0.4, size_level=None, trace=None, sigmas=None,
                   callback_on_step_end=None, callback_on_step_end_tensor_inputs=("latents",), enable_thinking_mode=None,
                   enable_reflection_mode=None, max_try_cnt=None, **unused):
    """Step1XEditPipeline / Step1XEditPipelineV1P2 `__call__` around the engine loop (Step1XEdit/inplace.py:185-330,:437-455;
    Step1XEditV1P2/inplace.py:214-300).  Not hosted: v1p2's thinking / reflection retry loop (VLM prompting, :192-212) - the
    v1p2 switches `enable_thinking_mode` / `enable_reflection_mode` / `max_try_cnt` (Step1XEditV1P2/inplace.py:107-109) are
    accepted when they switch that loop OFF (the reference driver's own call, src/Step1X-Edit-v1p2/main.py:42-43,:69-77),
    True raises NotImplementedError; left unspecified the hosted call runs ONE attempt without reflection (the reference's
    signature default is reflection on: stated deviation)."""
    v1p2 = _host_name(host).endswith("V1P2")
    think = dict(enable_thinking_mode=enable_thinking_mode, enable_reflection_mode=enable_reflection_mode, max_try_cnt=max_try_cnt)
    if v1p2:
        for k in ("enable_thinking_mode", "enable_reflection_mode"):
            if think[k]:
                raise NotImplementedError(f"{_host_name(host)}.__call__: {k}=True (the VLM thinking / reflection retry loop, "
                                          "Step1XEditV1P2/inplace.py:192-212) is not hosted by the HIP loop; pass False")
        if enable_reflection_mode is None:
            # the reference's signature default is reflection ON (Step1XEditV1P2/inplace.py:108): a caller that leaves it unspecified
            # gets ONE attempt here - say so instead of deviating silently (advisor finding, round 4)
            import warnings
            warnings.warn(f"{_host_name(host)}.__call__: enable_reflection_mode left unspecified - the reference defaults to True (a VLM "
                          "reflection retry loop), the HIP-hosted call runs ONE attempt without reflection; pass "
                          "enable_reflection_mode=False to acknowledge", stacklevel=3)
    else:                                   # v1p1 has no such arguments: a caller passing them gets the usual refusal
        unused = dict(unused, **{k: v for k, v in think.items() if v is not None})
    _refuse_unused(_host_name(host) + ".__call__", unused)
    dev = eng.transformer.device
    clk = _Clock(dev)
    exec_dev = getattr(host, "_execution_device", dev)
    _one_image_only(prompt, num_images_per_prompt)
    # 1. image (host.encode_image resizes, keeps the reference image for the VLM, remembers how to undo the resize)
    args = (image, width, height, size_level, exec_dev, 1) if v1p2 else (image, width, height, exec_dev, 1)
    image, ref_image, img_info, width, height = host.encode_image(*args)
    # 2./3. prompts through the host's VLM encoder
    has_neg = negative_prompt is not None or (negative_prompt_embeds is not None and negative_prompt_embeds_mask is not None)
    if not has_neg:
        negative_prompt = "" if image is not None else STEP1X_DEFAULT_NEGATIVE
    if v1p2:          # encode_prompt returns a record: embedding / mask / txt_ids / text_embeds / text_masks (:241-254)
        pos = host.encode_prompt(ref_image=ref_image, prompt=prompt, device=exec_dev, num_images_per_prompt=1)
        neg = host.encode_prompt(ref_image=ref_image, prompt=negative_prompt, device=exec_dev, num_images_per_prompt=1)
        prompt_embeds, prompt_embeds_mask = pos.embedding, pos.mask
        negative_prompt_embeds, negative_prompt_embeds_mask = neg.embedding, neg.mask
        text = [(pos.text_embeds, pos.text_masks), (neg.text_embeds, neg.text_masks)]
        dtype = pos.embedding.dtype
    else:
        prompt_embeds, prompt_embeds_mask, _ = host.encode_prompt(
            ref_image=ref_image, prompt=prompt, prompt_embeds=prompt_embeds, prompt_embeds_mask=prompt_embeds_mask,
            device=exec_dev, num_images_per_prompt=1)
        negative_prompt_embeds, negative_prompt_embeds_mask, _ = host.encode_prompt(
            ref_image=ref_image, prompt=negative_prompt, prompt_embeds=negative_prompt_embeds,
            prompt_embeds_mask=negative_prompt_embeds_mask, device=exec_dev, num_images_per_prompt=1)
        text, dtype = None, prompt_embeds.dtype
    # 4. latents
    pl = (image, 1, eng.transformer.cfg_model.in_channels // 4, height, width, dtype, exec_dev, generator)
    with _hip_vae_encode(host, dev):
        latents, image_latents, _, _ = host.prepare_latents(*pl) if v1p2 else host.prepare_latents(*pl, latents)
    clk.mark("encode_s")
    # 5.-6. loop on the engine; the host's connector feeds it per computed step
    tr = eng.transformer
    prev = tr.__dict__.get("connector")
    tr.connector = _HostConnector(host.transformer, dev, [prompt_embeds_mask, negative_prompt_embeds_mask], text)
    try:
        kw = dict(image=_bf(image_latents, dev), prompt_embeds=prompt_embeds, negative_prompt_embeds=negative_prompt_embeds,
                  pooled_prompt_embeds=None, negative_pooled_prompt_embeds=None, height=height, width=width,
                  num_inference_steps=num_inference_steps, true_cfg_scale=true_cfg_scale, guidance_scale=guidance_scale,
                  latents=_bf(latents, dev), return_dict=False, timesteps_truncate=timesteps_truncate,
                  process_norm_power=process_norm_power)
        if trace is not None:
            kw["trace"] = trace
        latents = eng(**_loop_extras(kw, sigmas, callback_on_step_end, callback_on_step_end_tensor_inputs))[0]
    finally:
        if prev is None:
            tr.__dict__.pop("connector", None)
        else:
            tr.connector = prev
    clk.mark("loop_s")
    # 7. decode (:437-446)
    if output_type == "latent":
        out = latents
    else:
        vae = host.vae
        lat = host._unpack_latents(latents.to(getattr(vae, "dtype", latents.dtype)), height, width, host.vae_scale_factor)
        lat = lat / vae.config.scaling_factor + vae.config.shift_factor
        out = host.image_processor.postprocess(_decode_image(host, vae, lat, dev), output_type=output_type)
        out = host._output_process_image(out, img_info)
    if hasattr(host, "maybe_free_model_hooks"):
        host.maybe_free_model_hooks()
    clk.mark("decode_s")
    return HostedOutput(out, clk.out) if return_dict else (out,)


def _qwen_dims(target_area, ratio):
    """calculate_dimensions (QwenImageEdit/utils.py:96-103)."""
    import math
    width = math.sqrt(target_area * ratio)
    height = width / ratio
    return round(width / 32) * 32, round(height / 32) * 32


def _hosted_qwen(host, eng, image=None, prompt=None, negative_prompt=None, true_cfg_scale: float = 4.0, height=None, width=None,
                 num_inference_steps: int = 28, guidance_scale=None, num_images_per_prompt: int = 1, generator=None, latents=None,
                 prompt_embeds=None, prompt_embeds_mask=None, negative_prompt_embeds=None, negative_prompt_embeds_mask=None,
                 output_type: str = "pil", return_dict: bool = True, max_sequence_length: int = 512, trace=None, sigmas=None,
                 callback_on_step_end=None, callback_on_step_end_tensor_inputs=("latents",), **unused):
    """QwenImageEditPipeline / QwenImageEditPlusPipeline `__call__` around the engine loop (QwenImageEdit/inplace.py:180-330,
    :434-455; QwenImageEditPlus/inplace.py:189-300: a LIST of condition images, each resized twice - 384^2 area for the VLM,
    1024^2 area for the VAE - the last one fixing the output size)."""
    _refuse_unused(_host_name(host) + ".__call__", unused)
    plus = "Plus" in _host_name(host)
    dev = eng.transformer.device
    clk = _Clock(dev)
    exec_dev = getattr(host, "_execution_device", dev)
    _one_image_only(prompt, num_images_per_prompt)
    imgs = image if isinstance(image, list) else [image]
    ref = imgs[-1] if plus else imgs[0]
    ref_w, ref_h = ref.size if hasattr(ref, "size") and not isinstance(ref, torch.Tensor) else (ref.shape[-1], ref.shape[-2])
    calc_w, calc_h = _qwen_dims(1024 * 1024, ref_w / ref_h)
    height, width = height or calc_h, width or calc_w
    multiple_of = host.vae_scale_factor * 2
    width, height = width // multiple_of * multiple_of, height // multiple_of * multiple_of
    tok = lambda px: px // host.vae_scale_factor // 2
    is_latent = isinstance(image, torch.Tensor) and image.size(1) == getattr(host, "latent_channels", -1)
    if plus and not is_latent:
        prompt_image, vae_images, cond_shapes = [], [], []
        for img in imgs:
            iw, ih = img.size if hasattr(img, "size") and not isinstance(img, torch.Tensor) else (img.shape[-1], img.shape[-2])
            cw, ch = _qwen_dims(QWEN_CONDITION_IMAGE_SIZE, iw / ih)
            vw, vh = _qwen_dims(QWEN_VAE_IMAGE_SIZE, iw / ih)
            prompt_image.append(host.image_processor.resize(img, ch, cw))
            vae_images.append(host.image_processor.preprocess(img, vh, vw).unsqueeze(2))
            cond_shapes.append((tok(vh), tok(vw)))
        image = vae_images
    elif not is_latent:
        image = host.image_processor.resize(image, calc_h, calc_w)
        prompt_image = image
        image = host.image_processor.preprocess(image, calc_h, calc_w).unsqueeze(2)
        cond_shapes = [(tok(calc_h), tok(calc_w))]
    else:
        prompt_image, cond_shapes = None, None
    has_neg = negative_prompt is not None or (negative_prompt_embeds is not None and negative_prompt_embeds_mask is not None)
    do_true_cfg = true_cfg_scale > 1 and has_neg
    enc = lambda p, e, m: host.encode_prompt(image=prompt_image, prompt=p, prompt_embeds=e, prompt_embeds_mask=m, device=exec_dev,
                                             num_images_per_prompt=1, max_sequence_length=max_sequence_length)
    prompt_embeds, prompt_embeds_mask = enc(prompt, prompt_embeds, prompt_embeds_mask)
    if do_true_cfg:
        negative_prompt_embeds, negative_prompt_embeds_mask = enc(negative_prompt, negative_prompt_embeds, negative_prompt_embeds_mask)
    latents, image_latents = host.prepare_latents(image, 1, eng.transformer.cfg_model.in_channels // 4, height, width,
                                                  prompt_embeds.dtype, exec_dev, generator, latents)
    clk.mark("encode_s")
    kw = dict(image=_bf(image_latents, dev), prompt_embeds=_bf(prompt_embeds, dev),
              negative_prompt_embeds=_bf(negative_prompt_embeds, dev) if do_true_cfg else None, height=height, width=width,
              num_inference_steps=num_inference_steps, true_cfg_scale=true_cfg_scale, latents=_bf(latents, dev),
              return_dict=False, cond_shapes=cond_shapes)
    if trace is not None:
        kw["trace"] = trace
    latents = eng(**_loop_extras(kw, sigmas, callback_on_step_end, callback_on_step_end_tensor_inputs))[0]
    clk.mark("loop_s")
    if output_type == "latent":
        out = latents
    else:                                                                                  # QwenImageEdit/inplace.py:437-451
        vae = host.vae
        lat = host._unpack_latents(latents, height, width, host.vae_scale_factor).to(vae.dtype)
        mean = torch.tensor(vae.config.latents_mean).view(1, vae.config.z_dim, 1, 1, 1).to(lat.device, lat.dtype)
        inv_std = 1.0 / torch.tensor(vae.config.latents_std).view(1, vae.config.z_dim, 1, 1, 1).to(lat.device, lat.dtype)
        out = host.image_processor.postprocess(vae.decode(lat / inv_std + mean, return_dict=False)[0][:, :, 0], output_type=output_type)
    if hasattr(host, "maybe_free_model_hooks"):
        host.maybe_free_model_hooks()
    clk.mark("decode_s")
    return HostedOutput(out, clk.out) if return_dict else (out,)


_HOSTED = {"FluxKontextPipeline": _hosted_flux, "Step1XEditPipeline": _hosted_step1x, "Step1XEditPipelineV1P2": _hosted_step1x,
           "QwenImageEditPipeline": _hosted_qwen, "QwenImageEditPlusPipeline": _hosted_qwen}


def is_engine_pipeline(pipe) -> bool:
    """True for a regione_amd.harness pipeline (latent-level, HIP transformer); False for a stock host pipeline."""
    from .harness import flux as HF
    return isinstance(pipe, HF.FluxKontextPipeline) or isinstance(getattr(pipe, "transformer", None), HF.FluxTransformer2DModel)


class HostedPipeline:
    """Wrapper object: `hosted(image=..., prompt=...)` = host pre / post-processing + engine loop.  `RegionEHelper(hosted)`
    patches the engine.  (RegionEHelper(pipe) on the stock pipeline itself is the reference's call shape - see attach.)"""

    def __init__(self, host, engine):
        self.host, self._regione_engine = host, engine
        self._call = _HOSTED[_host_name(host)]

    @property
    def engine(self):
        return self._regione_engine

    @torch.no_grad()
    def __call__(self, *a, **kw):
        if a:
            raise TypeError("pass the pipeline arguments by keyword (image=..., prompt=...)")
        return self._call(self.host, self._regione_engine, **kw)


HostedFluxKontextPipeline = HostedPipeline          # round-1 name


def _host_name(pipe) -> str:
    cls = getattr(pipe, "_regione_host_class", None) or pipe.__class__
    return cls.__name__


def adopt(pipe, device="cuda"):
    """Hosted wrapper (image + prompt in, image out) for any of the five pipeline classes."""
    if _host_name(pipe) not in _HOSTED:
        raise NotImplementedError(f"no hosted call for {pipe.__class__.__name__}")
    return HostedPipeline(pipe, attach(pipe, device))


def attach(pipe, device="cuda"):
    """Adopt the host pipeline's transformer once and keep the engine ON the host object (`pipe._regione_engine`)."""
    eng = pipe.__dict__.get("_regione_engine") if hasattr(pipe, "__dict__") else None
    if eng is None:
        eng = adopt_engine(pipe, device)
        pipe._regione_engine = eng
        # the VAE too, HERE rather than inside the first edit (weights are re-laid once: ~0.5 s that would otherwise sit in that edit's
        # decode_s / encode_s); a VAE of another layout returns None and keeps running on the host module
        if getattr(pipe, "vae", None) is not None:
            dev = eng.transformer.device
            hip_vae_for(pipe, dev)
            hip_vae_encoder_for(pipe, dev)
    return eng


def swap_host_class(pipe):
    """Hook (1) of the reference (`pipeline.__class__ = RegionE<...>Pipeline`, inplace.py:53-62) on the HOST object: a
    subclass of the host's own class whose `__call__` is the hosted call.  The user keeps calling `pipe(...)`."""
    if getattr(pipe, "_regione_host_class", None) is not None:
        return pipe
    base = pipe.__class__
    call = _HOSTED[base.__name__]

    @torch.no_grad()
    def __call__(self, *a, **kw):
        if a:
            raise TypeError("pass the pipeline arguments by keyword (image=..., prompt=...)")
        return call(self, self._regione_engine, **kw)
    pipe._regione_host_class = base
    pipe.__class__ = type("RegionE" + base.__name__, (base,), {"__call__": __call__})
    return pipe


def restore_host_class(pipe):
    base = getattr(pipe, "_regione_host_class", None)
    if base is not None:
        pipe.__class__ = base
        pipe._regione_host_class = None
    return pipe
