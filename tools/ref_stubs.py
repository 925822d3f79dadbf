"""Build-container-only harness that lets /root/reference's RegionE package be imported on CPU.

TEST INFRASTRUCTURE, never shipped to the GPU box and never imported by product code.

The reference (Peyton-Chen/RegionE) is pure Python on top of an un-vendored `diffusers` fork
(README.md:77) and `flash_attn` (README.md:80); neither is installed in this image and there is
no network.  Following SURVEY.md section 8c / Appendix C we pre-seed `sys.modules` with a stub
`diffusers` tree so that `RegionE/<Family>/{utils,inplace}.py` import, and patch two names:

  * `inplace.flash_attn = None`  -> reaches the SDPA branch (reference quirk A-1,
    RegionE/FluxKontext/inplace.py:39-43,796-806);
  * `inplace._partially_linear`  -> CPU index-linear with the Triton kernel's fp16 round trip
    (RegionE/FluxKontext/fused_kernels.py:77-80); Triton cannot launch without a GPU.

Everything RegionE itself authored (token_selector, morphology, ids_gather/scatter, the Manager,
the scheduler step, the attention-processor K/V-cache protocol, the denoise loop with the AVD
decision, the dual-RoPE transformer forward) then runs unmodified.

The pieces marked [EXT] below restate *upstream diffusers semantics* (the MMDiT block bodies,
RMSNorm, AdaLN, RoPE, the scheduler base).  They are NOT reference code: the reference only calls
them.  Fixtures that depend on them are labelled "parity unpinned for block arithmetic" in
DESIGN.md; what they do pin is the RegionE-authored logic wrapped around them.
"""
from __future__ import annotations

import importlib
import math
import sys
import types
from contextlib import contextmanager

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

REF_ROOT = "/root/reference"


import os as _os

sys.path.insert(0, _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "tests"))
# the [EXT] module trees live in tests/host_trunks.py (shipped with the tests: the host stand-ins use them too); this file
# is the import harness around them and stays in the build container
from host_trunks import (  # noqa: E402,F401
    AdaLayerNormContinuous,
    AdaLayerNormZero,
    AdaLayerNormZeroSingle,
    Attention,
    CombinedTimestepGuidanceTextProjEmbeddings,
    FeedForward,
    FluxPosEmbed,
    FluxSingleTransformerBlock,
    FluxTransformer2DModel,
    FluxTransformerBlock,
    QwenEmbedRope,
    QwenImageTransformer2DModel,
    QwenImageTransformerBlock,
    QwenTimestepProjEmbeddings,
    RMSNorm,
    Step1XEditTransformer2DModel,
    _Cfg,
    _GELUProj,
    _MLPEmbed,
    apply_rotary_emb,
    get_1d_rotary_pos_embed,
    get_timestep_embedding,
)


class FluxAttnProcessor:  # placeholder so unwarp_modules can construct one
    pass


class FlowMatchEulerDiscreteScheduler:
    """[EXT] minimal restatement of diffusers FlowMatchEulerDiscreteScheduler with
    use_dynamic_shifting=True / time_shift_type='exponential' (FLUX & Step1X scheduler config)."""

    order = 1

    def __init__(self, **config):
        cfg = dict(num_train_timesteps=1000, shift=3.0, use_dynamic_shifting=True, base_shift=0.5,
                   max_shift=1.15, base_image_seq_len=256, max_image_seq_len=4096,
                   stochastic_sampling=False)
        cfg.update(config)
        self.config = _Cfg(cfg)
        self.sigmas = None
        self.timesteps = None
        self._step_index = None
        self._begin_index = None

    @classmethod
    def from_config(cls, config):
        return cls(**dict(config))

    @property
    def step_index(self):
        return self._step_index

    def set_begin_index(self, begin_index=0):
        self._begin_index = begin_index

    def set_timesteps(self, num_inference_steps=None, device=None, sigmas=None, mu=None, timesteps=None):
        sigmas = np.array(sigmas).astype(np.float32)
        if self.config.use_dynamic_shifting:
            sigmas = math.exp(mu) / (math.exp(mu) + (1 / sigmas - 1) ** 1.0)
        else:
            s = self.config.shift
            sigmas = s * sigmas / (1 + (s - 1) * sigmas)
        sigmas = torch.from_numpy(np.asarray(sigmas, dtype=np.float32)).to(dtype=torch.float32, device=device)
        self.timesteps = sigmas * self.config.num_train_timesteps
        self.sigmas = torch.cat([sigmas, torch.zeros(1, device=sigmas.device)])
        self.num_inference_steps = len(self.timesteps)
        self._step_index = None

    def _init_step_index(self, timestep):
        if self._begin_index is None:
            idx = (self.timesteps == timestep).nonzero()
            pos = 1 if len(idx) > 1 else 0
            self._step_index = idx[pos].item()
        else:
            self._step_index = self._begin_index


class _BaseOutput(dict):
    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.__dict__.update(k)


class _FakePipelineBase:
    """[EXT] the few DiffusionPipeline attributes the reference's __call__ touches
    (RegionE/FluxKontext/inplace.py:112-410)."""

    latent_channels = 16
    vae_scale_factor = 8
    default_sample_size = 128
    _execution_device = torch.device("cpu")

    @property
    def joint_attention_kwargs(self):
        return self._joint_attention_kwargs

    @property
    def interrupt(self):
        return self._interrupt

    @property
    def guidance_scale(self):
        return self._guidance_scale

    def check_inputs(self, *a, **k):
        return None

    def maybe_free_model_hooks(self):
        return None

    @contextmanager
    def progress_bar(self, total=None):
        class _PB:
            def update(self_inner, *a):
                return None
        yield _PB()


def apply_rotary_emb_qwen(x, freqs_cis, use_real=True, use_real_unbind_dim=-1):
    """[EXT] transformer_qwenimage.apply_rotary_emb_qwen, use_real=False (complex) branch.
    x [B, S, H, D], freqs_cis complex [S, D/2]."""
    x_rotated = torch.view_as_complex(x.float().reshape(*x.shape[:-1], -1, 2))
    freqs_cis = freqs_cis.unsqueeze(1)
    x_out = torch.view_as_real(x_rotated * freqs_cis).flatten(3)
    return x_out.type_as(x)


class _FakeQwenBase(_FakePipelineBase):
    """[EXT] QwenImageEditPipeline members the reference loop touches (QwenImageEdit/inplace.py:180-420)."""

    @property
    def attention_kwargs(self):
        return self._attention_kwargs


class _FakeStep1XBase(_FakePipelineBase):
    """[EXT] Step1XEditPipeline members the reference loop touches (Step1XEdit/inplace.py:186-460)."""

    def process_diff_norm(self, diff_norm, k):
        """[EXT] Step1X-Edit sampling: norm > 1 -> norm^k, norm < 1 -> 1, else norm."""
        pow_result = torch.pow(diff_norm, k)
        return torch.where(diff_norm > 1.0, pow_result,
                           torch.where(diff_norm < 1.0, torch.ones_like(diff_norm), diff_norm))

    def _output_process_image(self, image, img_info):
        return image


# --------------------------------------------------------------------------------------
# stub installation + reference import
# --------------------------------------------------------------------------------------
class _Any(types.ModuleType):
    def __getattr__(self, n):
        if n.startswith("__"):
            raise AttributeError(n)
        t = type(n, (), {})
        setattr(self, n, t)
        return t


_STUB_NAMES = [
    "diffusers", "diffusers.image_processor", "diffusers.utils", "diffusers.schedulers",
    "diffusers.pipelines", "diffusers.pipelines.flux", "diffusers.pipelines.step1x_edit",
    "diffusers.pipelines.qwenimage", "diffusers.models", "diffusers.models.embeddings",
    "diffusers.models.attention_processor", "diffusers.models.modeling_outputs",
    "diffusers.models.transformers", "diffusers.models.transformers.transformer_flux",
    "diffusers.models.transformers.transformer_step1x_edit",
    "diffusers.models.transformers.transformer_qwenimage",
    "diffusers.pipelines.step1x_edit.pipeline_step1x_edit_thinker",
]


def partially_linear_cpu(x, W, b, idx, out):
    """CPU stand-in for the Triton launch `_partially_linear` (fused_kernels.py:81-101):
    fp32 accumulate, + bias, **round through fp16** (fused_kernels.py:80), store as cache dtype
    at rows idx."""
    y = F.linear(x.float(), W.float(), None if b is None else b.float())
    out[:, idx] = y.to(torch.float16).to(out.dtype)


def install():
    """Install the stub tree and import the reference. Returns a namespace of reference modules."""
    if "RegionE" in sys.modules and getattr(sys.modules["RegionE"], "_regione_ref", False):
        return sys.modules["RegionE"]._ns
    for name in _STUB_NAMES:
        m = _Any(name)
        m.__path__ = []
        sys.modules[name] = m
    du = sys.modules["diffusers.utils"]
    du.BaseOutput = _BaseOutput
    du.is_torch_xla_available = lambda: False
    du.USE_PEFT_BACKEND = False

    class _Log:
        @staticmethod
        def get_logger(name):
            import logging as _l
            return _l.getLogger(name)
    du.logging = _Log
    du.scale_lora_layers = lambda *a, **k: None
    du.unscale_lora_layers = lambda *a, **k: None
    sys.modules["diffusers.models.embeddings"].apply_rotary_emb = apply_rotary_emb
    sys.modules["diffusers.models.attention_processor"].Attention = Attention
    sys.modules["diffusers.schedulers"].FlowMatchEulerDiscreteScheduler = FlowMatchEulerDiscreteScheduler
    tf = sys.modules["diffusers.models.transformers.transformer_flux"]
    tf.FluxAttnProcessor = FluxAttnProcessor
    tf.FluxTransformer2DModel = FluxTransformer2DModel
    sys.modules["diffusers"].FluxKontextPipeline = type("FluxKontextPipeline", (_FakePipelineBase,), {})
    sys.modules["diffusers.pipelines.flux"].FluxPipelineOutput = _BaseOutput
    # Step1X-Edit (v1p1): same trunk; pipeline base needs process_diff_norm [EXT]
    sys.modules["diffusers"].Step1XEditPipeline = type("Step1XEditPipeline", (_FakeStep1XBase,), {})
    sys.modules["diffusers.pipelines.step1x_edit"].Step1XEditPipelineOutput = _BaseOutput
    sys.modules["diffusers"].Step1XEditPipelineV1P2 = type("Step1XEditPipelineV1P2", (_FakeStep1XBase,), {})
    ts = sys.modules["diffusers.models.transformers.transformer_step1x_edit"]
    ts.Step1XEditAttention = Attention
    ts.Step1XEditAttnProcessor = FluxAttnProcessor
    ts.Step1XEditTransformer2DModel = FluxTransformer2DModel

    # Qwen-Image-Edit
    tq = sys.modules["diffusers.models.transformers.transformer_qwenimage"]
    tq.apply_rotary_emb_qwen = apply_rotary_emb_qwen
    tq.QwenImageTransformer2DModel = FluxTransformer2DModel
    tq.QwenDoubleStreamAttnProcessor2_0 = FluxAttnProcessor
    sys.modules["diffusers"].QwenImageEditPipeline = type("QwenImageEditPipeline", (_FakeQwenBase,), {})
    sys.modules["diffusers"].QwenImageEditPlusPipeline = type("QwenImageEditPlusPipeline", (_FakeQwenBase,), {})
    sys.modules["diffusers.pipelines.qwenimage"].QwenImagePipelineOutput = _BaseOutput

    pkg = types.ModuleType("RegionE")
    pkg.__path__ = [REF_ROOT + "/RegionE"]
    pkg._regione_ref = True
    sys.modules["RegionE"] = pkg
    ns = types.SimpleNamespace()
    ns.flux = importlib.import_module("RegionE.FluxKontext.inplace")
    ns.flux_utils = importlib.import_module("RegionE.FluxKontext.utils")
    ns.flux.flash_attn = None
    ns.flux._partially_linear = partially_linear_cpu
    ns.step1x = importlib.import_module("RegionE.Step1XEdit.inplace")
    ns.step1x_utils = importlib.import_module("RegionE.Step1XEdit.utils")
    ns.step1x.flash_attn = None
    ns.step1x._partially_linear = partially_linear_cpu
    ns.qwen = importlib.import_module("RegionE.QwenImageEdit.inplace")
    ns.qwen.flash_attn = None
    ns.qwen._partially_linear = partially_linear_cpu
    ns.step1x_v1p2 = importlib.import_module("RegionE.Step1XEditV1P2.inplace")
    ns.step1x_v1p2.flash_attn = None
    ns.step1x_v1p2._partially_linear = partially_linear_cpu
    pkg._ns = ns
    return ns
