"""Host-side stand-in trunks: the torch.nn module trees (host parameter naming) of the transformers the reference patches,
WITH their own vanilla CPU forwards.

[EXT] Everything here restates upstream diffusers semantics (FluxTransformer2DModel, the diffusers-fork
Step1XEditTransformer2DModel, QwenImageTransformer2DModel and their vanilla attention processors); none of it is reference
code - the reference only *calls* these modules and replaces their forwards (RegionE/FluxKontext/inplace.py:53-62).

Two users:
  * tests/ (host_standins.py, test_adapters.py, test_hosted_pipelines.py): the trunk a stock pipeline object carries; its
    OWN forward (below, torch-CPU eager) is the independent implementation the adopted HIP engine is compared with;
  * tools/ref_stubs.py (build container only): the same module trees host the reference's patched forwards when
    tools/gen_golden.py generates the fixtures.  Module construction order is part of the fixtures' seeds - do not reorder
    the parameter-creating statements of the classes shared with it.
Test infrastructure only; never imported by regione_amd/.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn
import torch.nn.functional as F

# --------------------------------------------------------------------------------------
# [EXT] diffusers restatements (upstream semantics, used only to host the reference on CPU)
# --------------------------------------------------------------------------------------
def apply_rotary_emb(x, freqs_cis, use_real=True, use_real_unbind_dim=-1, sequence_dim=2):
    """[EXT] diffusers.models.embeddings.apply_rotary_emb, use_real / unbind_dim=-1 branch.
    x: [B, H, S, D]; freqs_cis = (cos, sin) each [S, D]."""
    cos, sin = freqs_cis
    cos = cos[None, None].to(x.device)
    sin = sin[None, None].to(x.device)
    x_real, x_imag = x.reshape(*x.shape[:-1], -1, 2).unbind(-1)
    x_rotated = torch.stack([-x_imag, x_real], dim=-1).flatten(3)
    return (x.float() * cos + x_rotated.float() * sin).to(x.dtype)


def get_1d_rotary_pos_embed(dim, pos, theta=10000.0):
    """[EXT] repeat_interleave_real=True, use_real=True, freqs_dtype=float64 branch."""
    freqs = 1.0 / (theta ** (torch.arange(0, dim, 2, dtype=torch.float64)[: dim // 2] / dim))
    freqs = torch.outer(pos.to(torch.float64), freqs)
    cos = freqs.cos().repeat_interleave(2, dim=1).float()
    sin = freqs.sin().repeat_interleave(2, dim=1).float()
    return cos, sin


class FluxPosEmbed(nn.Module):
    """[EXT] diffusers FluxPosEmbed: ids [S, n_axes] -> (cos, sin) [S, sum(axes_dim)]."""

    def __init__(self, theta=10000, axes_dim=(16, 56, 56)):
        super().__init__()
        self.theta = theta
        self.axes_dim = tuple(axes_dim)

    def forward(self, ids):
        n_axes = ids.shape[-1]
        cos_out, sin_out = [], []
        pos = ids.float()
        for i in range(n_axes):
            cos, sin = get_1d_rotary_pos_embed(self.axes_dim[i], pos[:, i], theta=self.theta)
            cos_out.append(cos)
            sin_out.append(sin)
        return torch.cat(cos_out, dim=-1).to(ids.device), torch.cat(sin_out, dim=-1).to(ids.device)


class RMSNorm(nn.Module):
    """[EXT] diffusers.models.normalization.RMSNorm (elementwise_affine=True, no bias)."""

    def __init__(self, dim, eps=1e-6):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(dim))

    def forward(self, hidden_states):
        input_dtype = hidden_states.dtype
        variance = hidden_states.to(torch.float32).pow(2).mean(-1, keepdim=True)
        hidden_states = hidden_states * torch.rsqrt(variance + self.eps)
        if self.weight.dtype in (torch.float16, torch.bfloat16):
            hidden_states = hidden_states.to(self.weight.dtype)
        hidden_states = hidden_states * self.weight
        return hidden_states.to(input_dtype) if self.weight.dtype == input_dtype else hidden_states


class AdaLayerNormZero(nn.Module):
    """[EXT] 6-way AdaLN-Zero used by FluxTransformerBlock."""

    def __init__(self, dim):
        super().__init__()
        self.silu = nn.SiLU()
        self.linear = nn.Linear(dim, 6 * dim, bias=True)
        self.norm = nn.LayerNorm(dim, elementwise_affine=False, eps=1e-6)

    def forward(self, x, emb=None):
        emb = self.linear(self.silu(emb))
        shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp = emb.chunk(6, dim=1)
        x = self.norm(x) * (1 + scale_msa[:, None]) + shift_msa[:, None]
        return x, gate_msa, shift_mlp, scale_mlp, gate_mlp


class AdaLayerNormZeroSingle(nn.Module):
    """[EXT] 3-way AdaLN-Zero used by FluxSingleTransformerBlock."""

    def __init__(self, dim):
        super().__init__()
        self.silu = nn.SiLU()
        self.linear = nn.Linear(dim, 3 * dim, bias=True)
        self.norm = nn.LayerNorm(dim, elementwise_affine=False, eps=1e-6)

    def forward(self, x, emb=None):
        emb = self.linear(self.silu(emb))
        shift_msa, scale_msa, gate_msa = emb.chunk(3, dim=1)
        x = self.norm(x) * (1 + scale_msa[:, None]) + shift_msa[:, None]
        return x, gate_msa


class AdaLayerNormContinuous(nn.Module):
    """[EXT] norm_out of FluxTransformer2DModel (elementwise_affine=False, eps=1e-6)."""

    def __init__(self, dim, cond_dim):
        super().__init__()
        self.silu = nn.SiLU()
        self.linear = nn.Linear(cond_dim, 2 * dim, bias=True)
        self.norm = nn.LayerNorm(dim, elementwise_affine=False, eps=1e-6)

    def forward(self, x, conditioning_embedding):
        emb = self.linear(self.silu(conditioning_embedding).to(x.dtype))
        scale, shift = torch.chunk(emb, 2, dim=1)
        return self.norm(x) * (1 + scale)[:, None, :] + shift[:, None, :]


class _GELUProj(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out, bias=True)

    def forward(self, x):
        return F.gelu(self.proj(x), approximate="tanh")


class FeedForward(nn.Module):
    """[EXT] diffusers FeedForward(activation_fn='gelu-approximate', mult=4)."""

    def __init__(self, dim, mult=4):
        super().__init__()
        self.net = nn.ModuleList([_GELUProj(dim, dim * mult), nn.Dropout(0.0), nn.Linear(dim * mult, dim)])

    def forward(self, x):
        for m in self.net:
            x = m(x)
        return x


class Attention(nn.Module):
    """[EXT] the attribute container the reference's processors expect
    (RegionE/FluxKontext/inplace.py:715-820) plus set_processor / forward dispatch."""

    def __init__(self, dim, heads, head_dim, added_kv=False, pre_only=False, processor=None):
        super().__init__()
        self.heads = heads
        inner = heads * head_dim
        self.to_q = nn.Linear(dim, inner, bias=True)
        self.to_k = nn.Linear(dim, inner, bias=True)
        self.to_v = nn.Linear(dim, inner, bias=True)
        self.norm_q = RMSNorm(head_dim, eps=1e-6)
        self.norm_k = RMSNorm(head_dim, eps=1e-6)
        if added_kv:
            self.add_q_proj = nn.Linear(dim, inner, bias=True)
            self.add_k_proj = nn.Linear(dim, inner, bias=True)
            self.add_v_proj = nn.Linear(dim, inner, bias=True)
            self.norm_added_q = RMSNorm(head_dim, eps=1e-6)
            self.norm_added_k = RMSNorm(head_dim, eps=1e-6)
            self.to_add_out = nn.Linear(inner, dim, bias=True)
        if not pre_only:
            self.to_out = nn.ModuleList([nn.Linear(inner, dim, bias=True), nn.Dropout(0.0)])
        self.processor = processor

    def set_processor(self, processor):
        self.processor = processor

    def forward(self, hidden_states, encoder_hidden_states=None, attention_mask=None, **kw):
        return self.processor(self, hidden_states, encoder_hidden_states=encoder_hidden_states,
                              attention_mask=attention_mask, **kw)


class FluxTransformerBlock(nn.Module):
    """[EXT] diffusers FluxTransformerBlock (double stream)."""

    def __init__(self, dim, heads, head_dim):
        super().__init__()
        self.norm1 = AdaLayerNormZero(dim)
        self.norm1_context = AdaLayerNormZero(dim)
        self.attn = Attention(dim, heads, head_dim, added_kv=True)
        self.norm2 = nn.LayerNorm(dim, elementwise_affine=False, eps=1e-6)
        self.ff = FeedForward(dim)
        self.norm2_context = nn.LayerNorm(dim, elementwise_affine=False, eps=1e-6)
        self.ff_context = FeedForward(dim)

    def forward(self, hidden_states, encoder_hidden_states, temb, image_rotary_emb=None, joint_attention_kwargs=None):
        norm_h, gate_msa, shift_mlp, scale_mlp, gate_mlp = self.norm1(hidden_states, emb=temb)
        norm_c, c_gate_msa, c_shift_mlp, c_scale_mlp, c_gate_mlp = self.norm1_context(encoder_hidden_states, emb=temb)
        attn_output, context_attn_output = self.attn(
            hidden_states=norm_h, encoder_hidden_states=norm_c, image_rotary_emb=image_rotary_emb,
            **(joint_attention_kwargs or {}))
        attn_output = gate_msa.unsqueeze(1) * attn_output
        hidden_states = hidden_states + attn_output
        norm_h = self.norm2(hidden_states)
        norm_h = norm_h * (1 + scale_mlp[:, None]) + shift_mlp[:, None]
        ff_output = self.ff(norm_h)
        ff_output = gate_mlp.unsqueeze(1) * ff_output
        hidden_states = hidden_states + ff_output
        context_attn_output = c_gate_msa.unsqueeze(1) * context_attn_output
        encoder_hidden_states = encoder_hidden_states + context_attn_output
        norm_c = self.norm2_context(encoder_hidden_states)
        norm_c = norm_c * (1 + c_scale_mlp[:, None]) + c_shift_mlp[:, None]
        context_ff_output = self.ff_context(norm_c)
        encoder_hidden_states = encoder_hidden_states + c_gate_mlp.unsqueeze(1) * context_ff_output
        return encoder_hidden_states, hidden_states


class FluxSingleTransformerBlock(nn.Module):
    """[EXT] diffusers FluxSingleTransformerBlock (takes the two streams, concatenates inside)."""

    def __init__(self, dim, heads, head_dim, mlp_ratio=4.0):
        super().__init__()
        self.mlp_hidden_dim = int(dim * mlp_ratio)
        self.norm = AdaLayerNormZeroSingle(dim)
        self.proj_mlp = nn.Linear(dim, self.mlp_hidden_dim)
        self.proj_out = nn.Linear(dim + self.mlp_hidden_dim, dim)
        self.attn = Attention(dim, heads, head_dim, added_kv=False, pre_only=True)

    def forward(self, hidden_states, encoder_hidden_states, temb, image_rotary_emb=None, joint_attention_kwargs=None):
        text_seq_len = encoder_hidden_states.shape[1]
        hidden_states = torch.cat([encoder_hidden_states, hidden_states], dim=1)
        residual = hidden_states
        norm_h, gate = self.norm(hidden_states, emb=temb)
        mlp_h = F.gelu(self.proj_mlp(norm_h), approximate="tanh")
        attn_output = self.attn(hidden_states=norm_h, image_rotary_emb=image_rotary_emb, **(joint_attention_kwargs or {}))
        hidden_states = torch.cat([attn_output, mlp_h], dim=2)
        hidden_states = gate.unsqueeze(1) * self.proj_out(hidden_states)
        hidden_states = residual + hidden_states
        return hidden_states[:, :text_seq_len], hidden_states[:, text_seq_len:]


def get_timestep_embedding(timesteps, embedding_dim=256, max_period=10000, scale=1.0):
    """[EXT] flip_sin_to_cos=True, downscale_freq_shift=0; `scale` multiplies the ANGLES (diffusers: emb = scale * emb)."""
    half = embedding_dim // 2
    exponent = -math.log(max_period) * torch.arange(0, half, dtype=torch.float32)
    exponent = exponent / half
    emb = torch.exp(exponent)
    emb = timesteps[:, None].float() * emb[None, :]
    if scale != 1.0:
        emb = scale * emb
    emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=-1)
    return torch.cat([emb[:, half:], emb[:, :half]], dim=-1)


class _MLPEmbed(nn.Module):
    def __init__(self, d_in, d):
        super().__init__()
        self.linear_1 = nn.Linear(d_in, d)
        self.linear_2 = nn.Linear(d, d)

    def forward(self, x):
        return self.linear_2(F.silu(self.linear_1(x)))


class CombinedTimestepGuidanceTextProjEmbeddings(nn.Module):
    """[EXT] FLUX time_text_embed (guidance_embeds=True)."""

    def __init__(self, d, pooled_dim):
        super().__init__()
        self.timestep_embedder = _MLPEmbed(256, d)
        self.guidance_embedder = _MLPEmbed(256, d)
        self.text_embedder = _MLPEmbed(pooled_dim, d)

    def forward(self, timestep, guidance, pooled_projection):
        t = self.timestep_embedder(get_timestep_embedding(timestep).to(pooled_projection.dtype))
        g = self.guidance_embedder(get_timestep_embedding(guidance).to(pooled_projection.dtype))
        return t + g + self.text_embedder(pooled_projection)


class _Cfg(dict):
    __getattr__ = dict.__getitem__


class FluxTransformer2DModel(nn.Module):
    """[EXT] module tree of diffusers FluxTransformer2DModel; forward is replaced by the reference."""

    def __init__(self, in_channels=64, n_double=2, n_single=2, heads=2, head_dim=128, joint_dim=256,
                 pooled_dim=64, axes_dim=(16, 56, 56)):
        super().__init__()
        d = heads * head_dim
        self.config = _Cfg(in_channels=in_channels, guidance_embeds=True)
        self.gradient_checkpointing = False
        self.pos_embed = FluxPosEmbed(theta=10000, axes_dim=axes_dim)
        self.time_text_embed = CombinedTimestepGuidanceTextProjEmbeddings(d, pooled_dim)
        self.context_embedder = nn.Linear(joint_dim, d)
        self.x_embedder = nn.Linear(in_channels, d)
        self.transformer_blocks = nn.ModuleList([FluxTransformerBlock(d, heads, head_dim) for _ in range(n_double)])
        self.single_transformer_blocks = nn.ModuleList(
            [FluxSingleTransformerBlock(d, heads, head_dim) for _ in range(n_single)])
        self.norm_out = AdaLayerNormContinuous(d, d)
        self.proj_out = nn.Linear(d, in_channels, bias=True)

    def forward(self, hidden_states, encoder_hidden_states=None, pooled_projections=None, timestep=None, img_ids=None,
                txt_ids=None, guidance=None, joint_attention_kwargs=None, return_dict=True):
        """[EXT] vanilla FluxTransformer2DModel.forward (the reference rebinds it; the host tests call it as is)."""
        hidden_states = self.x_embedder(hidden_states)
        timestep = timestep.to(hidden_states.dtype) * 1000
        guidance = guidance.to(hidden_states.dtype) * 1000
        temb = self.time_text_embed(timestep, guidance, pooled_projections)
        encoder_hidden_states = self.context_embedder(encoder_hidden_states)
        image_rotary_emb = self.pos_embed(torch.cat((txt_ids, img_ids), dim=0))
        for block in list(self.transformer_blocks) + list(self.single_transformer_blocks):
            encoder_hidden_states, hidden_states = block(hidden_states=hidden_states, encoder_hidden_states=encoder_hidden_states,
                                                         temb=temb, image_rotary_emb=image_rotary_emb,
                                                         joint_attention_kwargs=joint_attention_kwargs)
        output = self.proj_out(self.norm_out(hidden_states, temb))
        return (output,) if not return_dict else _Cfg(sample=output)


# ----- Step1X-Edit [EXT] stubs (FLUX trunk; diffusers-fork transformer_step1x_edit.py semantics, restated) -----
class Step1XEditTransformer2DModel(nn.Module):
    """[EXT] module tree the reference's Step1X forwards touch (Step1XEdit/inplace.py:514-522,
    Step1XEditV1P2/inplace.py:602-621): connector -> (encoder states, pooled y), x_embedder, time_proj / time_embed,
    vec_embed, context_embedder, pos_embed, FLUX-shaped double / single blocks, norm_out, proj_out.  The real
    connector is a Qwen2-VL adapter; here it hands through the prompt embeddings and returns the pooled vector that
    was registered for that prompt (`set_vec`), which is what the engine takes as an input too."""

    def __init__(self, in_channels=64, n_double=2, n_single=2, heads=2, head_dim=128, joint_dim=256,
                 pooled_dim=64, axes_dim=(16, 56, 56)):
        super().__init__()
        d = heads * head_dim
        self.config = _Cfg(in_channels=in_channels, guidance_embeds=False)
        self.gradient_checkpointing = False
        self.text_token_mapping = None
        self.pos_embed = FluxPosEmbed(theta=10000, axes_dim=axes_dim)
        self.time_embed = _MLPEmbed(256, d)
        self.vec_embed = _MLPEmbed(pooled_dim, d)
        self.context_embedder = nn.Linear(joint_dim, d)
        self.x_embedder = nn.Linear(in_channels, d)
        self.transformer_blocks = nn.ModuleList([FluxTransformerBlock(d, heads, head_dim) for _ in range(n_double)])
        self.single_transformer_blocks = nn.ModuleList(
            [FluxSingleTransformerBlock(d, heads, head_dim) for _ in range(n_single)])
        for b in list(self.transformer_blocks) + list(self.single_transformer_blocks):
            b.attn.added_kv_proj_dim = d if hasattr(b.attn, "add_q_proj") else None
        self.norm_out = AdaLayerNormContinuous(d, d)
        self.proj_out = nn.Linear(d, in_channels, bias=True)
        self._vec = {}

    def set_vec(self, prompt_embeds, y):
        self._vec[float(prompt_embeds.float().sum())] = y

    def time_proj(self, timestep):
        return get_timestep_embedding(timestep, 256)

    def connector(self, encoder_hidden_states, timestep, mask):
        ys = [self._vec[float(encoder_hidden_states[b:b + 1].float().sum())] for b in range(encoder_hidden_states.shape[0])]
        return encoder_hidden_states, torch.cat(ys, 0)

    def forward(self, hidden_states, encoder_hidden_states=None, prompt_embeds_mask=None, timestep=None, img_ids=None,
                txt_ids=None, guidance=None, joint_attention_kwargs=None, return_dict=True):
        """[EXT] vanilla Step1XEditTransformer2DModel.forward of the diffusers fork (call order as the reference's patched
        copy shows it, Step1XEdit/inplace.py:514-522): connector -> (states, y); temb = time_embed(t * 1000) + vec_embed(y)."""
        encoder_hidden_states, y = self.connector(encoder_hidden_states, timestep, prompt_embeds_mask)
        hidden_states = self.x_embedder(hidden_states)
        temb = self.time_embed(self.time_proj(timestep * 1000).to(timestep)) + self.vec_embed(y)
        encoder_hidden_states = self.context_embedder(encoder_hidden_states)
        image_rotary_emb = self.pos_embed(torch.cat((txt_ids, img_ids), dim=0))
        for block in list(self.transformer_blocks) + list(self.single_transformer_blocks):
            encoder_hidden_states, hidden_states = block(hidden_states=hidden_states, encoder_hidden_states=encoder_hidden_states,
                                                         temb=temb, image_rotary_emb=image_rotary_emb,
                                                         joint_attention_kwargs=joint_attention_kwargs)
        output = self.proj_out(self.norm_out(hidden_states, temb))
        return (output,) if not return_dict else _Cfg(sample=output)


# ----- Qwen-Image [EXT] stubs (diffusers transformer_qwenimage.py semantics, restated) -----------------
class QwenImageTransformerBlock(FluxTransformerBlock):
    """[EXT] QwenImageTransformerBlock: img_mod / txt_mod = SiLU + Linear(d, 6d) chunked (shift1, scale1, gate1,
    shift2, scale2, gate2), LayerNorm(no affine, eps 1e-6), joint attention, FeedForward(gelu-approximate) -
    the same dataflow and parameter shapes as the FLUX double block, so the FLUX stub is reused under the FLUX
    parameter names (norm1 = img_mod, norm1_context = txt_mod, ff = img_mlp, ff_context = txt_mlp)."""

    def forward(self, hidden_states, encoder_hidden_states, encoder_hidden_states_mask=None, temb=None,
                image_rotary_emb=None, joint_attention_kwargs=None):
        kw = joint_attention_kwargs or {}
        norm_h, gate_msa, shift_mlp, scale_mlp, gate_mlp = self.norm1(hidden_states, emb=temb)
        norm_c, c_gate_msa, c_shift_mlp, c_scale_mlp, c_gate_mlp = self.norm1_context(encoder_hidden_states, emb=temb)
        attn_output, context_attn_output = self.attn(
            hidden_states=norm_h, encoder_hidden_states=norm_c, encoder_hidden_states_mask=encoder_hidden_states_mask,
            image_rotary_emb=image_rotary_emb, **kw)
        hidden_states = hidden_states + gate_msa.unsqueeze(1) * attn_output
        encoder_hidden_states = encoder_hidden_states + c_gate_msa.unsqueeze(1) * context_attn_output
        norm_h = self.norm2(hidden_states) * (1 + scale_mlp[:, None]) + shift_mlp[:, None]
        hidden_states = hidden_states + gate_mlp.unsqueeze(1) * self.ff(norm_h)
        norm_c = self.norm2_context(encoder_hidden_states) * (1 + c_scale_mlp[:, None]) + c_shift_mlp[:, None]
        encoder_hidden_states = encoder_hidden_states + c_gate_mlp.unsqueeze(1) * self.ff_context(norm_c)
        return encoder_hidden_states, hidden_states


class QwenTimestepProjEmbeddings(nn.Module):
    """[EXT] Timesteps(256, flip_sin_to_cos=True, downscale_freq_shift=0, scale=1000) + TimestepEmbedding."""

    def __init__(self, d):
        super().__init__()
        self.timestep_embedder = _MLPEmbed(256, d)

    def forward(self, timestep, hidden_states):
        proj = get_timestep_embedding(timestep, 256, scale=1000.0)       # Timesteps(..., scale=1000) on timestep / 1000
        return self.timestep_embedder(proj.to(hidden_states.dtype))


class QwenEmbedRope(nn.Module):
    """[EXT] QwenEmbedRope(theta, axes_dim, scale_rope=True): complex tables (vid_freqs [sum f*h*w, 64], txt_freqs [T, 64])."""

    def __init__(self, theta=10000, axes_dim=(16, 56, 56)):
        super().__init__()
        self.theta, self.axes_dim = theta, axes_dim
        pos_index, neg_index = torch.arange(4096), torch.arange(4096).flip(0) * -1 - 1
        self.pos_freqs = torch.cat([self._params(pos_index, d) for d in axes_dim], dim=1)
        self.neg_freqs = torch.cat([self._params(neg_index, d) for d in axes_dim], dim=1)

    def _params(self, index, dim):
        freqs = torch.outer(index.float(), 1.0 / torch.pow(self.theta, torch.arange(0, dim, 2).to(torch.float32).div(dim)))
        return torch.polar(torch.ones_like(freqs), freqs)

    def forward(self, video_fhw, txt_seq_lens, device=None):
        if isinstance(video_fhw, list) and isinstance(video_fhw[0], (list, tuple)) and isinstance(video_fhw[0][0], (list, tuple)):
            video_fhw = video_fhw[0]
        vid, max_vid = [], 0
        half = [x // 2 for x in self.axes_dim]
        for idx, (frame, height, width) in enumerate(video_fhw):
            fp = self.pos_freqs.split(half, dim=1)
            fn = self.neg_freqs.split(half, dim=1)
            f = fp[0][idx: idx + frame].view(frame, 1, 1, -1).expand(frame, height, width, -1)
            hh = torch.cat([fn[1][-(height - height // 2):], fp[1][: height // 2]], 0).view(1, height, 1, -1).expand(frame, height, width, -1)
            ww = torch.cat([fn[2][-(width - width // 2):], fp[2][: width // 2]], 0).view(1, 1, width, -1).expand(frame, height, width, -1)
            vid.append(torch.cat([f, hh, ww], dim=-1).reshape(frame * height * width, -1))
            max_vid = max(max_vid, height // 2, width // 2)
        max_len = int(max(txt_seq_lens))
        return torch.cat(vid, 0), self.pos_freqs[max_vid: max_vid + max_len]


class QwenImageTransformer2DModel(nn.Module):
    """[EXT] module tree of diffusers QwenImageTransformer2DModel; forward is replaced by the reference."""

    def __init__(self, in_channels=64, n_double=3, heads=2, head_dim=128, joint_dim=256, axes_dim=(16, 56, 56)):
        super().__init__()
        d = heads * head_dim
        self.config = _Cfg(in_channels=in_channels, guidance_embeds=False)
        self.gradient_checkpointing = False
        self.pos_embed = QwenEmbedRope(theta=10000, axes_dim=axes_dim)
        self.time_text_embed = QwenTimestepProjEmbeddings(d)
        self.txt_norm = RMSNorm(joint_dim, eps=1e-6)
        self.img_in = nn.Linear(in_channels, d)
        self.txt_in = nn.Linear(joint_dim, d)
        self.transformer_blocks = nn.ModuleList([QwenImageTransformerBlock(d, heads, head_dim) for _ in range(n_double)])
        self.norm_out = AdaLayerNormContinuous(d, d)
        self.proj_out = nn.Linear(d, in_channels, bias=True)

    def cache_context(self, name):
        from contextlib import nullcontext
        return nullcontext()

    def forward(self, hidden_states, encoder_hidden_states=None, encoder_hidden_states_mask=None, timestep=None,
                img_shapes=None, txt_seq_lens=None, guidance=None, attention_kwargs=None, return_dict=True):
        """[EXT] vanilla QwenImageTransformer2DModel.forward (call order as the reference's patched copy shows it,
        QwenImageEdit/inplace.py:512-560)."""
        hidden_states = self.img_in(hidden_states)
        timestep = timestep.to(hidden_states.dtype)
        encoder_hidden_states = self.txt_in(self.txt_norm(encoder_hidden_states))
        temb = self.time_text_embed(timestep, hidden_states)
        image_rotary_emb = self.pos_embed(img_shapes, txt_seq_lens, device=hidden_states.device)
        for block in self.transformer_blocks:
            encoder_hidden_states, hidden_states = block(hidden_states=hidden_states, encoder_hidden_states=encoder_hidden_states,
                                                         encoder_hidden_states_mask=encoder_hidden_states_mask, temb=temb,
                                                         image_rotary_emb=image_rotary_emb, joint_attention_kwargs=attention_kwargs)
        output = self.proj_out(self.norm_out(hidden_states, temb))
        return (output,) if not return_dict else _Cfg(sample=output)


# ---------------------------------------------------------------------------------------------------------------------
# [EXT] vanilla attention processors (diffusers FluxAttnProcessor2_0 / QwenDoubleStreamAttnProcessor2_0 semantics) - what a
# stock pipeline's transformer carries before RegionEHelper.enable() replaces them
# ---------------------------------------------------------------------------------------------------------------------
class FluxAttnProcessor2_0:
    """Joint attention of the FLUX / Step1X-Edit blocks: per-head RMSNorm on q / k of both streams, [text ; image]
    concatenation, interleaved RoPE, SDPA, split, output projections (none for the pre_only single-stream module)."""

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, image_rotary_emb=None, **kw):
        B, H = hidden_states.shape[0], attn.heads
        def heads(x):
            return x.view(B, -1, H, x.shape[-1] // H).transpose(1, 2)
        q, k, v = heads(attn.to_q(hidden_states)), heads(attn.to_k(hidden_states)), heads(attn.to_v(hidden_states))
        q, k = attn.norm_q(q), attn.norm_k(k)
        if encoder_hidden_states is not None:
            eq, ek, ev = (heads(p(encoder_hidden_states)) for p in (attn.add_q_proj, attn.add_k_proj, attn.add_v_proj))
            eq, ek = attn.norm_added_q(eq), attn.norm_added_k(ek)
            q, k, v = torch.cat([eq, q], dim=2), torch.cat([ek, k], dim=2), torch.cat([ev, v], dim=2)
        if image_rotary_emb is not None:
            q, k = apply_rotary_emb(q, image_rotary_emb), apply_rotary_emb(k, image_rotary_emb)
        o = F.scaled_dot_product_attention(q, k, v, dropout_p=0.0, is_causal=False)
        o = o.transpose(1, 2).reshape(B, -1, q.shape[-1] * H).to(q.dtype)
        if encoder_hidden_states is not None:
            T = encoder_hidden_states.shape[1]
            return attn.to_out[1](attn.to_out[0](o[:, T:])), attn.to_add_out(o[:, :T])
        return o


def apply_rotary_emb_qwen(x, freqs_cis):
    """[EXT] transformer_qwenimage.apply_rotary_emb_qwen, use_real=False: x [B, S, H, D] times complex freqs [S, D/2]."""
    xc = torch.view_as_complex(x.float().reshape(*x.shape[:-1], -1, 2))
    return torch.view_as_real(xc * freqs_cis.unsqueeze(1)).flatten(3).type_as(x)


class QwenDoubleStreamAttnProcessor2_0:
    """Joint attention of the Qwen-Image double-stream block: [B, S, H, D] layout, complex rotary tables per stream (image
    rows: the video grid; text rows: positions after the image extent), [text ; image] concatenation."""

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, encoder_hidden_states_mask=None, attention_mask=None,
                 image_rotary_emb=None, **kw):
        H, T = attn.heads, encoder_hidden_states.shape[1]
        def heads(x):
            return x.unflatten(-1, (H, -1))
        iq, ik, iv = heads(attn.to_q(hidden_states)), heads(attn.to_k(hidden_states)), heads(attn.to_v(hidden_states))
        tq, tk, tv = (heads(p(encoder_hidden_states)) for p in (attn.add_q_proj, attn.add_k_proj, attn.add_v_proj))
        iq, ik, tq, tk = attn.norm_q(iq), attn.norm_k(ik), attn.norm_added_q(tq), attn.norm_added_k(tk)
        if image_rotary_emb is not None:
            img_freqs, txt_freqs = image_rotary_emb
            iq, ik = apply_rotary_emb_qwen(iq, img_freqs), apply_rotary_emb_qwen(ik, img_freqs)
            tq, tk = apply_rotary_emb_qwen(tq, txt_freqs), apply_rotary_emb_qwen(tk, txt_freqs)
        q, k, v = torch.cat([tq, iq], dim=1), torch.cat([tk, ik], dim=1), torch.cat([tv, iv], dim=1)
        o = F.scaled_dot_product_attention(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2), dropout_p=0.0, is_causal=False)
        o = o.transpose(1, 2).flatten(2, 3).to(q.dtype)
        return attn.to_out[1](attn.to_out[0](o[:, T:])), attn.to_add_out(o[:, :T])


# ---------------------------------------------------------------------------------------------------------------------
# [EXT] Qwen-Image trunk under its REAL parameter names (img_mod.1 / txt_mod.1 / img_mlp / txt_mlp / img_in / txt_in): the
# tree above reuses the FLUX block names (the reference's forward never looks inside a block); the adapter's key map is only
# exercised by a trunk that carries the host's own names.  Used by the host tests only (not by the fixtures).
# ---------------------------------------------------------------------------------------------------------------------
class QwenImageHostBlock(nn.Module):
    def __init__(self, dim, heads, head_dim):
        super().__init__()
        self.img_mod = nn.Sequential(nn.SiLU(), nn.Linear(dim, 6 * dim, bias=True))
        self.img_norm1 = nn.LayerNorm(dim, elementwise_affine=False, eps=1e-6)
        self.attn = Attention(dim, heads, head_dim, added_kv=True)
        self.img_norm2 = nn.LayerNorm(dim, elementwise_affine=False, eps=1e-6)
        self.img_mlp = FeedForward(dim)
        self.txt_mod = nn.Sequential(nn.SiLU(), nn.Linear(dim, 6 * dim, bias=True))
        self.txt_norm1 = nn.LayerNorm(dim, elementwise_affine=False, eps=1e-6)
        self.txt_norm2 = nn.LayerNorm(dim, elementwise_affine=False, eps=1e-6)
        self.txt_mlp = FeedForward(dim)

    @staticmethod
    def _modulate(x, mod_params):
        shift, scale, gate = mod_params.chunk(3, dim=-1)
        return x * (1 + scale.unsqueeze(1)) + shift.unsqueeze(1), gate.unsqueeze(1)

    def forward(self, hidden_states, encoder_hidden_states, encoder_hidden_states_mask=None, temb=None, image_rotary_emb=None,
                joint_attention_kwargs=None):
        img_mod1, img_mod2 = self.img_mod(temb).chunk(2, dim=-1)
        txt_mod1, txt_mod2 = self.txt_mod(temb).chunk(2, dim=-1)
        img_modulated, img_gate1 = self._modulate(self.img_norm1(hidden_states), img_mod1)
        txt_modulated, txt_gate1 = self._modulate(self.txt_norm1(encoder_hidden_states), txt_mod1)
        img_attn, txt_attn = self.attn(hidden_states=img_modulated, encoder_hidden_states=txt_modulated,
                                       encoder_hidden_states_mask=encoder_hidden_states_mask, image_rotary_emb=image_rotary_emb,
                                       **(joint_attention_kwargs or {}))
        hidden_states = hidden_states + img_gate1 * img_attn
        encoder_hidden_states = encoder_hidden_states + txt_gate1 * txt_attn
        img_modulated2, img_gate2 = self._modulate(self.img_norm2(hidden_states), img_mod2)
        hidden_states = hidden_states + img_gate2 * self.img_mlp(img_modulated2)
        txt_modulated2, txt_gate2 = self._modulate(self.txt_norm2(encoder_hidden_states), txt_mod2)
        encoder_hidden_states = encoder_hidden_states + txt_gate2 * self.txt_mlp(txt_modulated2)
        return encoder_hidden_states, hidden_states


class QwenImageHostTransformer2DModel(QwenImageTransformer2DModel):
    def __init__(self, in_channels=64, n_double=3, heads=2, head_dim=128, joint_dim=256, axes_dim=(16, 56, 56)):
        super().__init__(in_channels, 0, heads, head_dim, joint_dim, axes_dim)
        d = heads * head_dim
        self.transformer_blocks = nn.ModuleList([QwenImageHostBlock(d, heads, head_dim) for _ in range(n_double)])


def install_vanilla_processors(trunk):
    """What `from_pretrained` leaves on every block: the family's stock processor."""
    qwen = isinstance(trunk, QwenImageTransformer2DModel)
    for block in list(trunk.transformer_blocks) + list(getattr(trunk, "single_transformer_blocks", [])):
        block.attn.set_processor(QwenDoubleStreamAttnProcessor2_0() if qwen else FluxAttnProcessor2_0())
    return trunk
