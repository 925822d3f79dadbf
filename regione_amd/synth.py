"""Seeded synthetic weights and inputs for the MMDiT harness (no checkpoints / no network).

SURVEY.md section 8d: weights N(0, 0.02^2), RMSNorm weights 1 (perturbed slightly so the test can
see them), biases 0.01*N(0,1); latents / condition latents N(0,1); prompt embeds N(0,1).
Names follow the diffusers FLUX state-dict layout so a real checkpoint maps 1:1.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Tuple

import torch


@dataclass
class FluxConfig:
    """Public FLUX.1 / Step1X-Edit trunk dimensions [EXT]; override for toy-size parity tests."""
    in_channels: int = 64
    n_double: int = 19
    n_single: int = 38
    heads: int = 24
    head_dim: int = 128
    joint_dim: int = 4096
    pooled_dim: int = 768
    axes_dim: Tuple[int, ...] = (16, 56, 56)
    mlp_ratio: int = 4
    guidance_embeds: bool = True          # False: Step1X-Edit style temb = time_embed + vec_embed(y)
    pooled_embeds: bool = True            # False: Qwen-Image style temb = timestep embedding only
    txt_norm: bool = False                # True: RMSNorm on the prompt embeddings before the text projection (Qwen)

    @property
    def d(self) -> int:
        return self.heads * self.head_dim

    @property
    def n_layers(self) -> int:
        return self.n_double + self.n_single


TOY = dict(n_double=2, n_single=2, heads=2, head_dim=128, joint_dim=256, pooled_dim=64)
# Qwen-Image-Edit [EXT public config]: 60 double-stream blocks only, text width 3584, no pooled / guidance embedders
QWEN = dict(n_double=60, n_single=0, heads=24, head_dim=128, joint_dim=3584, guidance_embeds=False, pooled_embeds=False,
            txt_norm=True)
QWEN_TOY = dict(n_double=3, n_single=0, heads=2, head_dim=128, joint_dim=256, guidance_embeds=False, pooled_embeds=False,
                txt_norm=True)


def flux_param_shapes(cfg: FluxConfig) -> Dict[str, Tuple[int, ...]]:
    d, ff = cfg.d, cfg.d * cfg.mlp_ratio
    s: Dict[str, Tuple[int, ...]] = {}

    def lin(name, n_out, n_in):
        s[name + ".weight"] = (n_out, n_in)
        s[name + ".bias"] = (n_out,)

    lin("x_embedder", d, cfg.in_channels)
    lin("context_embedder", d, cfg.joint_dim)
    for e, din in (("timestep_embedder", 256), ("guidance_embedder", 256), ("text_embedder", cfg.pooled_dim)):
        if e == "guidance_embedder" and not cfg.guidance_embeds:
            continue
        if e == "text_embedder" and not cfg.pooled_embeds:
            continue
        lin(f"time_text_embed.{e}.linear_1", d, din)
        lin(f"time_text_embed.{e}.linear_2", d, d)
    for i in range(cfg.n_double):
        p = f"transformer_blocks.{i}"
        lin(p + ".norm1.linear", 6 * d, d)
        lin(p + ".norm1_context.linear", 6 * d, d)
        for n in ("to_q", "to_k", "to_v", "add_q_proj", "add_k_proj", "add_v_proj", "to_out.0", "to_add_out"):
            lin(p + ".attn." + n, d, d)
        for n in ("norm_q", "norm_k", "norm_added_q", "norm_added_k"):
            s[p + f".attn.{n}.weight"] = (cfg.head_dim,)
        for f in ("ff", "ff_context"):
            lin(p + f".{f}.net.0.proj", ff, d)
            lin(p + f".{f}.net.2", d, ff)
    for i in range(cfg.n_single):
        p = f"single_transformer_blocks.{i}"
        lin(p + ".norm.linear", 3 * d, d)
        lin(p + ".proj_mlp", ff, d)
        lin(p + ".proj_out", d, d + ff)
        for n in ("to_q", "to_k", "to_v"):
            lin(p + ".attn." + n, d, d)
        for n in ("norm_q", "norm_k"):
            s[p + f".attn.{n}.weight"] = (cfg.head_dim,)
    if cfg.txt_norm:
        s["txt_norm.weight"] = (cfg.joint_dim,)
    lin("norm_out.linear", 2 * d, d)
    lin("proj_out", cfg.in_channels, d)
    return s


def is_modulation(name: str) -> bool:
    """AdaLN modulation linears (`norm1.linear`, `norm1_context.linear`, `norm.linear`, `norm_out.linear`): their outputs are the
    shift / scale / gate vectors of every block."""
    return "norm" in name and (name.endswith(".linear.weight") or name.endswith(".linear.bias"))


def make_flux_weights(cfg: FluxConfig, seed: int = 42, dtype=torch.bfloat16, device="cpu",
                      w_std: float = 0.02, mod_scale: float = 1.0, norm_mean: float = 1.0, norm_std: float = 0.1) -> Dict[str, torch.Tensor]:
    """Deterministic per (cfg, seed, device-type): one generator, tensors drawn in dict order.

    `mod_scale`, `norm_mean / norm_std` calibrate the statistics towards a trained checkpoint's: with
    variance-preserving N(0, 1/d) Linears the AdaLN modulation vectors come out O(1) - every block then adds an O(1)-relative, gate-1
    update to the residual stream and a 57 / 60-block trunk amplifies any rounding difference; trained MMDiTs have gates / scales of a few
    tenths and RMSNorm weights above 1.  `mod_scale = 0.2` scales the modulation linears (weights and biases) AFTER they are drawn, so the
    random stream - and every other tensor - is unchanged."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    out = {}
    for name, shape in flux_param_shapes(cfg).items():
        if name.endswith(".bias"):
            t = torch.randn(shape, generator=g, device=device, dtype=torch.float32) * 0.01
        elif len(shape) == 1:                       # RMSNorm weight
            t = norm_mean + norm_std * torch.randn(shape, generator=g, device=device, dtype=torch.float32)
        else:
            t = torch.randn(shape, generator=g, device=device, dtype=torch.float32) * w_std
        if mod_scale != 1.0 and is_modulation(name):
            t = t * mod_scale
        out[name] = t.to(dtype)
    return out


def flux_latent_ids(h_tok: int, w_tok: int, device="cpu") -> torch.Tensor:
    """[EXT] FluxKontextPipeline._prepare_latent_image_ids for noise + condition image:
    rows (0,row,col) for the L noise tokens then (1,row,col) for the L_c condition tokens."""
    ids = torch.zeros(h_tok, w_tok, 3)
    ids[..., 1] = torch.arange(h_tok)[:, None]
    ids[..., 2] = torch.arange(w_tok)[None, :]
    ids = ids.reshape(h_tok * w_tok, 3)
    img = ids.clone()
    img[:, 0] = 1
    return torch.cat([ids, img], 0).to(device)


def make_edit_inputs(h_tok: int, w_tok: int, txt_len: int, cfg: FluxConfig, seed: int = 42,
                     dtype=torch.bfloat16, device="cpu"):
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    L = h_tok * w_tok
    latents = torch.randn(1, L, cfg.in_channels, generator=g).to(dtype).to(device)
    image_latents = torch.randn(1, L, cfg.in_channels, generator=g).to(dtype).to(device)
    prompt = torch.randn(1, txt_len, cfg.joint_dim, generator=g).to(dtype).to(device)
    pooled = torch.randn(1, cfg.pooled_dim, generator=g).to(dtype).to(device)
    return latents, image_latents, prompt, pooled


def region_target(h_tok: int, w_tok: int, box, image_latents: torch.Tensor, seed: int = 7,
                  ramp: float = 0.0) -> torch.Tensor:
    """Per-token target x0 that forces the edited region by construction:
    outside `box=(r0,r1,c0,c1)` target = condition latent (+ a smooth noise ramp so cosine values
    straddle the threshold when ramp > 0); inside it is independent N(0,1)."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    L, D = h_tok * w_tok, image_latents.shape[-1]
    cond = image_latents[0].float().cpu()
    noise = torch.randn(L, D, generator=g)
    tgt = cond.clone()
    if ramp > 0:
        col = (torch.arange(L) % w_tok).float() / max(w_tok - 1, 1)
        tgt = cond + noise * (ramp * col)[:, None]
    r0, r1, c0, c1 = box
    inside = torch.zeros(h_tok, w_tok, dtype=torch.bool)
    inside[r0:r1, c0:c1] = True
    inside = inside.reshape(-1)
    tgt[inside] = torch.randn(int(inside.sum()), D, generator=g)
    return tgt


def arp_case(seed: int, h: int, w: int, cond_dtype=torch.float32):
    """Seeded (estimate fp32 [1,L,64], condition [1,L,64]) pair for partition tests: the estimate is
    the condition plus column-ramped noise (cosine values straddle every threshold), an edited box
    of independent noise, and 3 % speckles that the erosion must remove."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    L = h * w
    cond = torch.randn(1, L, 64, generator=g)
    col = (torch.arange(L) % w).float() / max(w - 1, 1)
    est = cond + torch.randn(1, L, 64, generator=g) * (1.2 * col)[None, :, None]
    box = torch.zeros(h, w, dtype=torch.bool)
    box[h // 4: h // 4 + h // 3, w // 5: w // 5 + w // 3] = True
    sp = torch.rand(h, w, generator=g) < 0.03
    m = (box | sp).reshape(-1)
    est[0, m] = torch.randn(int(m.sum()), 64, generator=g)
    return est, cond.to(cond_dtype)
