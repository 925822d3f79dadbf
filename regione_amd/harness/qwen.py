"""Qwen-Image-Edit-shaped harness on the HIP engine.

[EXT] QwenImageTransformer2DModel = 60 double-stream MMDiT blocks (img_mod/txt_mod = SiLU + Linear(d, 6d),
LayerNorm without affine, joint attention with RMSNorm on q/k of both streams, GELU-tanh FeedForward) -
chunk order (shift, scale, gate) x 2 is the same as FLUX's AdaLN-Zero, so the FLUX double-block launch
plan is reused with n_single = 0.  Differences handled here:
  * conditioning = timestep embedding only (no pooled / guidance embedder);
  * `txt_norm` (RMSNorm over the 3584-wide prompt embeddings) before the text projection;
  * rotary table: QwenEmbedRope (complex, 3 axes 16/56/56, centred positions, text after the image
    extent) - applying complex freqs to (x[2i], x[2i+1]) pairs is the same rotation as FLUX's real
    interleaved form, so the kernels consume a (cos, sin) table built on the host here;
  * latent ids are 1-D (`arange`, QwenImageEdit/inplace.py:322);
  * CFG is sequential with 'cond' / 'uncond' tags and norm-preserving (QwenImageEdit/inplace.py:371-405).
"""
from __future__ import annotations

import torch

from .. import dist as D
from .. import ops
from .. import torch_ops as TO          # TO.R = torch.ops.regione_mi: the dispatcher-visible op surface
from ..synth import FluxConfig
from . import flux as H


def _rope_params(index: torch.Tensor, dim: int, theta: float = 10000.0) -> torch.Tensor:
    """[EXT] QwenEmbedRope.rope_params -> angles [len(index), dim/2] (fp32 like the module's float32 path)."""
    return torch.outer(index.float(), 1.0 / torch.pow(theta, torch.arange(0, dim, 2).float().div(dim)))


class QwenEmbedRope:
    """[EXT] diffusers QwenEmbedRope(theta=10000, axes_dim=(16,56,56), scale_rope=True), restated on the host."""

    def __init__(self, theta=10000, axes_dim=(16, 56, 56), scale_rope=True):
        self.theta, self.axes_dim, self.scale_rope = theta, tuple(axes_dim), scale_rope
        pos_index = torch.arange(4096)
        neg_index = torch.arange(4096).flip(0) * -1 - 1
        self.pos = [_rope_params(pos_index, d, theta) for d in self.axes_dim]
        self.neg = [_rope_params(neg_index, d, theta) for d in self.axes_dim]

    def video_angles(self, frame, height, width, idx):
        f = self.pos[0][idx: idx + frame].view(frame, 1, 1, -1).expand(frame, height, width, -1)
        if self.scale_rope:
            hh = torch.cat([self.neg[1][-(height - height // 2):], self.pos[1][: height // 2]], dim=0)
            ww = torch.cat([self.neg[2][-(width - width // 2):], self.pos[2][: width // 2]], dim=0)
        else:
            hh, ww = self.pos[1][:height], self.pos[2][:width]
        hh = hh.view(1, height, 1, -1).expand(frame, height, width, -1)
        ww = ww.view(1, 1, width, -1).expand(frame, height, width, -1)
        return torch.cat([f, hh, ww], dim=-1).reshape(frame * height * width, -1)

    def __call__(self, img_shapes, txt_len: int, device="cuda"):
        """img_shapes: [(frame, h, w), ...] -> (cos, sin) fp32 [txt_len + sum(f*h*w), 128], text rows first."""
        ang, max_vid = [], 0
        for idx, (fr, h, w) in enumerate(img_shapes):
            ang.append(self.video_angles(fr, h, w, idx))
            max_vid = max(max_vid, h // 2, w // 2) if self.scale_rope else max(max_vid, h, w)
        txt = torch.cat([p[max_vid: max_vid + txt_len] for p in self.pos], dim=1)
        a = torch.cat([txt] + ang, dim=0)                                  # [T + N, 64] angles
        # diffusers builds the table as torch.polar(1, angle) (complex64); its real / imaginary parts differ from
        # angle.cos() / angle.sin() by an ulp here and there, and parity is against THAT table
        z = torch.polar(torch.ones_like(a), a)
        cos = z.real.repeat_interleave(2, dim=1).contiguous().to(device)
        sin = z.imag.repeat_interleave(2, dim=1).contiguous().to(device)
        return cos, sin


class QwenDoubleStreamAttnProcessor2_0(H.FluxAttnProcessor):
    def __init__(self):
        super().__init__(False)


class QwenImageTransformer2DModel(H.FluxTransformer2DModel):
    def __init__(self, cfg: FluxConfig, device="cuda"):
        assert cfg.n_single == 0 and not cfg.pooled_embeds and not cfg.guidance_embeds
        super().__init__(cfg, device)
        self.pos_embed = QwenEmbedRope(10000, cfg.axes_dim, True)

    def full_rope(self, img_shapes, T):
        key = (tuple(tuple(s) for s in img_shapes), T)
        cache = self.__dict__.setdefault("_rope_cache", {})
        if key not in cache:
            if len(cache) > 8:
                cache.clear()
            cache[key] = self.pos_embed(img_shapes, T, self.device)
        return cache[key]

    def forward(self, hidden_states, encoder_hidden_states=None, encoder_hidden_states_mask=None, timestep=None,
                img_shapes=None, txt_seq_lens=None, guidance=None, attention_kwargs=None, latent_ids=None,
                return_dict=True):
        T = encoder_hidden_states.shape[1]
        rope = self.full_rope(img_shapes[0], T)
        return self._run(hidden_states, encoder_hidden_states, None, timestep, None, rope, return_dict, attention_kwargs)

    def cache_context(self, name):
        from contextlib import nullcontext
        return nullcontext()


class QwenImagePipelineOutput(H._Cfg):
    pass


class QwenImageEditPipeline(H.FluxKontextPipeline):
    """Vanilla loop: sequential cond / uncond forwards + norm-preserving CFG (inplace.py:371-405)."""

    def _shapes(self, height, width, cond_shapes=None):
        """img_shapes of the forward (QwenImageEdit/inplace.py:272-277): the noise grid, then one (1, h_tok, w_tok) per
        condition image - several for Qwen-Image-Edit-2509 (QwenImageEditPlus/inplace.py:293-300); default: one, same size."""
        s = (1, height // self.vae_scale_factor // 2, width // self.vae_scale_factor // 2)
        if cond_shapes is None:
            return [[s, s]]
        return [[s] + [(1, int(h), int(w)) for h, w in cond_shapes]]

    def prepare_qwen(self, image, height, width, latents, generator, num_inference_steps, cond_shapes=None, sigmas=None):
        """`image`: packed condition latents [1, L_c, 64], or a list of them (one per condition image, in order)."""
        if isinstance(image, (list, tuple)):
            if cond_shapes is not None:
                assert [int(h) * int(w) for h, w in cond_shapes] == [x.shape[1] for x in image], "cond_shapes vs latents"
            image = torch.cat(list(image), dim=1)
        if cond_shapes is not None:
            assert sum(int(h) * int(w) for h, w in cond_shapes) == image.shape[1], "cond_shapes do not cover the condition latents"
        dummy = torch.zeros(1, 1, 1)
        latents, image_latents, _, _, h_tok, w_tok = self.prepare(image, dummy, None, height, width, latents, generator,
                                                                  num_inference_steps, sigmas)
        latent_ids = torch.arange(latents.shape[1] + image_latents.shape[1])           # :322
        return latents, image_latents, latent_ids

    @torch.no_grad()
    def __call__(self, image=None, prompt_embeds=None, negative_prompt_embeds=None, height=1024, width=1024,
                 num_inference_steps=28, true_cfg_scale=4.0, latents=None, generator=None, output_type="latent",
                 return_dict=True, cond_shapes=None, sigmas=None,
                 callback_on_step_end=None, callback_on_step_end_tensor_inputs=("latents",)):
        latents, image_latents, latent_ids = self.prepare_qwen(image, height, width, latents, generator, num_inference_steps,
                                                               cond_shapes, sigmas)
        timesteps = self.scheduler.timesteps
        img_shapes = self._shapes(height, width, cond_shapes)
        do_true_cfg = true_cfg_scale > 1 and negative_prompt_embeds is not None
        self.scheduler.set_begin_index(0)
        self._precompute(timesteps, None, latents.dtype)
        tr = self.transformer
        for i, t in enumerate(timesteps):
            x = H.cat_tokens(self.transformer, latents, image_latents)
            timestep = t.expand(latents.shape[0]).to(latents.dtype)
            def branch(embeds, tag):
                tr.out_rows_hint = latents.size(1)
                return tr(hidden_states=x, timestep=timestep / 1000, encoder_hidden_states=embeds, img_shapes=img_shapes,
                          latent_ids=latent_ids, attention_kwargs={"tag": tag}, return_dict=False)[0][:, : latents.size(1)]
            if do_true_cfg:
                noise_pred, neg = D.run_cfg_branches(getattr(self, "_cfg_pair", None),
                                                     lambda: branch(prompt_embeds, "cond"),
                                                     lambda: branch(negative_prompt_embeds, "uncond"), batch_on=tr)
                noise_pred = TO.R.cfg_combine(noise_pred, neg, true_cfg_scale, ops.CFG_QWEN_NORM)
            else:
                noise_pred = branch(prompt_embeds, "cond")
            latents = self.scheduler.step(noise_pred, t, latents, return_dict=False)[0]
            latents, prompt_embeds = self._callback(callback_on_step_end, callback_on_step_end_tensor_inputs, i, t, latents,
                                                    prompt_embeds, noise_pred=noise_pred, image_latents=image_latents,
                                                    negative_prompt_embeds=negative_prompt_embeds)
        if not return_dict:
            return (latents,)
        return QwenImagePipelineOutput(images=latents)


class QwenImageEditPlusPipeline(QwenImageEditPipeline):
    """Qwen-Image-Edit-2509: same kernels; multi-image conditioning (QwenImageEditPlus/inplace.py:229-299) = a LIST of
    packed condition latents + `cond_shapes` [(h_tok, w_tok), ...]: one rotary frame index per image, keys / values over
    T + L + sum(L_c) rows."""
