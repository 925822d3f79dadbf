"""Step1X-Edit-shaped harness: the FLUX trunk [EXT: same double/single MMDiT blocks] with
temb = time_embed(t) + vec_embed(y) (no guidance embedder) and a batched-CFG pipeline
(/root/reference/RegionE/Step1XEdit/inplace.py:381-410).

The Qwen2.5-VL `connector` that turns the VLM hidden states into (encoder_hidden_states, y)
(Step1XEdit/inplace.py:514-516) is [EXT] and outside the hot path: the harness takes its OUTPUTS
(`prompt_embeds` [1,T,joint] and the pooled vector `y` [1,pooled]) as pipeline inputs.

Batch = 2 of the reference's CFG forward is executed as two passes over the single-image engine, the
branch index travelling as the K/V-cache tag ('cond' / 'uncond'); rows of a batch never interact in
the model, so this is numerically the batched forward.
"""
from __future__ import annotations

import numpy as np
import torch

from .. import dist as D
from .. import ops
from .. import torch_ops as TO          # TO.R = torch.ops.regione_mi: the dispatcher-visible op surface
from ..synth import FluxConfig
from . import flux as H


class Step1XEditAttnProcessor(H.FluxAttnProcessor):
    pass


class Step1XEditTransformer2DModel(H.FluxTransformer2DModel):
    def __init__(self, cfg: FluxConfig, device="cuda"):
        assert not cfg.guidance_embeds, "Step1X-Edit has no guidance embedder"
        super().__init__(cfg, device)
        self._vec = None

    def set_vec(self, y_rows):
        """Connector output y per batch row (cond, uncond)."""
        self._vec = dict(y_rows) if isinstance(y_rows, dict) else list(y_rows)

    def connector(self, encoder_hidden_states, timestep, prompt_embeds_mask):
        """[EXT] hook of the Qwen2 connector (Step1XEdit/inplace.py:514-516).  Latent-level default: the caller already
        passed the connector's OUTPUTS.  A hosted pipeline (regione_amd.adapters) installs the host transformer's real
        connector here as an instance attribute; it is then called once per computed step, like the reference does."""
        return encoder_hidden_states, self._vec

    def branch_inputs(self, encoder_hidden_states, y, timestep, tag):
        """Sequential-CFG form of the same hook (v1p2: one forward per branch `tag`)."""
        c = self.__dict__.get("connector")
        if c is None:
            return encoder_hidden_states, y
        return c(encoder_hidden_states, timestep, None, tag=tag)

    def forward(self, hidden_states, encoder_hidden_states=None, prompt_embeds_mask=None, timestep=None, img_ids=None,
                txt_ids=None, guidance=None, joint_attention_kwargs=None, return_dict=True):
        image_rotary_emb = self.pos_embed(torch.cat((txt_ids.cpu(), img_ids.cpu()), dim=0), self.device)
        return self._run_batched(hidden_states, encoder_hidden_states, prompt_embeds_mask, timestep, image_rotary_emb,
                                 return_dict)

    def _run_batched(self, hidden_states, encoder_hidden_states, prompt_embeds_mask, timestep, image_rotary_emb,
                     return_dict):
        enc, y = self.connector(encoder_hidden_states, timestep, prompt_embeds_mask)
        out_rows = self.__dict__.pop("out_rows_hint", None)
        B = hidden_states.shape[0]
        # the reference's batch of two (Step1XEdit/inplace.py:381-399) = two recorded branches executed as ONE batched pass
        # (RGN_BATCH_BRANCHES=0: two passes over the single-image engine, the round-1/2 behaviour)
        batched = B == 2 and D.branch_batching() and getattr(self, "_batch", None) is None
        if batched:
            self.begin_batch()
        outs = []
        try:
            for b in range(B):
                tag = "cond" if b == 0 else "uncond"
                outs.append(self._run(hidden_states[b:b + 1], enc[b:b + 1], y[b], timestep[b:b + 1], None, image_rotary_emb,
                                      False, {"tag": tag}, out_rows=out_rows)[0])
        except BaseException:
            if batched:
                self.abort_batch()
            raise
        if batched:
            res = self.end_batch()
            outs = [h.resolve(res) for h in outs]
        out = outs[0] if len(outs) == 1 else ops.cat_rows(outs, dim=0)
        return (out,) if not return_dict else H._Cfg(sample=out)


class Step1XEditPipelineOutput(H._Cfg):
    pass


class Step1XEditPipeline(H.FluxKontextPipeline):
    """Vanilla full-token loop with batched true-CFG (reference defaults true_cfg_scale = 6.0)."""

    def process_diff_norm(self, diff_norm, k):   # kept for API parity; the arithmetic runs in rgn_cfg_combine
        raise NotImplementedError("fused into ops.cfg_combine(mode=CFG_STEP1X_RESCALE)")

    def _cfg(self, noise_pred, t, true_cfg_scale, timesteps_truncate, process_norm_power):
        pos, neg = noise_pred[0:1], noise_pred[1:2]
        mode = ops.CFG_STEP1X_RESCALE if float(t) > timesteps_truncate else ops.CFG_PLAIN      # inplace.py:401
        return TO.R.cfg_combine(pos.contiguous(), neg.contiguous(), true_cfg_scale, mode, process_norm_power)

    def _batched_inputs(self, x, prompt_embeds, negative_prompt_embeds):
        """The B = 2 inputs of the reference's batched CFG (Step1XEdit/inplace.py:381-385): `cat((x, x))` as a copy-free repeat, the
        two prompt embeddings stacked once per (cond, uncond) pair of tensors (device-to-device copies, not once per step)."""
        # keyed on the tensors THEMSELVES (held, compared by identity) + their versions and shapes: an address key would be recycled by the
        # caching allocator for the next call's same-shaped embeddings and serve the previous prompt's conditioning (advisor, round 5)
        def ver(t):
            return None if t.is_inference() else t._version
        src = getattr(self, "_pe_src", None)
        key = (ver(prompt_embeds), ver(negative_prompt_embeds), tuple(prompt_embeds.shape), tuple(negative_prompt_embeds.shape))
        if src is None or src[0] is not prompt_embeds or src[1] is not negative_prompt_embeds or self._pe_key != key:
            self._pe_src, self._pe_key = (prompt_embeds, negative_prompt_embeds), key
            self._pe = ops.cat_rows((prompt_embeds, negative_prompt_embeds), dim=0)
        return H.repeat_batch(x, 2), self._pe

    @torch.no_grad()
    def __call__(self, image=None, prompt_embeds=None, pooled_prompt_embeds=None, negative_prompt_embeds=None,
                 negative_pooled_prompt_embeds=None, height=1024, width=1024, num_inference_steps=28,
                 true_cfg_scale=6.0, guidance_scale=6.0, latents=None, generator=None, output_type="latent",
                 return_dict=True, timesteps_truncate=0.93, process_norm_power=0.4, sigmas=None,
                 callback_on_step_end=None, callback_on_step_end_tensor_inputs=("latents",)):
        latents, image_latents, latent_ids, text_ids, _, _ = self.prepare(
            image, prompt_embeds, pooled_prompt_embeds, height, width, latents, generator, num_inference_steps, sigmas)
        timesteps = self.scheduler.timesteps
        self.scheduler.set_begin_index(0)
        tr = self.transformer
        if hasattr(tr, "set_vec"):
            tr.set_vec((pooled_prompt_embeds, negative_pooled_prompt_embeds))
        self._precompute(timesteps, None, latents.dtype, pooled_prompt_embeds, negative_pooled_prompt_embeds)
        for i, t in enumerate(timesteps):
            x, pe = self._batched_inputs(H.cat_tokens(self.transformer, latents, image_latents), prompt_embeds, negative_prompt_embeds)
            timestep = t.expand(latents.shape[0]).to(latents.dtype)
            timestep = torch.cat((timestep, timestep), dim=0)
            tr.out_rows_hint = latents.size(1)
            noise_pred = tr(hidden_states=x, timestep=timestep / 1000, guidance=None, encoder_hidden_states=pe,
                            prompt_embeds_mask=None, txt_ids=text_ids, img_ids=latent_ids, return_dict=False)[0]
            noise_pred = noise_pred[:, : latents.size(1)]
            noise_pred = self._cfg(noise_pred, t, true_cfg_scale, timesteps_truncate, process_norm_power)
            latents = self.scheduler.step(noise_pred, t, latents, return_dict=False)[0]
            latents, prompt_embeds = self._callback(callback_on_step_end, callback_on_step_end_tensor_inputs, i, t, latents,
                                                    prompt_embeds, noise_pred=noise_pred, image_latents=image_latents,
                                                    negative_prompt_embeds=negative_prompt_embeds)
        if not return_dict:
            return (latents,)
        return Step1XEditPipelineOutput(images=latents)


class Step1XEditPipelineV1P2(Step1XEditPipeline):
    """Vanilla v1p2 loop: SEQUENTIAL cond / uncond forwards (Step1XEditV1P2/inplace.py:388-430)."""

    @torch.no_grad()
    def __call__(self, image=None, prompt_embeds=None, pooled_prompt_embeds=None, negative_prompt_embeds=None,
                 negative_pooled_prompt_embeds=None, height=1024, width=1024, num_inference_steps=28,
                 true_cfg_scale=6.0, guidance_scale=6.0, latents=None, generator=None, output_type="latent",
                 return_dict=True, timesteps_truncate=0.93, process_norm_power=0.4, sigmas=None,
                 callback_on_step_end=None, callback_on_step_end_tensor_inputs=("latents",)):
        latents, image_latents, latent_ids, text_ids, _, _ = self.prepare(
            image, prompt_embeds, pooled_prompt_embeds, height, width, latents, generator, num_inference_steps, sigmas)
        neg_text_ids = torch.zeros(negative_prompt_embeds.shape[1], 3)
        timesteps = self.scheduler.timesteps
        self.scheduler.set_begin_index(0)
        tr = self.transformer
        self._precompute(timesteps, None, latents.dtype, pooled_prompt_embeds, negative_pooled_prompt_embeds)
        for i, t in enumerate(timesteps):
            x = H.cat_tokens(self.transformer, latents, image_latents)
            timestep = t.expand(latents.shape[0]).to(latents.dtype)
            def branch(pe, y, ids_t, tag):
                rope = tr.pos_embed(torch.cat((ids_t, latent_ids), dim=0), tr.device)
                pe, y = tr.branch_inputs(pe, y, timestep / 1000, tag)
                return tr._run(x, pe, y, timestep / 1000, None, rope, False, {"tag": tag},
                               out_rows=latents.size(1))[0][:, : latents.size(1)]
            outs = D.run_cfg_branches(getattr(self, "_cfg_pair", None),
                                      lambda: branch(prompt_embeds, pooled_prompt_embeds, text_ids, "cond"),
                                      lambda: branch(negative_prompt_embeds, negative_pooled_prompt_embeds, neg_text_ids, "uncond"),
                                      batch_on=tr)
            mode = ops.CFG_STEP1X_RESCALE if float(t) > timesteps_truncate else ops.CFG_PLAIN
            noise_pred = TO.R.cfg_combine(outs[0], outs[1], true_cfg_scale, mode, process_norm_power)
            latents = self.scheduler.step(noise_pred, t, latents, return_dict=False)[0]
            latents, prompt_embeds = self._callback(callback_on_step_end, callback_on_step_end_tensor_inputs, i, t, latents,
                                                    prompt_embeds, noise_pred=noise_pred, image_latents=image_latents,
                                                    negative_prompt_embeds=negative_prompt_embeds)
        if not return_dict:
            return (latents,)
        return Step1XEditPipelineOutput(images=latents)
