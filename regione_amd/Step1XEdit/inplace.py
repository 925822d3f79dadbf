"""RegionE patch set for Step1X-Edit (v1p1) on the HIP kernels.

Mirrors /root/reference/RegionE/Step1XEdit/inplace.py: `warp_modules` / `unwarp_modules` (:52-71),
`RegionEStep1XEditPipeline.__call__` (:73-457), `RegionEStep1XEditTransformer2DModelforward` (:460-578),
the scheduler (:581-695, identical to FLUX's) and `RegionEStep1XEditAttnProcessor` (:698-811, identical
K/V-cache protocol).  Family deltas vs FLUX:
  * CFG is BATCHED: one forward on [cond ; uncond] (:381-399); the partition sees the already
    combined B = 1 prediction and the edited ids are shared by the pair (quirk A-5);
  * norm-rescaled CFG for t > timesteps_truncate (:401-410) -> rgn_cfg_combine(mode 1);
  * its own fitted gamma table (:47-49), default thresholds 0.88 / 0.02 (tool/RegionE.py:3).
"""
from __future__ import annotations

from typing import Optional

import torch

from .. import ops
from .. import torch_ops as TO          # TO.R = torch.ops.regione_mi: the dispatcher-visible op surface
from ..FluxKontext import inplace as fk
from ..harness import flux as H
from ..harness import step1x as HS
from .utils import Step1XEditManager, ids_gather

gamma = torch.tensor([0.9746, 0.9593, 1.0036, 1.0084, 1.0106, 1.0114, 1.0138, 1.0163, 1.0152,
                      1.0163, 1.0197, 1.0186, 1.0219, 1.0218, 1.0223, 1.0266, 1.0272, 1.0305,
                      1.0311, 1.0362, 1.0385, 1.0423, 1.0500, 1.0536, 1.0671, 1.0866, 1.1015], dtype=torch.float16)

RegionEFlowMatchEulerDiscreteScheduler = fk.RegionEFlowMatchEulerDiscreteScheduler
RegionEStep1XEditAttnProcessor = fk.RegionEFluxAttnProcessor


def warp_modules(pipeline, **args):
    manager = Step1XEditManager()
    manager.set_parameters(dict(args))
    pipeline._regione_manager = manager
    pipeline._regione_vanilla_class = pipeline.__class__
    pipeline.__class__ = RegionEStep1XEditPipeline
    sch = RegionEFlowMatchEulerDiscreteScheduler.from_config(pipeline.scheduler.config)
    sch.manager = manager
    pipeline.scheduler = sch
    tr = pipeline.transformer
    tr._regione_manager = manager
    tr.forward = RegionEStep1XEditTransformer2DModelforward.__get__(tr, tr.__class__)
    for block in tr.transformer_blocks:
        block.attn.set_processor(RegionEStep1XEditAttnProcessor(False, manager))
    for block in tr.single_transformer_blocks:
        block.attn.set_processor(RegionEStep1XEditAttnProcessor(True, manager))
    return pipeline


def unwarp_modules(pipeline):
    pipeline.__class__ = getattr(pipeline, "_regione_vanilla_class", HS.Step1XEditPipeline)
    pipeline.scheduler = H.FlowMatchEulerDiscreteScheduler.from_config(pipeline.scheduler.config)
    tr = pipeline.transformer
    if "forward" in tr.__dict__:
        del tr.__dict__["forward"]
    for block in tr.transformer_blocks:
        block.attn.set_processor(HS.Step1XEditAttnProcessor(False))
    for block in tr.single_transformer_blocks:
        block.attn.set_processor(HS.Step1XEditAttnProcessor(True))
    pipeline._regione_manager = None
    return pipeline


class RegionEStep1XEditPipeline(HS.Step1XEditPipeline):

    @torch.no_grad()
    def __call__(self, image=None, prompt_embeds=None, pooled_prompt_embeds=None, negative_prompt_embeds=None,
                 negative_pooled_prompt_embeds=None, height=1024, width=1024, num_inference_steps=28,
                 true_cfg_scale=6.0, guidance_scale=6.0, latents=None, generator=None, output_type="latent",
                 return_dict=True, timesteps_truncate=0.93, process_norm_power=0.4, trace: Optional[dict] = None, sigmas=None,
                 callback_on_step_end=None, callback_on_step_end_tensor_inputs=("latents",)):
        MANAGER = self._regione_manager
        assert num_inference_steps == MANAGER.inference_step, "inference step mismatch"
        do_true_cfg = true_cfg_scale > 1                                     # :230
        latents, image_latents, latent_ids, text_ids, _, _ = self.prepare(
            image, prompt_embeds, pooled_prompt_embeds, height, width, latents, generator, num_inference_steps, sigmas)
        timesteps = self.scheduler.timesteps
        MANAGER.refresh(latents, image_latents, latent_ids, text_ids, 2, self.vae_scale_factor, height, width)
        avd, cache = fk.AvdState(), None
        self.scheduler.set_begin_index(0)
        tr = self.transformer
        if hasattr(tr, "set_vec"):
            tr.set_vec((pooled_prompt_embeds, negative_pooled_prompt_embeds))
        self._precompute(timesteps, None, latents.dtype, pooled_prompt_embeds,
                         negative_pooled_prompt_embeds if do_true_cfg else None)
        for i, t in enumerate(timesteps):
            assert i == MANAGER.current_step
            should_cache, ratio = fk.avd_decide(MANAGER, avd, i, timesteps, gamma)      # :345-363
            if should_cache:                                                        # :365-369
                first_hit = cache.shape[1] != latents.shape[1]
                noise_pred = TO.R.avd_apply(cache, float(ratio), MANAGER.edited_ids if first_hit else None, MANAGER.avd_round_ratio)
                if first_hit:
                    cache = ids_gather(cache, MANAGER.edited_ids)
            else:
                x = latents
                if MANAGER.is_full_input_step():                                    # :378-379
                    x = H.cat_tokens(self.transformer, latents, image_latents)
                timestep = t.expand(latents.shape[0]).to(latents.dtype)
                assert do_true_cfg, "the reference leaves noise_pred undefined without true CFG (:381-399)"
                xb, pe = self._batched_inputs(x, prompt_embeds, negative_prompt_embeds)
                timestep = torch.cat((timestep, timestep), dim=0)
                tr.out_rows_hint = latents.size(1)
                noise_pred = tr(hidden_states=xb, timestep=timestep / 1000, guidance=None, encoder_hidden_states=pe,
                                prompt_embeds_mask=None, txt_ids=text_ids, img_ids=latent_ids, return_dict=False)[0]
                noise_pred = noise_pred[:, : latents.size(1)]
                noise_pred = self._cfg(noise_pred, t, true_cfg_scale, timesteps_truncate, process_norm_power)
                cache = noise_pred                                                  # :411
            if trace is not None:
                trace.setdefault("kind", []).append("C" if should_cache else ("F" if MANAGER.is_full_input_step() else "R"))
                trace.setdefault("noise_pred", []).append(noise_pred.clone())
            latents = self.scheduler.step(noise_pred, t, latents, return_dict=False)[0]
            latents, prompt_embeds = self._callback(callback_on_step_end, callback_on_step_end_tensor_inputs, i, t, latents,
                                                    prompt_embeds, noise_pred=noise_pred, image_latents=image_latents,
                                                    negative_prompt_embeds=negative_prompt_embeds)
            latents, latent_ids = MANAGER.step(latents, latent_ids)
            if trace is not None:
                trace.setdefault("latents", []).append(latents.clone())
                trace.setdefault("prev_refresh", []).append(MANAGER.prev_refresh_step)
        if not return_dict:
            return (latents,)
        return HS.Step1XEditPipelineOutput(images=latents)


def RegionEStep1XEditTransformer2DModelforward(self, hidden_states, encoder_hidden_states=None, prompt_embeds_mask=None,
                                               timestep=None, img_ids=None, txt_ids=None, guidance=None,
                                               joint_attention_kwargs=None, return_dict=True):
    """Step1XEdit/inplace.py:460-578: query RoPE table from the current ids, key table from the full ids."""
    MANAGER = self._regione_manager
    image_rotary_emb = fk.dual_rope_tables(self, MANAGER, txt_ids, img_ids)
    return self._run_batched(hidden_states, encoder_hidden_states, prompt_embeds_mask, timestep, image_rotary_emb,
                             return_dict)
