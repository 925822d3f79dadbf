"""RegionE patch set for Step1X-Edit v1p2 on the HIP kernels.

Mirrors /root/reference/RegionE/Step1XEditV1P2/inplace.py (`warp_modules` :53-62, `__call__` :76-545,
forward :546-690, processor :806-945).  Family deltas:
  * CFG is SEQUENTIAL: a 'cond' and an 'uncond' forward per computed step, the branch travelling as
    `joint_attention_kwargs['tag']` into the processors, which keep one K/V cache per tag
    (`k_cache_even/odd`, :800-888);
  * the two branches have different text lengths (`txt_length` / `neg_txt_length`, selection :833,:868);
  * norm-rescaled CFG (:421-430) -> rgn_cfg_combine(mode 1); v1p2 gamma table (:48-50).
Out of scope (VLM prompting, not the hot path): the thinking / reflection retry loop (:192-212,470-486)
and `text_token_mapping` ([EXT] extra text projection, :606-609) - the harness consumes the connector /
mapping OUTPUTS as prompt embeddings.
"""
from __future__ import annotations

from typing import Optional

import torch

from .. import dist as D
from .. import ops
from .. import torch_ops as TO          # TO.R = torch.ops.regione_mi: the dispatcher-visible op surface
from ..FluxKontext import inplace as fk
from ..harness import flux as H
from ..harness import step1x as HS
from .utils import Step1XEditV1P2Manager, ids_gather

gamma = torch.tensor([0.7936, 0.9807, 1.0063, 1.0205, 0.9946, 1.0125, 1.0116, 1.0125, 1.0172,
                      1.0171, 1.0183, 1.0170, 1.0170, 1.0236, 1.0263, 1.0264, 1.0277, 1.0321,
                      1.0338, 1.0361, 1.0396, 1.0454, 1.0492, 1.0566, 1.0696, 1.0879, 1.1179], dtype=torch.float16)

RegionEFlowMatchEulerDiscreteScheduler = fk.RegionEFlowMatchEulerDiscreteScheduler
RegionEStep1XEditAttnProcessor = fk.RegionEFluxAttnProcessor


def warp_modules(pipeline, **args):
    manager = Step1XEditV1P2Manager()
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
    pipeline.__class__ = getattr(pipeline, "_regione_vanilla_class", HS.Step1XEditPipelineV1P2)
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


class RegionEStep1XEditPipeline(HS.Step1XEditPipelineV1P2):

    @torch.no_grad()
    def __call__(self, image=None, prompt_embeds=None, pooled_prompt_embeds=None, negative_prompt_embeds=None,
                 negative_pooled_prompt_embeds=None, height=1024, width=1024, num_inference_steps=28,
                 true_cfg_scale=6.0, guidance_scale=6.0, latents=None, generator=None, output_type="latent",
                 return_dict=True, timesteps_truncate=0.93, process_norm_power=0.4, trace: Optional[dict] = None, sigmas=None,
                 callback_on_step_end=None, callback_on_step_end_tensor_inputs=("latents",)):
        MANAGER = self._regione_manager
        assert num_inference_steps == MANAGER.inference_step, "num_inference_steps must be equal to 28"
        latents, image_latents, latent_ids, text_ids, _, _ = self.prepare(
            image, prompt_embeds, pooled_prompt_embeds, height, width, latents, generator, num_inference_steps, sigmas)
        neg_text_ids = torch.zeros(negative_prompt_embeds.shape[1], 3)
        timesteps = self.scheduler.timesteps
        MANAGER.refresh(latents, image_latents, latent_ids, text_ids, neg_text_ids, 2, self.vae_scale_factor, height, width)
        avd, cache = fk.AvdState(), None
        self.scheduler.set_begin_index(0)
        tr = self.transformer
        if hasattr(tr, "set_vec"):
            tr.set_vec({"cond": pooled_prompt_embeds, "uncond": negative_pooled_prompt_embeds})
        self._precompute(timesteps, None, latents.dtype, pooled_prompt_embeds, negative_pooled_prompt_embeds)
        for i, t in enumerate(timesteps):
            assert i == MANAGER.current_step
            should_cache, ratio = fk.avd_decide(MANAGER, avd, i, timesteps, gamma)
            if should_cache:
                first_hit = cache.shape[1] != latents.shape[1]
                noise_pred = TO.R.avd_apply(cache, float(ratio), MANAGER.edited_ids if first_hit else None, MANAGER.avd_round_ratio)
                if first_hit:
                    cache = ids_gather(cache, MANAGER.edited_ids)
            else:
                x = latents
                if MANAGER.is_full_input_step():
                    x = H.cat_tokens(self.transformer, latents, image_latents)
                timestep = t.expand(latents.shape[0]).to(latents.dtype)
                def branch(embeds, ids, tag):                                                                       # :388-419
                    tr.out_rows_hint = latents.size(1)
                    return tr(hidden_states=x, timestep=timestep / 1000, guidance=None, encoder_hidden_states=embeds,
                              prompt_embeds_mask=None, txt_ids=ids, img_ids=latent_ids, joint_attention_kwargs={"tag": tag},
                              return_dict=False)[0][:, : latents.size(1)]
                full_step = MANAGER.is_full_input_step()
                conc = D.branches_concurrent(MANAGER, (full_step, prompt_embeds.shape[1], negative_prompt_embeds.shape[1]),
                                             not full_step)
                pos, neg = D.run_cfg_branches(getattr(self, "_cfg_pair", None),
                                              lambda: branch(prompt_embeds, text_ids, "cond"),
                                              lambda: branch(negative_prompt_embeds, neg_text_ids, "uncond"), concurrent=conc,
                                              batch_on=tr)
                mode = ops.CFG_STEP1X_RESCALE if float(t) > timesteps_truncate else ops.CFG_PLAIN                    # :421
                noise_pred = TO.R.cfg_combine(pos, neg, true_cfg_scale, mode, process_norm_power)
                cache = noise_pred
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
        if not return_dict:
            return (latents,)
        return HS.Step1XEditPipelineOutput(images=latents)


def RegionEStep1XEditTransformer2DModelforward(self, hidden_states, encoder_hidden_states=None, prompt_embeds_mask=None,
                                               timestep=None, img_ids=None, txt_ids=None, guidance=None,
                                               joint_attention_kwargs=None, return_dict=True, text_embeddings=None,
                                               text_mask=None):
    """Step1XEditV1P2/inplace.py:546-690."""
    MANAGER = self._regione_manager
    image_rotary_emb = fk.dual_rope_tables(self, MANAGER, txt_ids, img_ids)
    tag = (joint_attention_kwargs or {}).get("tag", "cond")
    enc, y = self.branch_inputs(encoder_hidden_states, self._vec[tag], timestep, tag)      # :602-609 (host connector, if any)
    return self._run(hidden_states, enc, y, timestep, None, image_rotary_emb, return_dict, {"tag": tag})
