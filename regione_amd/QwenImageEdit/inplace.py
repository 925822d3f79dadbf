"""RegionE patch set for Qwen-Image-Edit on the HIP kernels.

Mirrors /root/reference/RegionE/QwenImageEdit/inplace.py (`warp_modules` :53-62, `__call__` :66-460,
forward :462-571, scheduler :574-690, `QwenDoubleStreamAttnProcessor2_0` :731-890).  Family deltas:
  * double-stream blocks only (no single blocks, :58-59); 1-D latent ids (:322);
  * sequential CFG with 'cond' / 'uncond' tags -> `k_cache_even/odd` (:731-734, :756-815);
  * norm-preserving CFG (:401-405) -> rgn_cfg_combine(mode 2); Qwen gamma table (:47-50);
  * rotary tables: query rows use the current ids, key rows the full id table (:531, :847-855).
"""
from __future__ import annotations

from typing import Optional

import torch

from .. import dist as D
from .. import ops
from .. import torch_ops as TO          # TO.R = torch.ops.regione_mi: the dispatcher-visible op surface
from ..FluxKontext import inplace as fk
from ..harness import flux as H
from ..harness import qwen as HQ
from .utils import QwenImageEditManager, ids_gather

gamma = torch.tensor([1.0195, 1.0233, 1.0243, 1.0185, 1.0321, 1.0208, 1.0260, 1.0233, 1.0258,
                      1.0292, 1.0316, 1.0306, 1.0289, 1.0347, 1.0329, 1.0402, 1.0378, 1.0384,
                      1.0413, 1.0444, 1.0526, 1.0400, 1.0555, 1.0439, 1.0357, 1.0118, 0.7603], dtype=torch.float16)

RegionEFlowMatchEulerDiscreteScheduler = fk.RegionEFlowMatchEulerDiscreteScheduler


class QwenDoubleStreamAttnProcessor2_0(fk.RegionEFluxAttnProcessor):
    """Region-Instruction KV cache for the joint double-stream attention; one cache per CFG tag."""

    def __init__(self, manager=None):
        super().__init__(False, manager)


def warp_modules(pipeline, pipeline_cls=None, **args):
    manager = QwenImageEditManager()
    manager.set_parameters(dict(args))
    pipeline._regione_manager = manager
    pipeline._regione_vanilla_class = pipeline.__class__
    pipeline.__class__ = pipeline_cls or RegionEQwenImageEditPipeline
    sch = RegionEFlowMatchEulerDiscreteScheduler.from_config(pipeline.scheduler.config)
    sch.manager = manager
    pipeline.scheduler = sch
    tr = pipeline.transformer
    tr._regione_manager = manager
    tr.forward = RegionEQwenImageTransformer2DModelforward.__get__(tr, tr.__class__)
    for block in tr.transformer_blocks:
        block.attn.set_processor(QwenDoubleStreamAttnProcessor2_0(manager))
    return pipeline


def unwarp_modules(pipeline):
    pipeline.__class__ = getattr(pipeline, "_regione_vanilla_class", HQ.QwenImageEditPipeline)
    pipeline.scheduler = H.FlowMatchEulerDiscreteScheduler.from_config(pipeline.scheduler.config)
    tr = pipeline.transformer
    if "forward" in tr.__dict__:
        del tr.__dict__["forward"]
    for block in tr.transformer_blocks:
        block.attn.set_processor(HQ.QwenDoubleStreamAttnProcessor2_0())
    pipeline._regione_manager = None
    return pipeline


class RegionEQwenImageEditPipeline(HQ.QwenImageEditPipeline):
    gamma = gamma

    @torch.no_grad()
    def __call__(self, image=None, prompt_embeds=None, negative_prompt_embeds=None, height=1024, width=1024,
                 num_inference_steps=28, true_cfg_scale=4.0, latents=None, generator=None, output_type="latent",
                 return_dict=True, trace: Optional[dict] = None, cond_shapes=None, sigmas=None,
                 callback_on_step_end=None, callback_on_step_end_tensor_inputs=("latents",)):
        MANAGER = self._regione_manager
        assert num_inference_steps == MANAGER.inference_step, "num_inference_steps should be equal to 28"
        latents, image_latents, latent_ids = self.prepare_qwen(image, height, width, latents, generator, num_inference_steps,
                                                               cond_shapes, sigmas)
        timesteps = self.scheduler.timesteps
        img_shapes = self._shapes(height, width, cond_shapes)
        do_true_cfg = true_cfg_scale > 1 and negative_prompt_embeds is not None           # :238
        # The partition compares the one-step estimate with the condition latent token by token (utils.py:310-312), which
        # needs L_c == L.  With several condition images (2509) the reference's own call breaks there (shape mismatch);
        # here the LAST image - the one that fixes the output size, QwenImageEditPlus/inplace.py:190 - is the reference
        # image of the partition, and every image stays in the K/V cache.
        arp_cond = image_latents
        if image_latents.shape[1] != latents.shape[1]:
            arp_cond = image_latents[:, image_latents.shape[1] - latents.shape[1]:]
            assert cond_shapes is not None and int(cond_shapes[-1][0]) * int(cond_shapes[-1][1]) == latents.shape[1], \
                "the last condition image must have the output token grid"
        MANAGER.refresh(latents, arp_cond, latent_ids, 2, self.vae_scale_factor, height, width)
        MANAGER.txt_length = prompt_embeds.shape[1]
        avd, cache = fk.AvdState(), None
        self.scheduler.set_begin_index(0)
        self._precompute(timesteps, None, latents.dtype)
        tr = self.transformer
        for i, t in enumerate(timesteps):
            assert i == MANAGER.current_step
            should_cache, ratio = fk.avd_decide(MANAGER, avd, i, timesteps, self.gamma)      # :332-350
            if should_cache:                                                             # :352-356
                first_hit = cache.shape[1] != latents.shape[1]
                noise_pred = TO.R.avd_apply(cache, float(ratio), MANAGER.edited_ids if first_hit else None, MANAGER.avd_round_ratio)
                if first_hit:
                    cache = ids_gather(cache, MANAGER.edited_ids)
            else:
                x = latents
                if MANAGER.is_full_input_step():                                         # :364-365
                    x = H.cat_tokens(self.transformer, latents, image_latents)
                timestep = t.expand(latents.shape[0]).to(latents.dtype)
                def branch(embeds, tag):
                    tr.out_rows_hint = latents.size(1)
                    return tr(hidden_states=x, timestep=timestep / 1000, encoder_hidden_states=embeds, img_shapes=img_shapes,
                              latent_ids=latent_ids, attention_kwargs={"tag": tag}, return_dict=False)[0][:, : latents.size(1)]
                if do_true_cfg:                                                          # :371-405
                    full_step = MANAGER.is_full_input_step()
                    conc = D.branches_concurrent(MANAGER, (full_step, prompt_embeds.shape[1], negative_prompt_embeds.shape[1]),
                                                 not full_step)
                    noise_pred, neg = D.run_cfg_branches(getattr(self, "_cfg_pair", None),
                                                         lambda: branch(prompt_embeds, "cond"),
                                                         lambda: branch(negative_prompt_embeds, "uncond"), concurrent=conc,
                                                         batch_on=tr)
                    if trace is not None:            # the two branch velocities before the combine (parity attribution tools)
                        trace.setdefault("branches", {})[i] = (noise_pred.clone(), neg.clone())
                    noise_pred = TO.R.cfg_combine(noise_pred, neg, true_cfg_scale, ops.CFG_QWEN_NORM)
                else:
                    noise_pred = branch(prompt_embeds, "cond")
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
        return HQ.QwenImagePipelineOutput(images=latents)


def RegionEQwenImageTransformer2DModelforward(self, hidden_states, encoder_hidden_states=None,
                                              encoder_hidden_states_mask=None, timestep=None, img_shapes=None,
                                              txt_seq_lens=None, guidance=None, attention_kwargs=None, latent_ids=None,
                                              return_dict=True):
    """QwenImageEdit/inplace.py:462-571."""
    MANAGER = self._regione_manager
    T = encoder_hidden_states.shape[1]
    rope_q = fk.dual_rope_tables(self, MANAGER, T, latent_ids, build_full=lambda t: self.full_rope(img_shapes[0], t))
    return self._run(hidden_states, encoder_hidden_states, None, timestep, None, rope_q, return_dict, attention_kwargs)
