"""RegionE patch set for FLUX.1-Kontext on the HIP kernels.

Same public surface as /root/reference/RegionE/FluxKontext/inplace.py: `warp_modules`,
`unwarp_modules` (:53-73), `RegionEFluxKontextPipeline.__call__` (:76-410),
`RegionEFluxTransformer2DModelforward` (:413-576), `RegionEFlowMatchEulerDiscreteScheduler.step`
(:579-691) and `RegoionEFluxAttnProcessor2_0` (:694-824, the reference's spelling is kept as an
alias).  What changes is *where the work runs*:

  reference                                   here
  -----------------------------------------   ------------------------------------------------------
  AVD decision on device tensors, 2 syncs/    host fp32 arithmetic with the same dtype path
  step (:295-313)                             (fp16 gamma x fp32), zero syncs
  cache * ratio + optional gather (:315-318)  rgn_avd_apply (one launch, gather fused)
  ARP: ~10 torch kernels + nonzero sync       rgn_arp_partition (2 launches, one 4-byte D2H)
  split Euler: 4 gathers + 2 scatters +       rgn_euler_step with the edited mask (one pass)
  zeros + 2 axpy (:648-677)
  _partially_linear x2 (Triton) + RMSNorm +   fused [k|v|q(|mlp)] MFMA GEMM -> rgn_qk_norm_rope_store
  RoPE + cat over ALL T+N rows (:734-794)     rewriting only rows sel_rows of the post-norm cache
  flash_attn_func (:796-801)                  rgn_attention over (T+K_e) x (T+N)
"""
from __future__ import annotations

from typing import Optional

import torch

from .. import ops
from .. import dist as D
from .. import torch_ops as TO          # TO.R = torch.ops.regione_mi: the dispatcher-visible op surface
from ..harness import flux as H
from .utils import FluxKontextManager, ids_gather

# fitted decay factors, fp16 like the reference (inplace.py:47-50)
gamma = torch.tensor([0.8352, 0.9986, 1.0090, 1.0097, 1.0161, 1.0152, 1.0160, 1.0173, 1.0177,
                      1.0199, 1.0213, 1.0203, 1.0257, 1.0236, 1.0235, 1.0278, 1.0302, 1.0311,
                      1.0352, 1.0371, 1.0391, 1.0459, 1.0498, 1.0581, 1.0693, 1.0866, 1.1090],
                     dtype=torch.float16)


def warp_modules(pipeline, **args):
    """inplace.py:53-62.  The manager is attached to the pipeline instead of being a module global.
    Extra (non-reference) key `strict_reference`: share ONE K/V cache between the cond / uncond
    forwards of FLUX true-CFG exactly like the reference does (quirk A-4); default False = one cache
    per branch, the fix Qwen / Step1X-v1p2 apply (QwenImageEdit/inplace.py:731-734)."""
    args = dict(args)
    strict = bool(args.pop("strict_reference", False))
    manager = FluxKontextManager()
    manager.set_parameters(args)
    manager.strict_reference = strict
    pipeline._regione_manager = manager
    pipeline._regione_vanilla_class = pipeline.__class__
    pipeline.__class__ = RegionEFluxKontextPipeline
    sch = RegionEFlowMatchEulerDiscreteScheduler.from_config(pipeline.scheduler.config)
    sch.manager = manager
    pipeline.scheduler = sch
    tr = pipeline.transformer
    tr._regione_manager = manager
    tr.forward = RegionEFluxTransformer2DModelforward.__get__(tr, tr.__class__)
    for block in tr.transformer_blocks:
        block.attn.set_processor(RegionEFluxAttnProcessor(False, manager))
    for block in tr.single_transformer_blocks:
        block.attn.set_processor(RegionEFluxAttnProcessor(True, manager))
    return pipeline


def unwarp_modules(pipeline):
    """inplace.py:65-73."""
    pipeline.__class__ = getattr(pipeline, "_regione_vanilla_class", H.FluxKontextPipeline)
    pipeline.scheduler = H.FlowMatchEulerDiscreteScheduler.from_config(pipeline.scheduler.config)
    tr = pipeline.transformer
    if "forward" in tr.__dict__:
        del tr.__dict__["forward"]
    for block in tr.transformer_blocks:
        block.attn.set_processor(H.FluxAttnProcessor(False))
    for block in tr.single_transformer_blocks:
        block.attn.set_processor(H.FluxAttnProcessor(True))
    pipeline._regione_manager = None
    return pipeline


class AvdState:
    """`cache, should_cache, accumulate, error` of inplace.py:288."""

    def __init__(self):
        self.accumulate = 1
        self.should_cache = False


def avd_decide(M: FluxKontextManager, avd: AvdState, i: int, timesteps: torch.Tensor, gamma: torch.Tensor = gamma):
    """inplace.py:295-313 on HOST tensors: gamma[i-1] is a 0-dim fp16 tensor, timesteps are fp32, so
    `ratio` is fp32 and `accumulate` is carried in fp32 - the same dtype path as the reference's
    device tensors, minus the two implicit device->host syncs per step (quirk A-7)."""
    if getattr(M, "gamma", None) is not None:      # caller-provided table (extension, num_inference_steps != 28)
        gamma = M.gamma
    ratio = None
    if M.current_step <= M.warmup_step or M.current_step > M.inference_step - M.post_step - 1 or \
            M.current_step == M.prev_refresh_step:
        avd.should_cache, avd.accumulate = False, 1
    else:
        ratio = gamma[i - 1] * (1 + (timesteps[i] - timesteps[i - 1]) / 1000)
        if ratio >= 1:
            avd.should_cache, avd.accumulate = False, 1
        else:
            avd.accumulate = avd.accumulate * ratio
            error = 1 - avd.accumulate
            if error > M.cache_threshold:
                avd.should_cache, avd.accumulate = False, 1
            else:
                avd.should_cache = True
    return avd.should_cache, ratio


class RegionEFluxKontextPipeline(H.FluxKontextPipeline):

    @torch.no_grad()
    def __call__(self, image=None, prompt_embeds=None, pooled_prompt_embeds=None, height=1024, width=1024,
                 num_inference_steps=28, guidance_scale=2.5, latents=None, generator=None, output_type="latent",
                 return_dict=True, callback_on_step_end=None, true_cfg_scale: float = 1.0,
                 negative_prompt_embeds=None, negative_pooled_prompt_embeds=None, trace: Optional[dict] = None, sigmas=None,
                 callback_on_step_end_tensor_inputs=("latents",)):
        MANAGER: FluxKontextManager = self._regione_manager
        assert num_inference_steps == MANAGER.inference_step, "num_inference_steps should be equal to 28"
        latents, image_latents, latent_ids, text_ids, _, _ = self.prepare(
            image, prompt_embeds, pooled_prompt_embeds, height, width, latents, generator, num_inference_steps, sigmas)
        timesteps = self.scheduler.timesteps
        guidance = torch.full([1], guidance_scale, dtype=torch.float32)
        MANAGER.refresh(latents, image_latents, latent_ids, text_ids, 2, self.vae_scale_factor, height, width)
        avd, cache = AvdState(), None
        do_true_cfg = true_cfg_scale > 1 and negative_prompt_embeds is not None and negative_pooled_prompt_embeds is not None
        self.scheduler.set_begin_index(0)
        self._precompute(timesteps, guidance, latents.dtype, pooled_prompt_embeds,
                         negative_pooled_prompt_embeds if do_true_cfg else None)
        for i, t in enumerate(timesteps):
            assert i == MANAGER.current_step
            should_cache, ratio = avd_decide(MANAGER, avd, i, timesteps)
            if should_cache:                                                    # inplace.py:315-318
                first_hit = cache.shape[1] != latents.shape[1]
                noise_pred = TO.R.avd_apply(cache, float(ratio), MANAGER.edited_ids if first_hit else None, MANAGER.avd_round_ratio)
                if first_hit:
                    cache = ids_gather(cache, MANAGER.edited_ids)
            else:
                latent_model_input = latents
                if MANAGER.is_full_input_step():                                # inplace.py:331-332
                    latent_model_input = H.cat_tokens(self.transformer, latents, image_latents)
                timestep = t.expand(latents.shape[0]).to(latents.dtype)

                def branch(embeds, pooled_e, tag):
                    self.transformer.out_rows_hint = latents.size(1)
                    return self.transformer(hidden_states=latent_model_input, timestep=timestep / 1000, guidance=guidance,
                                            pooled_projections=pooled_e, encoder_hidden_states=embeds, txt_ids=text_ids,
                                            img_ids=latent_ids, joint_attention_kwargs={"tag": tag},
                                            return_dict=False)[0][:, : latents.size(1)]
                if do_true_cfg:                                                 # inplace.py:349-364
                    # two forwards per computed step: side by side on two streams in region steps, unless the reference's
                    # shared K/V cache (strict_reference, quirk A-4) makes the second forward depend on the first
                    full_step = MANAGER.is_full_input_step()
                    conc = (not getattr(MANAGER, "strict_reference", False)) and D.branches_concurrent(
                        MANAGER, (full_step, prompt_embeds.shape[1], negative_prompt_embeds.shape[1]), not full_step)
                    noise_pred, neg = D.run_cfg_branches(None, lambda: branch(prompt_embeds, pooled_prompt_embeds, "cond"),
                                                         lambda: branch(negative_prompt_embeds, negative_pooled_prompt_embeds, "uncond"),
                                                         concurrent=conc,
                                                         batch_on=None if getattr(MANAGER, "strict_reference", False) else self.transformer)
                    noise_pred = TO.R.cfg_combine(noise_pred, neg, true_cfg_scale, ops.CFG_PLAIN)
                else:
                    noise_pred = branch(prompt_embeds, pooled_prompt_embeds, "cond")
                cache = noise_pred                                              # inplace.py:365
            if trace is not None:
                trace.setdefault("kind", []).append("C" if should_cache else ("F" if MANAGER.is_full_input_step() else "R"))
                trace.setdefault("noise_pred", []).append(noise_pred.clone())
            latents = self.scheduler.step(noise_pred, t, latents, return_dict=False)[0]
            latents, prompt_embeds = self._callback(callback_on_step_end, callback_on_step_end_tensor_inputs, i, t, latents,
                                                    prompt_embeds, noise_pred=noise_pred, image_latents=image_latents,
                                                    negative_prompt_embeds=negative_prompt_embeds)                  # inplace.py:376-383
            latents, latent_ids = MANAGER.step(latents, latent_ids)
            if trace is not None:
                trace.setdefault("latents", []).append(latents.clone())
                trace.setdefault("prev_refresh", []).append(MANAGER.prev_refresh_step)
        if not return_dict:
            return (latents,)
        return H.FluxPipelineOutput(images=latents)


def dual_rope_tables(transformer, MANAGER, txt_ids, img_ids, build_full=None):
    """Query-row rotary table for this forward; also makes sure the FULL-id key table of this text
    length exists (MANAGER.image_rotary_emb of the reference, inplace.py:495-500).  `build_full(T)`
    overrides the FLUX table builder (Qwen: QwenImageEdit/inplace.py:531)."""
    T = txt_ids.shape[0] if hasattr(txt_ids, "shape") else int(txt_ids)
    if T not in MANAGER.rope_full_by_T:
        MANAGER.rope_full_by_T[T] = build_full(T) if build_full is not None else transformer.pos_embed(
            torch.cat((txt_ids.cpu(), MANAGER.latent_ids.cpu()), dim=0), transformer.device)
    full = MANAGER.rope_full_by_T[T]
    if MANAGER.image_rotary_emb is None:
        MANAGER.image_rotary_emb = full
    if img_ids.shape[0] == MANAGER.latent_ids.shape[0]:
        return full
    return MANAGER.rope_q_for(T, full)               # rows of the full table at [text ; edited ids]


def RegionEFluxTransformer2DModelforward(self, hidden_states, encoder_hidden_states=None, pooled_projections=None,
                                         timestep=None, img_ids=None, txt_ids=None, guidance=None,
                                         joint_attention_kwargs=None, return_dict=True):
    """inplace.py:413-576: identical to the vanilla forward except that TWO rotary tables exist -
    one for the query rows (current, possibly compacted ids) and one for the keys (always the full id
    table, `MANAGER.image_rotary_emb`, :495-500)."""
    MANAGER: FluxKontextManager = self._regione_manager
    image_rotary_emb = dual_rope_tables(self, MANAGER, txt_ids, img_ids)
    return self._run(hidden_states, encoder_hidden_states, pooled_projections, timestep, guidance, image_rotary_emb,
                     return_dict, joint_attention_kwargs)


class RegionEFlowMatchEulerDiscreteScheduler(H.FlowMatchEulerDiscreteScheduler):
    manager: FluxKontextManager = None

    def step(self, model_output, timestep, sample, return_dict: bool = True, **kw):
        """inplace.py:581-691 (non-stochastic path).  dt's are host fp32 differences of `sigmas`."""
        if isinstance(timestep, int) or (isinstance(timestep, torch.Tensor) and not timestep.is_floating_point()):
            raise ValueError("Passing integer indices (e.g. from `enumerate(timesteps)`) as timesteps to"
                             " `FlowMatchEulerDiscreteScheduler.step()` is not supported.")
        if self._step_index is None:
            self._init_step_index(timestep)
        MANAGER = self.manager
        s = self.sigmas
        sigma, sigma_next = s[self._step_index], s[self._step_index + 1]
        dt = float(sigma_next - sigma)
        cur = MANAGER.current_step
        if cur == MANAGER.warmup_step - 1:                                        # :630-634, :648-663
            MANAGER.prev_refresh_step = MANAGER.refresh_step_real_time.pop(0) - 1
            dt_final = float(s[-1] - sigma)
            dt_direct = float(s[MANAGER.prev_refresh_step] - sigma)
            e, u, mask = TO.R.arp_partition(
                sample, model_output, MANAGER.condition_latent, dt_final, MANAGER.threshold,
                MANAGER.height // (MANAGER.patch_size * MANAGER.vae_scale_factor),
                MANAGER.width // (MANAGER.patch_size * MANAGER.vae_scale_factor), MANAGER.erosion_dilation)
            MANAGER.set_partition(e, u, mask)
            prev = TO.R.split_euler_step(sample, model_output, dt, MANAGER.edited_mask, dt_direct)
        elif MANAGER.prev_refresh_step is not None and cur == MANAGER.prev_refresh_step:   # :636-639, :665-677
            if len(MANAGER.refresh_step_real_time) != 0:
                MANAGER.next_refresh_step = MANAGER.refresh_step_real_time.pop(0) - 1
            dt_direct = float(s[MANAGER.next_refresh_step] - sigma)
            prev = TO.R.split_euler_step(sample, model_output, dt, MANAGER.edited_mask, dt_direct)
        else:
            prev = TO.R.split_euler_step(sample, model_output, dt)                       # :680
        self._step_index += 1
        return (prev,) if not return_dict else H._Cfg(prev_sample=prev)


class RegionEFluxAttnProcessor(H.FluxAttnProcessor):
    """Region-Instruction KV-cache protocol (inplace.py:694-824).  The cache of a layer is one K slab
    [skv_pad, d] + one V^T slab [d, skv_pad] covering [text rows ; all image rows], one pair per CFG
    branch tag (`k_cache_even/odd` of Step1XEditV1P2/inplace.py:800-888, QwenImageEdit/inplace.py:731)."""

    def __init__(self, single: bool, manager: Optional[FluxKontextManager] = None):
        super().__init__(single)
        self.manager = manager
        self.caches = {}                 # tag -> (k_slab, vt_slab, skv)

    @property
    def k_cache(self):
        c = next(iter(self.caches.values()), None)
        return None if c is None else c[0]

    @property
    def v_cache(self):
        c = next(iter(self.caches.values()), None)
        return None if c is None else c[1]

    def kv_target(self, attn, ctx):
        MANAGER, ws = self.manager, ctx.ws
        phase = MANAGER.kv_phase()
        if phase == "plain":                                                      # :717-719
            return ws.k_scratch, ws.vt_scratch, None, ctx.T + ctx.M, None
        tag = None if getattr(MANAGER, "strict_reference", False) else ctx.tag
        d = attn.heads * attn.head_dim
        if phase == "store":                                                      # :721-725
            skv = ctx.T + ctx.M
            pad = ops.padded(skv)
            c = self.caches.get(tag)
            if c is None or c[0].shape[0] != pad:
                # a slab outlives the forward that creates it and is read by both streams later: allocate it under the
                # DEFAULT stream of the device, so the caching allocator never ties its block to a side stream
                cur = torch.cuda.current_stream(ws.device)
                dflt = torch.cuda.default_stream(ws.device)
                with torch.cuda.stream(dflt):
                    c = (ops.zeros((pad, d), dtype=torch.bfloat16, device=ws.device),
                         ops.zeros((d, pad), dtype=torch.bfloat16, device=ws.device), skv)
                cur.wait_stream(dflt)            # the zero fill is ordered before this forward's writes
                if cur != dflt:
                    c[0].record_stream(cur)
                    c[1].record_stream(cur)
            self.caches[tag] = (c[0], c[1], skv)
            return c[0], c[1], None, skv, None
        # update: only rows [text ; T + edited_ids] are recomputed (:727-750)
        k, v, skv = self.caches[tag]
        return k, v, MANAGER.sel_rows_for(ctx.T), skv, MANAGER.rope_full_by_T[ctx.T]


RegoionEFluxAttnProcessor2_0 = RegionEFluxAttnProcessor      # the reference's (misspelled) class name
