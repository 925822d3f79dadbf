"""Region utilities + per-pipeline state for FLUX.1-Kontext, HIP-backed.

Mirrors the public names of /root/reference/RegionE/FluxKontext/utils.py (token_selector :282-354,
ids_gather :260-279, ids_scatter :240-257, remove_scattered_points :214-237, FluxKontextManager
:357-465) so code written against the reference keeps working; the arithmetic runs in
libregione_hip.so.  Differences by design:
  * the manager is an object owned by the patched pipeline, not a module-global singleton, so two
    pipelines of the same family can coexist in one process;
  * token ids live on the device; the only host sync of a whole edit is the 4-byte edited-token
    count returned by the partition kernel.
"""
from __future__ import annotations

from typing import List, Optional

import torch

from .. import ops
from .. import torch_ops as TO          # TO.R = torch.ops.regione_mi: the dispatcher-visible op surface


def ids_gather(latent: torch.Tensor, ids: torch.Tensor, rope: bool = False, condition_length=None) -> torch.Tensor:
    """out[b,k,:] = latent[b, ids[b,k], :]  (reference utils.py:260-279)."""
    return TO.R.gather_rows(latent, ids)


def ids_scatter(gathered_latent: torch.Tensor, ids: torch.Tensor, src: torch.Tensor) -> torch.Tensor:
    """src[b, ids[b,k], :] = gathered_latent[b,k,:] in place; returns src (reference utils.py:240-257)."""
    TO.R.scatter_rows_(gathered_latent, ids, src)
    return src


def remove_scattered_points(binary_matrix: torch.Tensor, kernel_size: int = 3, kernel_type: str = "square") -> torch.Tensor:
    """3x3-cross erosion then 5x5-square dilation; like the reference the size arguments are ignored
    (utils.py:228-229).  Accepts / returns a float 0/1 matrix [H, W]."""
    _, _, out = ops.morph_compact(binary_matrix.to(torch.uint8).contiguous(), True)
    return out.to(binary_matrix.dtype)


def token_selector(tensor1, tensor2, k, similarity_type="cosine", height=-1, width=-1, erosion_dilation=False,
                   kernel_size=5, kernel_type="square", patch_size=2, vae_scale_factor=8):
    """Adaptive Region Partition (reference utils.py:282-354).  Only the 'cosine' similarity is
    implemented - it is the only one any reference caller uses (inplace.py:651)."""
    if similarity_type != "cosine":
        raise ValueError("regione_amd implements similarity_type='cosine' only (the only one the reference uses)")
    h_tok, w_tok = height // (patch_size * vae_scale_factor), width // (patch_size * vae_scale_factor)
    if not erosion_dilation and (h_tok * w_tok != tensor1.shape[1]):
        h_tok, w_tok = 1, tensor1.shape[1]
    e, u, _ = TO.R.arp_partition(tensor1, None, tensor2, 0.0, k, h_tok, w_tok, erosion_dilation)
    return e, u


class FluxKontextManager:
    """Sequence-length state machine of one edit (reference utils.py:357-465)."""

    def __init__(self) -> None:
        self.patch_size = 2
        self.vae_scale_factor = 8
        self.inference_step = 28
        self.txt_length = None
        self.height = None
        self.width = None
        self.latent_length = 0
        self.condition_latent = None
        self.condition_length = 0
        self.latent_ids = None
        self.warmup_step = 8
        self.post_step = 0
        self.erosion_dilation = False
        self.threshold = None
        self.cache_threshold = 0
        self.refresh_step: List[int] = []
        self.current_step = 0
        self.edited_ids = None
        self.unedited_ids = None
        self.unedited_latent = None
        self.prev_refresh_step = None
        self.next_refresh_step = None
        self.refresh_step_real_time: List[int] = []
        # device-side helpers (not in the reference)
        self.edited_mask = None          # u8 [L]: 1 = edited (drives the gather-free split Euler step)
        self.sel_rows = None             # i64 [T + K_e]: cache rows a region step rewrites
        self.image_rotary_emb = None     # (cos, sin) for the FULL id table
        self.rope_q_region = None        # (cos, sin) rows of the compacted query set
        self.strict_reference = False    # share one K/V cache between CFG branches like the reference (A-4)
        self.gamma = None                # caller-provided AVD decay table (extension: num_inference_steps != 28)

    def set_parameters(self, args) -> None:
        # reference: `num_inference_steps == 28` always (utils.py:391, the shipped gamma tables have 27 entries).
        # Extension: a caller that brings its own table (`gamma`, N-1 entries; tools/fit_gamma.py fits one) may use N != 28.
        self.gamma = args.get("gamma")
        assert args["warmup_step"] >= 1 and (args["num_inference_steps"] == 28 or self.gamma is not None), \
            "Changing the inference step requires fitting a new gamma"
        if self.gamma is not None:
            self.gamma = torch.as_tensor(self.gamma, dtype=torch.float16).cpu()
            assert self.gamma.numel() == args["num_inference_steps"] - 1, "gamma must have num_inference_steps - 1 entries"
        # Extension `gpu_eager_scalars` (default False = what the CPU-generated fixtures pin): torch's DEVICE kernels cast a
        # 0-dim fp32 tensor operand to the bf16 of the other operand before `cache * ratio` (inplace.py:318), its CPU kernels
        # keep the fp32 scalar.  True reproduces the reference as it runs on a GPU (cache-served velocities differ by up to
        # 2^-9 relative between the two); the Euler update already follows the device behaviour on both (quirk A-2).
        self.avd_round_ratio = bool(args.get("gpu_eager_scalars", False))
        self.inference_step = args["num_inference_steps"]
        self.warmup_step = args["warmup_step"]
        self.post_step = args["post_step"]
        self.threshold = args["threshold"]
        self.cache_threshold = args["cache_threshold"]
        self.erosion_dilation = args["erosion_dilation"]
        self.refresh_step = sorted(int(item) for item in args["refresh_step"].split(","))
        assert min(self.refresh_step) > self.warmup_step + 1 and \
            max(self.refresh_step) <= self.inference_step - self.post_step - 1
        assert not any(abs(a - b) == 1 for a, b in zip(self.refresh_step, self.refresh_step[1:])), \
            "Refresh steps must not be adjacent."
        self.refresh_step.append(self.inference_step - self.post_step + 1)

    # ---------------------------------------------------------------------------------------------
    def set_partition(self, edited_ids: torch.Tensor, unedited_ids: torch.Tensor, mask: torch.Tensor):
        """Called by the scheduler at step warmup-1 with the partition kernel's outputs."""
        self.edited_ids, self.unedited_ids, self.edited_mask = edited_ids, unedited_ids, mask
        self._ids_edited_host = None
        self._sel_by_T, self._ropeq_by_T = {}, {}
        self._branch_warm = set()        # the lazily built per-text-length tables are gone: next forward pair runs in order
        self.sel_rows = self.sel_rows_for(self.txt_length)
        if self.image_rotary_emb is not None:
            self.rope_q_region = self.rope_q_for(self.txt_length, self.image_rotary_emb)

    # per text length (the cond / uncond branches of Step1X-v1p2 and Qwen have different T:
    # `selection` of Step1XEditV1P2/inplace.py:833,868)
    def sel_rows_for(self, T: int) -> torch.Tensor:
        if T not in self._sel_by_T:
            self._sel_by_T[T] = ops.sel_rows(self.edited_ids.squeeze(0), T)          # rgn_sel_rows: one launch
        return self._sel_by_T[T]

    def rope_q_for(self, T: int, full_table):
        if T not in self._ropeq_by_T:
            self._ropeq_by_T[T] = tuple(TO.R.gather_rows(t, self.sel_rows_for(T)) for t in full_table)
        return self._ropeq_by_T[T]

    def _compact(self, latent, latent_ids):
        self.unedited_latent = ids_gather(latent, self.unedited_ids)
        latent = ids_gather(latent, self.edited_ids)
        if self._ids_edited_host is None:           # host id table only feeds shape logic + RoPE cache keys
            self._ids_edited_host = latent_ids[self.edited_ids.squeeze(0).cpu()]
        return latent, self._ids_edited_host

    def _restore(self, latent):
        full = torch.empty_like(self.condition_latent)   # every row is written by the two scatters
        ids_scatter(latent, self.edited_ids, full)
        ids_scatter(self.unedited_latent, self.unedited_ids, full)
        return full, self.latent_ids

    def step(self, latent, latent_ids):
        self.current_step += 1
        if self.current_step == self.warmup_step:
            latent, latent_ids = self._compact(latent, latent_ids)
        elif self.current_step == self.inference_step - self.post_step:
            latent, latent_ids = self._restore(latent)
            self.prev_refresh_step = None
        elif self.prev_refresh_step is not None and self.current_step == self.prev_refresh_step:
            latent, latent_ids = self._restore(latent)
        elif self.prev_refresh_step is not None and self.current_step == self.prev_refresh_step + 1:
            latent, latent_ids = self._compact(latent, latent_ids)
            self.prev_refresh_step = self.next_refresh_step
        return latent, latent_ids

    def refresh(self, latents, image_latents, latent_ids, text_ids, patch_size=2, vae_scale_factor=8, height=None,
                width=None) -> None:
        self.width, self.height = width, height
        self.patch_size, self.vae_scale_factor = patch_size, vae_scale_factor
        self.latent_length = latents.size(1)
        self.txt_length = text_ids.size(0)
        self.condition_latent = image_latents
        self.condition_length = image_latents.size(1) if image_latents is not None else 0
        self.current_step = 0
        self.prev_refresh_step = None
        self.next_refresh_step = None
        self.edited_ids = self.unedited_ids = self.unedited_latent = None
        self.edited_mask = self.sel_rows = self.rope_q_region = None
        self.image_rotary_emb = None
        self.rope_full_by_T = {}         # text length -> (cos, sin) of [text ids ; FULL latent ids]
        self._sel_by_T, self._ropeq_by_T = {}, {}
        self._branch_warm = set()         # regione_amd.dist.branches_concurrent: step kinds whose tables exist
        self.latent_ids = latent_ids
        self.refresh_step_real_time = list(self.refresh_step)

    # phase predicates (inplace.py:331, :717-732) ---------------------------------------------------
    def is_full_input_step(self) -> bool:
        c = self.current_step
        return c <= self.warmup_step - 1 or c > self.inference_step - self.post_step - 1 or c == self.prev_refresh_step

    def kv_phase(self) -> str:
        c = self.current_step
        if c < self.warmup_step - 1 or c > self.inference_step - self.post_step - 1:
            return "plain"
        if c == self.warmup_step - 1 or c == self.prev_refresh_step:
            return "store"
        return "update"
