"""CPU oracle for RegionE's region-aware denoising hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under regione_amd/ imports this file; only tests/,
__graft_entry__.smoke() and bench.py's `cpu_baseline` leg may, and only as the checker.

It is a torch-CPU / numpy *restatement* (own code, functional style) of the algorithm in
/root/reference/RegionE/FluxKontext/{utils,inplace,fused_kernels}.py (the other families differ
only where noted).  Each function cites the reference lines it follows.

Parity status
-------------
* Everything RegionE itself authored (partition, morphology, gather/scatter, manager state
  machine, split-Euler scheduler step, AVD decision, K/V-cache protocol, denoise loop, dual RoPE
  tables) is PINNED: tests/test_oracle_golden.py checks this file bit-for-bit against fixtures in
  tests/golden/ that tools/gen_golden.py produced by importing and running the reference itself
  in the build container.
* The MMDiT block arithmetic (AdaLN-Zero, RMSNorm, RoPE, FeedForward, scheduler base) lives in an
  un-vendored diffusers fork (`git+https://github.com/Peyton-Chen/diffusers.git@step1xedit_v1p2`,
  reference README.md:77, no commit pin) that is absent from /root/reference and from this image.
  Those functions restate upstream diffusers semantics and are marked [EXT]:
  **parity unpinned** by the reference for that part.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

# fitted decay factors, one table per model family, fp16 like the reference
# (FluxKontext/inplace.py:47-50, Step1XEdit/inplace.py, QwenImageEdit/inplace.py).
GAMMA = {
    "flux": [0.8352, 0.9986, 1.0090, 1.0097, 1.0161, 1.0152, 1.0160, 1.0173, 1.0177,
             1.0199, 1.0213, 1.0203, 1.0257, 1.0236, 1.0235, 1.0278, 1.0302, 1.0311,
             1.0352, 1.0371, 1.0391, 1.0459, 1.0498, 1.0581, 1.0693, 1.0866, 1.1090],
    # Step1XEdit/inplace.py:47-49
    "step1x": [0.9746, 0.9593, 1.0036, 1.0084, 1.0106, 1.0114, 1.0138, 1.0163, 1.0152,
               1.0163, 1.0197, 1.0186, 1.0219, 1.0218, 1.0223, 1.0266, 1.0272, 1.0305,
               1.0311, 1.0362, 1.0385, 1.0423, 1.0500, 1.0536, 1.0671, 1.0866, 1.1015],
    # QwenImageEdit/inplace.py:47-50
    "qwen": [1.0195, 1.0233, 1.0243, 1.0185, 1.0321, 1.0208, 1.0260, 1.0233, 1.0258,
             1.0292, 1.0316, 1.0306, 1.0289, 1.0347, 1.0329, 1.0402, 1.0378, 1.0384,
             1.0413, 1.0444, 1.0526, 1.0400, 1.0555, 1.0439, 1.0357, 1.0118, 0.7603],
    # QwenImageEditPlus/inplace.py:47-50
    "qwen_plus": [1.0186, 1.0241, 1.0236, 1.0205, 1.0298, 1.0221, 1.0248, 1.0246, 1.0269,
                  1.0275, 1.0323, 1.0311, 1.0298, 1.0353, 1.0343, 1.0397, 1.0387, 1.0393,
                  1.0404, 1.0458, 1.0507, 1.0418, 1.0518, 1.0426, 1.0311, 1.0068, 0.7628],
    # Step1XEditV1P2/inplace.py:48-50
    "step1x_v1p2": [0.7936, 0.9807, 1.0063, 1.0205, 0.9946, 1.0125, 1.0116, 1.0125, 1.0172,
                    1.0171, 1.0183, 1.0170, 1.0170, 1.0236, 1.0263, 1.0264, 1.0277, 1.0321,
                    1.0338, 1.0361, 1.0396, 1.0454, 1.0492, 1.0566, 1.0696, 1.0879, 1.1179],
}


# --------------------------------------------------------------------------------------
# a1/a2  Adaptive Region Partition
# --------------------------------------------------------------------------------------
def cosine_rows(t1: torch.Tensor, t2: torch.Tensor) -> torch.Tensor:
    """utils.py:310-312.  Each operand is L2-normalised *in its own dtype* (F.normalize,
    eps 1e-12), the product and the row sum are taken in the promoted dtype."""
    return torch.sum(F.normalize(t1, dim=-1) * F.normalize(t2, dim=-1), dim=-1)


def cosine_rows_explicit(t1: torch.Tensor, t2: torch.Tensor) -> torch.Tensor:
    """The same quantity as cosine_rows() for rows of 64 channels with every floating-point rounding and every
    reduction tree written out (numpy, no torch reductions).  torch eager's CPU kernels reduce a contiguous last
    dimension of 64 with FIXED trees; `similarity <= threshold` (utils.py:333) is only reproducible bit for bit by a
    kernel that follows them, so they are pinned here (tests/test_oracle_golden.py checks this function against torch
    and against the reference-generated fixtures, incl. the +-4-ulp adversarial rows of arp_adv.npz):
      * fp32 L2 norm (F.normalize -> linalg_vector_norm): 8 lanes, acc[j] = x[j]^2, acc[j] = fma(x[8b+j], x[8b+j], acc[j])
        for b = 1..7, lanes added left to right, sqrt;
      * bf16 L2 norm: fp32 squares (exact), xor butterfly over 64 lanes with strides 32, 16, 8, 4, 2, 1, sqrt, result
        rounded to bf16; the quotient is rounded to bf16 again;
      * torch.sum of the fp32 products: 8-lane vectors v0..v7, t[k] = v[k] + v[k+4], a = ((t0 + t1) + t2) + t3, lanes
        added left to right."""
    f32, f64 = np.float32, np.float64
    assert t1.dtype == torch.float32 and t1.shape[-1] == 64

    def seq(v):
        acc = v[:, 0].copy()
        for j in range(1, v.shape[1]):
            acc = acc + v[:, j]
        return acc

    def norm_f32(x):
        acc = x[:, :8] * x[:, :8]
        for b in range(1, 8):
            xb = x[:, 8 * b: 8 * b + 8].astype(f64)
            acc = (acc.astype(f64) + xb * xb).astype(f32)       # fused multiply-add (one rounding)
        return np.sqrt(seq(acc))

    def rbf(x):
        return torch.from_numpy(np.ascontiguousarray(x)).to(torch.bfloat16).float().numpy()

    x = t1.reshape(-1, 64).numpy()
    a = x / np.maximum(norm_f32(x), f32(1e-12))[:, None]
    if t2.dtype == torch.bfloat16:
        c = t2.reshape(-1, 64).float().numpy()
        v = c * c
        for o in (32, 16, 8, 4, 2, 1):
            v = v + v[:, np.arange(64) ^ o]
        n2 = np.maximum(rbf(np.sqrt(v[:, 0])), rbf(np.array([1e-12], f32)))
        b = rbf(c / n2[:, None])
    else:
        c = t2.reshape(-1, 64).numpy()
        b = c / np.maximum(norm_f32(c), f32(1e-12))[:, None]
    p = a * b
    t = [p[:, 8 * k: 8 * k + 8] + p[:, 8 * (k + 4): 8 * (k + 4) + 8] for k in range(4)]
    return torch.from_numpy(seq(((t[0] + t[1]) + t[2]) + t[3])).reshape(t1.shape[:-1])


def erode_cross3(mask: np.ndarray) -> np.ndarray:
    """utils.py:152-179 with the 3x3 cross kernel of utils.py:228: a pixel survives iff it and
    its 4-neighbours are all set; zero padding => the outermost ring never survives."""
    m = np.pad(mask.astype(np.int32), 1)
    s = m[1:-1, 1:-1] + m[:-2, 1:-1] + m[2:, 1:-1] + m[1:-1, :-2] + m[1:-1, 2:]
    return (s == 5).astype(np.uint8)


def dilate_square5(mask: np.ndarray) -> np.ndarray:
    """utils.py:182-211 with the 5x5 square kernel of utils.py:229 (the kernel_size argument
    is ignored by the reference): set iff any pixel of the 5x5 window is set."""
    h, w = mask.shape
    m = np.pad(mask.astype(np.int32), 2)
    s = np.zeros((h, w), np.int32)
    for dy in range(5):
        for dx in range(5):
            s += m[dy:dy + h, dx:dx + w]
    return (s > 0).astype(np.uint8)


def remove_scattered_points(mask: np.ndarray) -> np.ndarray:
    """utils.py:214-237."""
    return dilate_square5(erode_cross3(mask))


def token_selector(t1, t2, threshold, h_tok, w_tok, erosion_dilation=True):
    """utils.py:282-354 ('cosine' branch, the only one any caller uses).
    Returns (edited_ids i64[1,K], unedited_ids i64[1,L-K], raw_mask u8[L], final_mask u8[L])."""
    assert t1.shape[0] == 1, "reference is batch-1 only (quirk A-5)"
    sim = cosine_rows(t1, t2)                       # [1, L]
    raw = (sim <= threshold).squeeze(0).numpy().astype(np.uint8)   # utils.py:333
    final = raw
    if erosion_dilation:                            # utils.py:335-343
        final = remove_scattered_points(raw.reshape(h_tok, w_tok)).reshape(-1)
    ar = np.arange(raw.shape[0], dtype=np.int64)    # utils.py:346-352, both ascending
    edited = torch.from_numpy(ar[final.astype(bool)]).unsqueeze(0)
    unedited = torch.from_numpy(ar[~final.astype(bool)]).unsqueeze(0)
    return edited, unedited, raw, final


# --------------------------------------------------------------------------------------
# a3  gather / scatter
# --------------------------------------------------------------------------------------
def ids_gather(x: torch.Tensor, ids: torch.Tensor) -> torch.Tensor:
    """utils.py:260-279: out[b,k,:] = x[b, ids[b,k], :]."""
    b = torch.arange(ids.shape[0]).unsqueeze(1).expand(-1, ids.shape[1])
    return x[b, ids]


def ids_scatter(src: torch.Tensor, ids: torch.Tensor, dst: torch.Tensor) -> torch.Tensor:
    """utils.py:240-257: dst[b, ids[b,k], :] = src[b,k,:] in place; returns dst."""
    dst[torch.arange(src.shape[0]).unsqueeze(1), ids] = src
    return dst


# --------------------------------------------------------------------------------------
# a4  manager state machine
# --------------------------------------------------------------------------------------
@dataclass
class RegionState:
    """FluxKontextManager, utils.py:357-465 (per-instance instead of module-global)."""
    inference_step: int = 28
    warmup_step: int = 6
    post_step: int = 2
    threshold: float = 0.93
    cache_threshold: float = 0.04
    erosion_dilation: bool = True
    refresh_step: List[int] = field(default_factory=list)
    # per image
    h_tok: int = 0
    w_tok: int = 0
    txt_length: int = 0
    condition_latent: Optional[torch.Tensor] = None
    latent_ids: Optional[torch.Tensor] = None
    current_step: int = 0
    edited_ids: Optional[torch.Tensor] = None
    unedited_ids: Optional[torch.Tensor] = None
    unedited_latent: Optional[torch.Tensor] = None
    prev_refresh_step: Optional[int] = None
    next_refresh_step: Optional[int] = None
    refresh_step_real_time: List[int] = field(default_factory=list)

    def set_parameters(self, num_inference_steps=28, warmup_step=6, post_step=2, refresh_step="16",
                       threshold=0.93, cache_threshold=0.04, erosion_dilation=True, gamma=None):
        """utils.py:390-402.  `gamma` (N-1 decay factors) is the product's documented extension for
        num_inference_steps != 28; the reference itself asserts 28."""
        self.gamma = None if gamma is None else torch.as_tensor(gamma, dtype=torch.float16)
        assert warmup_step >= 1 and (num_inference_steps == 28 or self.gamma is not None), \
            "Changing the inference step requires fitting a new gamma"
        self.inference_step, self.warmup_step, self.post_step = num_inference_steps, warmup_step, post_step
        self.threshold, self.cache_threshold, self.erosion_dilation = threshold, cache_threshold, erosion_dilation
        self.refresh_step = sorted(int(s) for s in refresh_step.split(","))
        assert min(self.refresh_step) > warmup_step + 1 and max(self.refresh_step) <= num_inference_steps - post_step - 1
        assert not any(abs(a - b) == 1 for a, b in zip(self.refresh_step, self.refresh_step[1:])), \
            "Refresh steps must not be adjacent."
        self.refresh_step.append(num_inference_steps - post_step + 1)      # sentinel

    def refresh(self, image_latents, latent_ids, txt_length, h_tok, w_tok):
        """utils.py:437-465."""
        self.h_tok, self.w_tok, self.txt_length = h_tok, w_tok, txt_length
        self.condition_latent, self.latent_ids = image_latents, latent_ids
        self.current_step = 0
        self.prev_refresh_step = self.next_refresh_step = None
        self.edited_ids = self.unedited_ids = self.unedited_latent = None
        self.refresh_step_real_time = list(self.refresh_step)

    def step(self, latent, latent_ids):
        """utils.py:404-435."""
        self.current_step += 1
        cur = self.current_step
        if cur == self.warmup_step:
            self.unedited_latent = ids_gather(latent, self.unedited_ids)
            latent = ids_gather(latent, self.edited_ids)
            latent_ids = ids_gather(latent_ids.unsqueeze(0), self.edited_ids).squeeze(0)
        elif cur == self.inference_step - self.post_step:
            full = torch.zeros_like(self.condition_latent)
            full = ids_scatter(latent, self.edited_ids, full)
            full = ids_scatter(self.unedited_latent, self.unedited_ids, full)
            latent, latent_ids = full, self.latent_ids
            self.prev_refresh_step = None
        elif self.prev_refresh_step is not None and cur == self.prev_refresh_step:
            full = torch.zeros_like(self.condition_latent)
            full = ids_scatter(latent, self.edited_ids, full)
            full = ids_scatter(self.unedited_latent, self.unedited_ids, full)
            latent, latent_ids = full, self.latent_ids
        elif self.prev_refresh_step is not None and cur == self.prev_refresh_step + 1:
            self.unedited_latent = ids_gather(latent, self.unedited_ids)
            latent = ids_gather(latent, self.edited_ids)
            latent_ids = ids_gather(latent_ids.unsqueeze(0), self.edited_ids).squeeze(0)
            self.prev_refresh_step = self.next_refresh_step
        return latent, latent_ids

    # phase predicates used by the loop and the processors ------------------------------
    def is_full_input_step(self):
        """inplace.py:331: image_latents are concatenated only on these steps."""
        c = self.current_step
        return c <= self.warmup_step - 1 or c > self.inference_step - self.post_step - 1 or c == self.prev_refresh_step

    def kv_phase(self):
        """inplace.py:717-732: 'plain' | 'store' | 'update'."""
        c = self.current_step
        if c < self.warmup_step - 1 or c > self.inference_step - self.post_step - 1:
            return "plain"
        if c == self.warmup_step - 1 or c == self.prev_refresh_step:
            return "store"
        return "update"


# --------------------------------------------------------------------------------------
# [EXT] scheduler base: sigma / timestep table
# --------------------------------------------------------------------------------------
def calculate_shift(image_seq_len, base_seq_len=256, max_seq_len=4096, base_shift=0.5, max_shift=1.15):
    """utils.py:38-49 (verbatim arithmetic, it is four lines of algebra)."""
    m = (max_shift - base_shift) / (max_seq_len - base_seq_len)
    b = base_shift - m * base_seq_len
    return image_seq_len * m + b


def flow_match_schedule(num_steps: int, image_seq_len: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """inplace.py:229-244 + [EXT] FlowMatchEulerDiscreteScheduler.set_timesteps with dynamic
    exponential shifting.  Returns (sigmas fp32 [N+1] incl. trailing 0, timesteps fp32 [N])."""
    sig = np.linspace(1.0, 1 / num_steps, num_steps).astype(np.float32)
    mu = calculate_shift(image_seq_len)
    sig = math.exp(mu) / (math.exp(mu) + (1 / sig - 1) ** 1.0)
    sig = torch.from_numpy(np.asarray(sig, dtype=np.float32))
    return torch.cat([sig, torch.zeros(1)]), sig * 1000


# --------------------------------------------------------------------------------------
# a5  scheduler step (split Euler + ARP trigger)
# --------------------------------------------------------------------------------------
def scheduler_step(st: RegionState, sigmas: torch.Tensor, step_index: int,
                   model_output: torch.Tensor, sample: torch.Tensor) -> torch.Tensor:
    """inplace.py:581-691 (non-stochastic, no per-token timesteps).
    Mixed precision exactly as the reference: sample -> fp32; `dt * model_output` is a 0-dim fp32
    tensor times a (possibly bf16) tensor, so dt is rounded to model_output.dtype and the product
    is rounded to that dtype before the fp32 add (quirk A-2)."""
    sample = sample.to(torch.float32)
    sigma, sigma_next = sigmas[step_index], sigmas[step_index + 1]
    cur = st.current_step
    dt_direct = dt_final = None
    if cur == st.warmup_step - 1:                                                  # :630-634
        st.prev_refresh_step = st.refresh_step_real_time.pop(0) - 1
        dt_final = sigmas[-1] - sigma
        dt_direct = sigmas[st.prev_refresh_step] - sigma
    elif st.prev_refresh_step is not None and cur == st.prev_refresh_step and st.refresh_step_real_time:  # :636-639
        st.next_refresh_step = st.refresh_step_real_time.pop(0) - 1
        dt_direct = sigmas[st.next_refresh_step] - sigma
    dt = sigma_next - sigma
    split = False
    if cur == st.warmup_step - 1:                                                  # :648-663
        x0 = sample + dt_final * model_output
        st.edited_ids, st.unedited_ids, _, _ = token_selector(
            x0, st.condition_latent, st.threshold, st.h_tok, st.w_tok, st.erosion_dilation)
        split = True
    elif st.prev_refresh_step is not None and cur == st.prev_refresh_step:         # :665-677
        split = True
    if split:
        sel = ids_gather(sample, st.edited_ids) + dt * ids_gather(model_output, st.edited_ids)
        uns = ids_gather(sample, st.unedited_ids) + dt_direct * ids_gather(model_output, st.unedited_ids)
        prev = torch.zeros_like(sample)
        prev = ids_scatter(sel, st.edited_ids, prev)
        prev = ids_scatter(uns, st.unedited_ids, prev)
    else:
        prev = sample + dt * model_output                                          # :680
    return prev.to(model_output.dtype)                                             # :686


# --------------------------------------------------------------------------------------
# a6  Adaptive Velocity Decay decision
# --------------------------------------------------------------------------------------
@dataclass
class AvdState:
    accumulate: object = 1
    should_cache: bool = False


def avd_decide(st: RegionState, avd: AvdState, i: int, timesteps: torch.Tensor, gamma: torch.Tensor):
    """inplace.py:295-313.  Returns (should_cache, ratio or None).  Arithmetic dtype follows the
    reference: gamma fp16 0-dim * fp32 0-dim -> fp32 (quirk A-7); `accumulate` is carried as a
    tensor once a ratio < 1 has been multiplied in."""
    cur = st.current_step
    ratio = None
    if cur <= st.warmup_step or cur > st.inference_step - st.post_step - 1 or cur == st.prev_refresh_step:
        avd.should_cache, avd.accumulate = False, 1
    else:
        ratio = gamma[i - 1] * (1 + (timesteps[i] - timesteps[i - 1]) / 1000)
        if ratio >= 1:
            avd.should_cache, avd.accumulate = False, 1
        else:
            avd.accumulate = avd.accumulate * ratio
            error = 1 - avd.accumulate
            if error > st.cache_threshold:
                avd.should_cache, avd.accumulate = False, 1
            else:
                avd.should_cache = True
    return avd.should_cache, ratio


def derive_schedule(seq_len: int, family: str = "flux", warmup=6, post=2, refresh="16",
                    cache_threshold=0.04, n=28, gamma=None) -> List[str]:
    """Data-independent step plan (quirk A-7): 'F' full, 'S' full+store K/V, 'R' region forward,
    'C' cache-served.  Simulates avd_decide + the refresh bookkeeping of scheduler_step/step."""
    st = RegionState()
    st.set_parameters(n, warmup, post, refresh, 0.0, cache_threshold, True, gamma=gamma)
    st.refresh(None, None, 0, 0, 0)
    _, ts = flow_match_schedule(n, seq_len)
    gamma = torch.tensor(GAMMA[family], dtype=torch.float16) if gamma is None else st.gamma
    avd, plan = AvdState(), []
    for i in range(n):
        hit, _ = avd_decide(st, avd, i, ts, gamma)
        if hit:
            plan.append("C")
        elif st.is_full_input_step():
            plan.append("S" if st.kv_phase() == "store" else "F")
        else:
            plan.append("R")
        # refresh bookkeeping of scheduler_step / Manager.step without tensors
        cur = st.current_step
        if cur == st.warmup_step - 1:
            st.prev_refresh_step = st.refresh_step_real_time.pop(0) - 1
        elif st.prev_refresh_step is not None and cur == st.prev_refresh_step and st.refresh_step_real_time:
            st.next_refresh_step = st.refresh_step_real_time.pop(0) - 1
        st.current_step += 1
        c = st.current_step
        if c == st.inference_step - st.post_step:
            st.prev_refresh_step = None
        elif st.prev_refresh_step is not None and c == st.prev_refresh_step + 1 and c != st.warmup_step:
            st.prev_refresh_step = st.next_refresh_step
    return plan


# --------------------------------------------------------------------------------------
# a8  index-scatter linear
# --------------------------------------------------------------------------------------
def partially_linear(x, weight, bias, index, out, fp16_roundtrip=True):
    """fused_kernels.py:9-101: out[b, index[m], :] = x[b,m,:] @ W^T + bias (fp32 accumulate).
    The Triton kernel stores `accumulator.to(tl.float16)` (:80) into the cache dtype; pass
    fp16_roundtrip=False for the single-rounding variant the HIP build uses (quirk A-3)."""
    y = F.linear(x.float(), weight.float(), None if bias is None else bias.float())
    out[:, index] = (y.to(torch.float16) if fp16_roundtrip else y).to(out.dtype)
    return out


# --------------------------------------------------------------------------------------
# [EXT] MMDiT arithmetic (upstream diffusers semantics; parity unpinned by the reference)
# --------------------------------------------------------------------------------------
def rms_norm(x, weight, eps=1e-6):
    """[EXT] diffusers RMSNorm: fp32 variance, x*rsqrt in fp32, round to weight dtype, * weight."""
    var = x.to(torch.float32).pow(2).mean(-1, keepdim=True)
    y = x * torch.rsqrt(var + eps)
    if weight.dtype in (torch.float16, torch.bfloat16):
        y = y.to(weight.dtype)
    return y * weight


def layer_norm(x, eps=1e-6):
    return F.layer_norm(x, (x.shape[-1],), None, None, eps)


def apply_rope(x, cos, sin):
    """[EXT] apply_rotary_emb(use_real, unbind_dim=-1). x [B,H,S,D]; cos/sin [S,D] fp32."""
    xr, xi = x.reshape(*x.shape[:-1], -1, 2).unbind(-1)
    rot = torch.stack([-xi, xr], dim=-1).flatten(3)
    return (x.float() * cos[None, None] + rot.float() * sin[None, None]).to(x.dtype)


def flux_pos_embed(ids: torch.Tensor, axes_dim=(16, 56, 56), theta=10000.0):
    """[EXT] FluxPosEmbed: ids [S,3] -> (cos, sin) fp32 [S, sum(axes_dim)], float64 angles."""
    cos_out, sin_out = [], []
    pos = ids.float()
    for i, dim in enumerate(axes_dim):
        freqs = 1.0 / (theta ** (torch.arange(0, dim, 2, dtype=torch.float64)[: dim // 2] / dim))
        ang = torch.outer(pos[:, i].to(torch.float64), freqs)
        cos_out.append(ang.cos().repeat_interleave(2, dim=1).float())
        sin_out.append(ang.sin().repeat_interleave(2, dim=1).float())
    return torch.cat(cos_out, -1), torch.cat(sin_out, -1)


def timestep_embedding(t: torch.Tensor, dim=256, max_period=10000, scale=1.0):
    """[EXT] get_timestep_embedding(flip_sin_to_cos=True, downscale_freq_shift=0); `scale` multiplies the angles
    AFTER t * freqs (diffusers `emb = scale * emb`: Qwen's Timesteps(scale=1000) on timestep / 1000)."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / half)
    a = t[:, None].float() * freqs[None]
    if scale != 1.0:
        a = scale * a
    return torch.cat([torch.cos(a), torch.sin(a)], dim=-1)


def _lin(w: Dict[str, torch.Tensor], name: str, x):
    return F.linear(x, w[name + ".weight"], w.get(name + ".bias"))


def _mlp_embed(w, name, x):
    return _lin(w, name + ".linear_2", F.silu(_lin(w, name + ".linear_1", x)))


def time_text_embed(w, timestep, guidance, pooled, scale=1.0):
    """[EXT] CombinedTimestepGuidanceTextProjEmbeddings."""
    p = "time_text_embed."
    if pooled is None:       # Qwen-Image: conditioning = timestep embedding only [EXT QwenTimestepProjEmbeddings]
        dt = w[p + "timestep_embedder.linear_1.weight"].dtype
        return _mlp_embed(w, p + "timestep_embedder", timestep_embedding(timestep, scale=scale).to(dt))
    t = _mlp_embed(w, p + "timestep_embedder", timestep_embedding(timestep).to(pooled.dtype))
    if guidance is None:     # Step1X-Edit: temb = time_embed(t) + vec_embed(y)  (Step1XEdit/inplace.py:519-520)
        return t + _mlp_embed(w, p + "text_embedder", pooled)
    g = _mlp_embed(w, p + "guidance_embedder", timestep_embedding(guidance).to(pooled.dtype))
    return t + g + _mlp_embed(w, p + "text_embedder", pooled)


@dataclass
class KVCache:
    """RegoionEFluxAttnProcessor2_0.{k_cache,v_cache}, inplace.py:700-702; raw projections."""
    k: Optional[torch.Tensor] = None
    v: Optional[torch.Tensor] = None


def attn_processor(w, prefix, heads, st: RegionState, cache: KVCache, single: bool,
                   hidden, encoder_hidden, rope_q, rope_k, fp16_roundtrip=True):
    """inplace.py:704-824 with flash_attn=None (SDPA branch, quirk A-1).
    rope_q = table for the current (possibly compacted) ids, rope_k = table for the FULL ids
    (MANAGER.image_rotary_emb, inplace.py:495-500)."""
    B = hidden.shape[0]
    a = prefix + ".attn."
    q = _lin(w, a + "to_q", hidden)
    phase = st.kv_phase()
    if phase in ("plain", "store"):                                       # :717-725
        k, v = _lin(w, a + "to_k", hidden), _lin(w, a + "to_v", hidden)
        if phase == "store":
            cache.k, cache.v = k, v
    else:                                                                 # :727-750
        e = st.edited_ids.squeeze(0)
        sel = torch.cat((torch.arange(st.txt_length), e + st.txt_length)) if single else e
        partially_linear(hidden, w[a + "to_k.weight"], w.get(a + "to_k.bias"), sel, cache.k, fp16_roundtrip)
        partially_linear(hidden, w[a + "to_v.weight"], w.get(a + "to_v.bias"), sel, cache.v, fp16_roundtrip)
        k, v = cache.k, cache.v
    hd = k.shape[-1] // heads
    q = q.view(B, -1, heads, hd).transpose(1, 2)
    k = k.view(B, -1, heads, hd).transpose(1, 2)
    v = v.view(B, -1, heads, hd).transpose(1, 2)
    q = rms_norm(q, w[a + "norm_q.weight"])
    k = rms_norm(k, w[a + "norm_k.weight"])
    if encoder_hidden is not None:                                        # :766-790
        eq = _lin(w, a + "add_q_proj", encoder_hidden).view(B, -1, heads, hd).transpose(1, 2)
        ek = _lin(w, a + "add_k_proj", encoder_hidden).view(B, -1, heads, hd).transpose(1, 2)
        ev = _lin(w, a + "add_v_proj", encoder_hidden).view(B, -1, heads, hd).transpose(1, 2)
        eq = rms_norm(eq, w[a + "norm_added_q.weight"])
        ek = rms_norm(ek, w[a + "norm_added_k.weight"])
        q, k, v = torch.cat([eq, q], 2), torch.cat([ek, k], 2), torch.cat([ev, v], 2)
    q = apply_rope(q, *rope_q)                                            # :792-794
    k = apply_rope(k, *rope_k)
    o = F.scaled_dot_product_attention(q, k, v, dropout_p=0.0, is_causal=False)
    o = o.transpose(1, 2).reshape(B, -1, heads * hd).to(q.dtype)
    if encoder_hidden is not None:                                        # :809-822
        T = encoder_hidden.shape[1]
        return _lin(w, a + "to_out.0", o[:, T:]), _lin(w, a + "to_add_out", o[:, :T])
    return o


def _ada6(w, name, x, temb):
    emb = _lin(w, name + ".linear", F.silu(temb))
    sh, sc, g, sh2, sc2, g2 = emb.chunk(6, dim=1)
    return layer_norm(x) * (1 + sc[:, None]) + sh[:, None], g, sh2, sc2, g2


def _ff(w, name, x):
    return _lin(w, name + ".net.2", F.gelu(_lin(w, name + ".net.0.proj", x), approximate="tanh"))


def double_block(w, p, heads, st, cache, h, c, temb, rope_q, rope_k, fp16_roundtrip=True):
    """[EXT] FluxTransformerBlock; call site inplace.py:518-524."""
    nh, g_msa, sh_mlp, sc_mlp, g_mlp = _ada6(w, p + ".norm1", h, temb)
    nc, cg_msa, csh_mlp, csc_mlp, cg_mlp = _ada6(w, p + ".norm1_context", c, temb)
    ao, co = attn_processor(w, p, heads, st, cache, False, nh, nc, rope_q, rope_k, fp16_roundtrip)
    h = h + g_msa.unsqueeze(1) * ao
    nh = layer_norm(h) * (1 + sc_mlp[:, None]) + sh_mlp[:, None]
    h = h + g_mlp.unsqueeze(1) * _ff(w, p + ".ff", nh)
    c = c + cg_msa.unsqueeze(1) * co
    nc = layer_norm(c) * (1 + csc_mlp[:, None]) + csh_mlp[:, None]
    c = c + cg_mlp.unsqueeze(1) * _ff(w, p + ".ff_context", nc)
    return c, h


def single_block(w, p, heads, st, cache, h, c, temb, rope_q, rope_k, fp16_roundtrip=True):
    """[EXT] FluxSingleTransformerBlock; call site inplace.py:549-555."""
    T = c.shape[1]
    x = torch.cat([c, h], dim=1)
    emb = _lin(w, p + ".norm.linear", F.silu(temb))
    sh, sc, gate = emb.chunk(3, dim=1)
    nx = layer_norm(x) * (1 + sc[:, None]) + sh[:, None]
    mlp = F.gelu(_lin(w, p + ".proj_mlp", nx), approximate="tanh")
    ao = attn_processor(w, p, heads, st, cache, True, nx, None, rope_q, rope_k, fp16_roundtrip)
    x = x + gate.unsqueeze(1) * _lin(w, p + ".proj_out", torch.cat([ao, mlp], dim=2))
    return x[:, :T], x[:, T:]


@dataclass
class FluxCfg:
    in_channels: int = 64
    n_double: int = 19
    n_single: int = 38
    heads: int = 24
    head_dim: int = 128
    joint_dim: int = 4096
    pooled_dim: int = 768
    axes_dim: Tuple[int, ...] = (16, 56, 56)
    mlp_ratio: int = 4

    @property
    def d(self):
        return self.heads * self.head_dim


def qwen_rope(img_shapes, txt_len, axes_dim=(16, 56, 56), theta=10000.0):
    """[EXT] QwenEmbedRope(scale_rope=True): (cos, sin) fp32 [txt_len + sum(f*h*w), 128], text rows first
    (text positions start after the image extent; image axes are centred)."""
    def params(index, dim):
        return torch.outer(index.float(), 1.0 / torch.pow(theta, torch.arange(0, dim, 2).float().div(dim)))
    pos_i, neg_i = torch.arange(4096), torch.arange(4096).flip(0) * -1 - 1
    pos, neg = [params(pos_i, d) for d in axes_dim], [params(neg_i, d) for d in axes_dim]
    ang, max_vid = [], 0
    for idx, (fr, h, w) in enumerate(img_shapes):
        f = pos[0][idx: idx + fr].view(fr, 1, 1, -1).expand(fr, h, w, -1)
        hh = torch.cat([neg[1][-(h - h // 2):], pos[1][: h // 2]], 0).view(1, h, 1, -1).expand(fr, h, w, -1)
        ww = torch.cat([neg[2][-(w - w // 2):], pos[2][: w // 2]], 0).view(1, 1, w, -1).expand(fr, h, w, -1)
        ang.append(torch.cat([f, hh, ww], -1).reshape(fr * h * w, -1))
        max_vid = max(max_vid, h // 2, w // 2)
    a = torch.cat([torch.cat([p[max_vid: max_vid + txt_len] for p in pos], 1)] + ang, 0)
    z = torch.polar(torch.ones_like(a), a)                 # [EXT] rope_params: torch.polar(ones, freqs), complex64
    return z.real.repeat_interleave(2, dim=1), z.imag.repeat_interleave(2, dim=1)


def transformer_forward(w, cfg: FluxCfg, st: RegionState, caches: List[KVCache], hidden, enc, pooled,
                        timestep, img_ids, txt_ids, guidance, fp16_roundtrip=True, rope_full=None):
    """inplace.py:413-576.  `timestep` arrives already divided by 1000 (inplace.py:336).
    rope_full: precomputed (cos, sin) for [text ; FULL latent ids] (Qwen: 1-D ids index its image rows,
    QwenImageEdit/inplace.py:531); the query table is its rows at [text ; current ids]."""
    if "txt_norm.weight" in w:                       # [EXT] QwenImageTransformer2DModel.txt_norm
        enc = rms_norm(enc, w["txt_norm.weight"])
    h = _lin(w, "x_embedder", hidden)
    if "txt_norm.weight" in w:
        # Qwen: the forward keeps timestep / 1000 in the model dtype (QwenImageEdit/inplace.py:518) and the [EXT] embedder
        # scales the ANGLES by 1000 in fp32 - unlike FLUX / Step1X, whose forwards multiply the bf16 timestep by 1000
        temb = time_text_embed(w, timestep.to(h.dtype), None, None, scale=1000.0)
    else:
        ts = timestep.to(h.dtype) * 1000
        g = guidance.to(h.dtype) * 1000 if guidance is not None else None
        temb = time_text_embed(w, ts, g, pooled)
    c = _lin(w, "context_embedder", enc)
    if rope_full is not None:
        T = enc.shape[1]
        sel = torch.cat((torch.arange(T), T + img_ids.long()))
        rope_q, rope_k = (rope_full[0][sel], rope_full[1][sel]), rope_full
    else:
        rope_q = flux_pos_embed(torch.cat((txt_ids, img_ids), 0), cfg.axes_dim)            # :495-496
        rope_k = flux_pos_embed(torch.cat((txt_ids, st.latent_ids), 0), cfg.axes_dim)      # :499
    li = 0
    for i in range(cfg.n_double):
        c, h = double_block(w, f"transformer_blocks.{i}", cfg.heads, st, caches[li], h, c, temb, rope_q, rope_k,
                            fp16_roundtrip)
        li += 1
    for i in range(cfg.n_single):
        c, h = single_block(w, f"single_transformer_blocks.{i}", cfg.heads, st, caches[li], h, c, temb, rope_q,
                            rope_k, fp16_roundtrip)
        li += 1
    emb = _lin(w, "norm_out.linear", F.silu(temb).to(h.dtype))
    scale, shift = emb.chunk(2, dim=1)
    h = layer_norm(h) * (1 + scale)[:, None, :] + shift[:, None, :]
    return _lin(w, "proj_out", h)


# --------------------------------------------------------------------------------------
# a11  denoise loop
# --------------------------------------------------------------------------------------
def process_diff_norm(diff_norm, k):
    """[EXT] Step1XEditPipeline.process_diff_norm (call site Step1XEdit/inplace.py:407)."""
    return torch.where(diff_norm > 1.0, torch.pow(diff_norm, k),
                       torch.where(diff_norm < 1.0, torch.ones_like(diff_norm), diff_norm))


def cfg_combine(family, pos, neg, scale, t=None, truncate=0.93, power=0.4):
    """flux: inplace.py:364 | step1x: Step1XEdit/inplace.py:401-410 | qwen: QwenImageEdit/inplace.py:401-405."""
    if family in ("step1x", "step1x_v1p2"):
        if t.item() > truncate:
            diff = pos - neg
            diff_norm = torch.norm(diff, dim=(2), keepdim=True)
            return neg + scale * (pos - neg) / process_diff_norm(diff_norm, k=power)
        return neg + scale * (pos - neg)
    comb = neg + scale * (pos - neg)
    if family == "qwen":
        return comb * (torch.norm(pos, dim=-1, keepdim=True) / torch.norm(comb, dim=-1, keepdim=True))
    return comb


def denoise(model_fn, st: RegionState, latents, image_latents, latent_ids, txt_length, h_tok, w_tok,
            family="flux", regione=True, trace: Optional[dict] = None, neg_model_fn=None, true_cfg_scale: float = 1.0):
    """inplace.py:229-244,287-392 (true_cfg_scale = 1: FLUX-Kontext's normal, guidance-distilled use).

    model_fn(latent_model_input, t (0-dim fp32 timestep), img_ids) -> velocity for all input rows.
    neg_model_fn + true_cfg_scale > 1: the sequential second (unconditional) forward of inplace.py:349-364.
    In the reference both forwards go through the SAME processors, i.e. one K/V cache is shared by the
    two branches (quirk A-4); give the two closures the same `caches` list to reproduce that, or
    separate lists for the per-branch caches Qwen / Step1X-v1p2 use.
    latent_ids: FULL id table [L + L_c, 3].  With regione=False this is the vanilla loop
    (every step full-token, plain Euler), used for the speed-up / PSNR-vs-vanilla figures."""
    n = st.inference_step
    L = latents.shape[1]
    sigmas, timesteps = flow_match_schedule(n, L)
    gamma = torch.tensor(GAMMA[family], dtype=torch.float16) if getattr(st, "gamma", None) is None else st.gamma
    st.refresh(image_latents, latent_ids, txt_length, h_tok, w_tok)
    if not regione:
        for i in range(n):
            v = model_fn(torch.cat([latents, image_latents], 1), timesteps[i], latent_ids)[:, :L]
            latents = (latents.float() + (sigmas[i + 1] - sigmas[i]) * v).to(v.dtype)
            if trace is not None:
                trace.setdefault("latents", []).append(latents.clone())
        return latents
    avd, cache, ids = AvdState(), None, latent_ids
    for i in range(n):
        assert i == st.current_step                                                   # :293
        hit, ratio = avd_decide(st, avd, i, timesteps, gamma)
        if hit:                                                                       # :315-318
            if cache.shape[1] != latents.shape[1]:
                cache = ids_gather(cache, st.edited_ids)
            noise_pred = cache * ratio
        else:
            x = latents
            if st.is_full_input_step():                                               # :331-332
                x = torch.cat([latents, image_latents], dim=1)
            noise_pred = model_fn(x, timesteps[i], ids)[:, :latents.size(1)]          # :336-347
            if neg_model_fn is not None and true_cfg_scale > 1:                       # true CFG, :349-364
                neg = neg_model_fn(x, timesteps[i], ids)[:, :latents.size(1)]
                if trace is not None:                                                 # the two branch velocities before the combine
                    trace.setdefault("branches", {})[i] = (noise_pred.clone(), neg.clone())
                noise_pred = cfg_combine(family, noise_pred, neg, true_cfg_scale, timesteps[i])
            cache = noise_pred                                                        # :365
        if trace is not None:
            trace.setdefault("kind", []).append("C" if hit else ("F" if st.is_full_input_step() else "R"))
            trace.setdefault("noise_pred", []).append(noise_pred.clone())
        latents = scheduler_step(st, sigmas, i, noise_pred, latents)                  # :369
        latents, ids = st.step(latents, ids)                                          # :392
        if trace is not None:
            trace.setdefault("latents", []).append(latents.clone())
            trace.setdefault("prev_refresh", []).append(st.prev_refresh_step)
    return latents


def psnr(a: torch.Tensor, b: torch.Tensor) -> float:
    """PSNR in dB over latents with peak = max|b| (replaces evaluation/metric_all_task.py:85-100,
    which works on uint8 images; on latents the data range is taken from the reference tensor)."""
    a, b = a.double(), b.double()
    mse = torch.mean((a - b) ** 2).item()
    if mse == 0:
        return float("inf")
    peak = b.abs().max().item()
    return 10.0 * math.log10(peak * peak / mse)
