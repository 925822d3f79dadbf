/*
 * regione_hip.h - C ABI of libregione_hip.so: RegionE's region-aware denoising hot path as
 * hand-written HIP kernels for gfx950 (MI355X / CDNA4).
 *
 * The reference (Peyton-Chen/RegionE) is pure Python + one Triton kernel and has NO native
 * surface; each entry point below names the reference Python function (file:line under
 * /root/reference/RegionE/FluxKontext/) it replaces.  INTEGRATION.md shows the ctypes binding a
 * maintainer adds on the reference side.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in _host;
 *   - tensors are dense row-major; `ld*` arguments are row strides in ELEMENTS;
 *   - dtype codes: RGN_F32 = 0, RGN_BF16 = 1;
 *   - `stream` is a hipStream_t (PyTorch: torch.cuda.current_stream().cuda_stream); all work is
 *     enqueued on it, nothing synchronises, nothing allocates, buffers are borrowed;
 *   - return value: 0 on success, otherwise a negative RGN_E_* code or a positive hipError_t;
 *     rgn_last_error() returns a static string describing the last failure on this thread.
 */
#ifndef REGIONE_HIP_H
#define REGIONE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RGN_F32 0
#define RGN_BF16 1

#define RGN_E_BADARG (-1)
#define RGN_E_UNSUPPORTED (-2)

/* Bumped whenever a struct layout or a signature of this header changes.  rgn_version() returns the value the LIBRARY was built
 * with, rgn_abi_struct_bytes() = sizeof(rgn_qkv_epilogue) * 1000 + sizeof(rgn_gemm_problem) as the library sees them: a binding
 * compiled against another header (a stale libregione_torch.so next to a rebuilt libregione_hip.so) compares both at load time
 * and refuses to run instead of misreading structs passed by pointer. */
#define RGN_ABI_VERSION 107
int rgn_version(void);
size_t rgn_abi_struct_bytes(void);
const char* rgn_last_error(void);
/* Launch-plan override: the library's one test / measurement hook (no reference counterpart).  Every knob is -1 (= the launch cost
 * models decide) in a shipped run.  `key` in {gemm_pieces (1 = one plain launch, n >= 2 = n K pieces of a round's remainder),
 * gemm_geometry (128 | 256), gemm_asm (0 = compiler-scheduled kernels only), gemm_quarter (0 never | 1 always), attn_waves (4 | 8),
 * attn_split (0 = never cut KV), attn_streamk (0 = equal pieces only | 1 = wherever possible), attn_asm (0 = compiler-scheduled
 * loop)}; value -1 restores the default; key NULL resets every knob.  Process-wide, not synchronised with launches in flight on other
 * threads.  The same knobs can be preset once per process with RGN_PLAN_OVERRIDE="key=value,key=value" - the only environment
 * variable libregione_hip.so reads, at the first launch.  Results never depend on a knob beyond fp32 summation order. */
int rgn_plan_override(const char* key, int value);
/* The current value of one knob (-1 = not forced), so that a scoped override can restore what it found instead of resetting
 * (nested scopes, a process preset through RGN_PLAN_OVERRIDE). */
int rgn_plan_override_get(const char* key, int* value);
/* The launch plan the GEMM planner chose for the last rgn_gemm_* call on this thread (introspection for tests and traces; the
 * reference has no counterpart): bits 0-7 = K pieces of the remainder (1 = none), bit 8 = quarter-tile remainder, bit 10 =
 * 256 x 256 tile geometry. */
int rgn_gemm_last_plan(void);
/* The plan the planner WOULD choose for a group of `nprob` (<= 4) problems with M = Ms[i] rows, N output channels, depth K, `distinct_w`
 * different weight matrices (problems of one stream share W), bf16 (w8 = 0) or fp8 (w8 = 1) weights and a workspace of
 * `workspace_bytes` (0 = none): same bits as rgn_gemm_last_plan.  Pure host arithmetic on the launch cost model - no launch, no GPU
 * (CPU tests pin the planner's decisions for the region-step shapes with it); negative RGN_E_* on a bad argument. */
int rgn_gemm_plan_query(const int* Ms, int nprob, int N, int K, int distinct_w, int w8, size_t workspace_bytes);

/* ------------------------------------------------------------------------------------------
 * a1/a2  Adaptive Region Partition.  Replaces token_selector (utils.py:282-354) + morphology
 * (utils.py:124-237) + the one-step estimate of the scheduler (inplace.py:650).
 *
 *   est   = model_output ? f32(sample) + round_mo(round_mo(dt_final) * model_output) : f32(sample)
 *   sim   = sum_d normalize(est)[d] * normalize(cond)[d]   (each normalised in its own dtype)
 *   raw   = sim <= threshold
 *   mask  = erosion_dilation ? dilate5x5(erode3x3cross(raw)) : raw     on the [h_tok,w_tok] grid
 *   edited_ids / unedited_ids = ascending token ids with mask 1 / 0;  *count = number edited.
 *
 * sample [L,D] (f32 or bf16, upcast like inplace.py:610), model_output [L,D] or NULL,
 * cond [L,D]; D must be 64 (packed 2x2x16 latent channels).  sim_out may be NULL.
 * Two launches: rows -> similarity -> raw mask (one wave per token), then one workgroup does
 * morphology in LDS and the ballot/popcount prefix-sum compaction.
 */
int rgn_arp_partition(const void* sample, int sample_dtype, const void* model_output, int mo_dtype,
                      const void* cond, int cond_dtype, float dt_final, float threshold,
                      int L, int D, int h_tok, int w_tok, int erosion_dilation,
                      int64_t* edited_ids, int64_t* unedited_ids, uint8_t* raw_mask, uint8_t* mask,
                      float* sim_out, int32_t* count, void* stream);

/* a2 alone: remove_scattered_points (utils.py:214-237) on a u8 [h,w] mask + compaction. */
int rgn_morph_compact(const uint8_t* raw_mask, int h_tok, int w_tok, int erosion_dilation,
                      int64_t* edited_ids, int64_t* unedited_ids, uint8_t* mask, int32_t* count,
                      void* stream);

/* ------------------------------------------------------------------------------------------
 * a3  ids_gather (utils.py:260-279) / ids_scatter (utils.py:240-257) on rows of `row_bytes`
 * bytes (multiple of 4).  dst[k] = src[ids[k]]   /   dst[ids[k]] = src[k].   ids are int64.
 */
int rgn_gather_rows(const void* src, const int64_t* ids, void* dst, int K, int row_bytes, void* stream);
int rgn_scatter_rows(const void* src, const int64_t* ids, void* dst, int K, int row_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * a5  RegionEFlowMatchEulerDiscreteScheduler.step (inplace.py:581-691), arithmetic part.
 *   out[l,:] = cast_v( f32(sample[l,:]) + round_v(round_v(dt_l) * v[l,:]) )
 *   dt_l = mask ? (mask[l] ? dt : dt_direct) : dt        (split Euler on partition/refresh steps)
 * sample f32|bf16, v f32|bf16, out has v's dtype (inplace.py:686).  Gather-free: one pass over
 * the L rows instead of the reference's 4 gathers + 2 scatters + zeros + 2 axpy.
 */
int rgn_euler_step(const void* sample, int sample_dtype, const void* v, int v_dtype, void* out,
                   const uint8_t* mask, float dt, float dt_direct, int L, int D, void* stream);

/* ------------------------------------------------------------------------------------------
 * a6  Adaptive Velocity Decay cache hit (inplace.py:315-318):
 *   out[k,:] = round_c( r * cache[ids ? ids[k] : k, :] ),   r = round_ratio ? round_c(ratio) : ratio
 * The optional ids fuse the first-hit gather (inplace.py:316-317).
 * `cache * ratio` multiplies a bf16 tensor by a 0-dim fp32 tensor.  Torch's CPU kernel keeps the
 * scalar in fp32 (round_ratio = 0, what the reference-generated fixtures contain); its CUDA kernel
 * casts a *device* 0-dim tensor to bf16 first (round_ratio = 1).  Both are offered.
 */
int rgn_avd_apply(const void* cache, int dtype, const int64_t* ids, float ratio, int round_ratio, void* out,
                  int K, int D, void* stream);

/* ------------------------------------------------------------------------------------------
 * a11  classifier-free-guidance combine of the cond / uncond velocities, rows of 64 channels:
 *   mode 0  FLUX true-CFG (inplace.py:364)                      out = neg + s*(pos-neg)
 *   mode 1  Step1X norm-rescaled (Step1XEdit/inplace.py:401-410) out = neg + s*(pos-neg)/f(||pos-neg||),
 *           f(n) = n > 1 ? n^power : (n < 1 ? 1 : n)
 *   mode 2  Qwen norm-preserving (QwenImageEdit/inplace.py:401-405)
 *           c = neg + s*(pos-neg);  out = c * (||pos|| / ||c||)
 * Every intermediate is rounded to the tensor dtype like the eager op sequence.
 */
int rgn_cfg_combine(const void* pos, const void* neg, void* out, int dtype, float scale, int mode, float power,
                    int K, int D, void* stream);

/* ------------------------------------------------------------------------------------------
 * bf16 MFMA GEMM  C = epilogue(A[M,K] @ W[N,K]^T + bias)  (fp32 accumulate).
 * Replaces torch nn.Linear calls of the block bodies [EXT diffusers] and, with `out_rows`,
 * the Triton index-scatter GEMM _partially_linear (fused_kernels.py:9-101).
 *   epilogue: RGN_EPI_BIAS        C[r] = bf16(acc + bias)
 *             RGN_EPI_GELU        C[r] = bf16(gelu_tanh(bf16(acc + bias)))  for columns >= gelu_from_col
 *             RGN_EPI_GATE_RESID  C[r] = bf16(f32(resid[r]) + f32(bf16(gate[n] * bf16(acc + bias))))
 *   r = out_rows ? out_rows[m] : m   (row scatter; out_rows int64 or NULL)
 * K % 64 == 0; M, N arbitrary (>0).  resid uses ldc and may alias C.
 */
#define RGN_EPI_BIAS 0
#define RGN_EPI_GELU 1
#define RGN_EPI_GATE_RESID 2
int rgn_gemm_bf16(const void* A, int lda, const void* W, int ldw, const void* bias, void* C, int ldc,
                  int M, int N, int K, int epilogue, int gelu_from_col, const void* gate,
                  const void* resid, const int64_t* out_rows, void* workspace, size_t workspace_bytes,
                  void* stream);
/* `workspace` (optional, fp32 scratch, rgn_gemm_workspace_bytes()) enables the round-aware schedule:
 * output tiles that do not fill a whole round of the chip's workgroup slots are cut along K, spread
 * over all CUs and finished by a reduce pass (8 workgroups per tile, epilogue in registers; bit-identical to a reduce by the
 * tile's own workgroup).  NULL = plain single launch.
 * Sizing: rgn_gemm_workspace_bytes() is a shape-independent upper bound (256 MiB = 255 remainder tiles x 4 pieces x 256 KiB of
 * fp32 fragments, plus room for the fp8 weight widening of rgn_gemm_w8*); any smaller buffer is valid too - the planner only
 * considers piece counts whose partials fit in `workspace_bytes` (and skips the split / the widening when nothing fits).
 * A workspace belongs to ONE stream at a time: calls on two streams need two buffers. */
size_t rgn_gemm_workspace_bytes(void);

/* Two independent problems with the same N, K and epilogue in ONE launch (the text and image
 * streams of a double block: different A / W / bias / C / gate / resid).  The small problem's tiles
 * fill the tail of the large one instead of running as an under-occupied launch of their own. */
int rgn_gemm_bf16_pair(const void* A0, int lda0, const void* W0, const void* bias0, void* C0, int ldc0, int M0,
                       const void* gate0, const void* resid0, const void* A1, int lda1, const void* W1,
                       const void* bias1, void* C1, int ldc1, int M1, const void* gate1, const void* resid1,
                       int N, int K, int epilogue, int gelu_from_col, void* workspace, size_t workspace_bytes,
                       void* stream);

/* Fused QKV projection (reference: the Linear projections + `attn.norm_q/k` + `apply_rotary_emb` + K/V cache
 * placement of RegoionEFluxAttnProcessor2_0.__call__, FluxKontext/inplace.py:735-794; the K/V partial update
 * `_partially_linear`, fused_kernels.py:81-101).  Same GEMM as rgn_gemm_bf16, but the epilogue of each 256-column
 * block of C does what rgn_qk_norm_rope_store does as a separate pass:
 *   columns [k_col, k_col + heads*128): per-head RMSNorm + RoPE (k tables, row kv_rows[r]) -> k_slab row kv_rows[r]
 *   columns [v_col, ...):               transposed into vt_slab (kv index permuted as rgn_attention expects)
 *   columns [q_col, ...):               per-head RMSNorm + RoPE (q tables, row r) -> C in place
 *   columns >= gelu_from_col:           GELU-tanh -> C           (the fused MLP half of a single-stream block)
 * K and V columns are NOT written to C.  r = row_base + local row (row_base: where this problem's rows sit in
 * the joint [text | image] sequence that the tables / kv_rows are indexed by).  Results are bit-identical to
 * rgn_gemm_bf16 followed by rgn_qk_norm_rope_store. */
typedef struct rgn_qkv_epilogue {
    const void* wq;            /* [128] bf16 RMSNorm weights of this stream (norm_q / norm_added_q) */
    const void* wk;
    const float* cos_q;        /* [rows][128] fp32 */
    const float* sin_q;
    const float* cos_k;
    const float* sin_k;
    const int64_t* kv_rows;    /* joint row -> K/V cache row, NULL = identity */
    void* k_slab;              /* [kv rows][heads*128] bf16 */
    void* vt_slab;             /* [heads*128][skv_pad] bf16 */
    int row_base, skv_pad, k_col, v_col, q_col, heads;
    float eps;
    int fp16_roundtrip;        /* 1: K / V columns round fp32 -> fp16 -> bf16 like the reference's partial-update kernel
                                  (fused_kernels.py:80); 0: one rounding, like F.linear on store / plain steps */
} rgn_qkv_epilogue;
#define RGN_EPI_QKV 3
#define RGN_EPI_CONV 4            /* rgn_conv_bf16 only */
int rgn_gemm_bf16_qkv(const void* A, int lda, const void* W, int ldw, const void* bias, void* C, int ldc, int M, int N,
                      int K, int gelu_from_col, const rgn_qkv_epilogue* e, void* workspace, size_t workspace_bytes,
                      void* stream);
int rgn_gemm_bf16_qkv_pair(const void* A0, int lda0, const void* W0, const void* bias0, void* C0, int ldc0, int M0,
                           const rgn_qkv_epilogue* e0, const void* A1, int lda1, const void* W1, const void* bias1,
                           void* C1, int ldc1, int M1, const rgn_qkv_epilogue* e1, int N, int K, void* workspace,
                           size_t workspace_bytes, void* stream);

/* Up to FOUR problems with the same N, K, epilogue and weight format in ONE launch: the text and image streams of a double
 * block for BOTH classifier-free-guidance branches (reference: the B = 2 batched CFG forward, Step1XEdit/inplace.py:381-399;
 * Step1XEditV1P2/inplace.py:398,416 and QwenImageEdit/inplace.py:371-405 run the branches in sequence - rows of different
 * branches never interact in a Linear, so one launch computes both).  Each problem keeps its own activations, output,
 * gate / residual (per-branch AdaLN gates) and - with RGN_EPI_QKV - its own Q/K/V epilogue descriptor (per-branch K / V^T
 * cache slabs, rotary tables and cache-row lists); problems may share W (streamed from HBM once for both branches).
 * W is [N, K] contiguous (bf16, or fp8 bytes when `wscale` is set - then for every problem).  M == 0 problems are skipped.
 * Per output element the accumulation order depends only on the tile geometry and the split-K piece count the planner
 * picks for the launch, never on which other problems share it. */
typedef struct rgn_gemm_problem {
    const void* A;             /* [M, K] bf16, row stride lda */
    const void* W;             /* [N, K] */
    const float* wscale;       /* per-output-channel fp32 scale: W is OCP e4m3fn bytes; NULL: W is bf16 */
    const void* bias;          /* [N] bf16 or NULL */
    void* C;                   /* [M, N] bf16, row stride ldc */
    const void* gate;          /* RGN_EPI_GATE_RESID */
    const void* resid;
    const rgn_qkv_epilogue* qkv;   /* RGN_EPI_QKV */
    int lda, ldc, M;
} rgn_gemm_problem;
int rgn_gemm_group(const rgn_gemm_problem* probs, int nprob, int N, int K, int epilogue, int gelu_from_col, void* workspace,
                   size_t workspace_bytes, void* stream);

/* fp8 weights (BASELINE.json configs[4]: "fp8 weights on CDNA4").  The same four GEMM entry points with W stored as OCP
 * e4m3fn bytes ([N, K], ldw in BYTES, a multiple of 16) plus one fp32 scale per output channel (`wscale[N]`, 16-byte
 * aligned): C = epilogue((A @ dequant(W8)^T) * wscale[n] + bias).  Activations, bias, outputs and every epilogue stay
 * bf16 / fp32 exactly as in the bf16 calls; the fp8 -> bf16 conversion is exact (v_cvt_scalef32_pk_bf16_fp8 in registers, after
 * the LDS read), the scale multiplies the fp32 accumulator.  Replaces nothing in the reference (which ships bf16 weights):
 * storage format of the [EXT] Linear weights only; quantisation (per-channel absmax / 448) is done by the caller
 * (regione_amd.harness.flux.FluxTransformer2DModel.quantize_fp8_).
 * Default since round 3: the fp8 tiles are converted in registers inside the hand-scheduled K loop.  A/B switch RGN_W8_WIDEN_MIN_M=<rows>
 * (default 0 = never): with a workspace and at least that many rows in total the call first widens W8 to bf16 (exact) into the TAIL
 * of the workspace (N x K x 2 bytes per distinct weight matrix, >= 64 MiB left for the split remainders) and runs the bf16 K loop on
 * it - same result bit for bit, weights stay fp8 in HBM. */
int rgn_gemm_w8(const void* A, int lda, const void* W8, int ldw, const float* wscale, const void* bias, void* C, int ldc,
                int M, int N, int K, int epilogue, int gelu_from_col, const void* gate, const void* resid,
                const int64_t* out_rows, void* workspace, size_t workspace_bytes, void* stream);
int rgn_gemm_w8_pair(const void* A0, int lda0, const void* W0, const float* wscale0, const void* bias0, void* C0, int ldc0,
                     int M0, const void* gate0, const void* resid0, const void* A1, int lda1, const void* W1,
                     const float* wscale1, const void* bias1, void* C1, int ldc1, int M1, const void* gate1,
                     const void* resid1, int N, int K, int epilogue, int gelu_from_col, void* workspace,
                     size_t workspace_bytes, void* stream);
int rgn_gemm_w8_qkv(const void* A, int lda, const void* W8, int ldw, const float* wscale, const void* bias, void* C, int ldc,
                    int M, int N, int K, int gelu_from_col, const rgn_qkv_epilogue* e, void* workspace,
                    size_t workspace_bytes, void* stream);
int rgn_gemm_w8_qkv_pair(const void* A0, int lda0, const void* W0, const float* wscale0, const void* bias0, void* C0,
                         int ldc0, int M0, const rgn_qkv_epilogue* e0, const void* A1, int lda1, const void* W1,
                         const float* wscale1, const void* bias1, void* C1, int ldc1, int M1,
                         const rgn_qkv_epilogue* e1, int N, int K, void* workspace, size_t workspace_bytes, void* stream);


/* Skinny GEMV for the AdaLN modulation / timestep embedders:
 *   y[b,n] = bf16( sum_k W[n,k] * act(x[b,k]) + bias[n] ),  act = silu (rounded to bf16) if silu_input.
 * B <= 4, K % 8 == 0.  HBM-bound on W. */
int rgn_gemv_bf16(const void* x, int ldx, const void* W, const void* bias, void* y, int ldy, int B, int N,
                  int K, int silu_input, void* stream);

/* out[m,:] = bf16(bf16(x[m,:] * rsqrt(mean(x^2) + eps)) * w)  - RMSNorm over rows of width d
 * (QwenImageTransformer2DModel.txt_norm [EXT], call site QwenImageEdit/inplace.py:514). */
int rgn_rms_norm_rows(const void* x, int ldx, const void* w, void* out, int ldo, int M, int d, float eps,
                      void* stream);

/* y = bf16(silu(x)) elementwise on bf16 (F.silu of the AdaLN conditioning vector). */
int rgn_silu_bf16(const void* x, void* y, size_t n, void* stream);

/* y = bf16(a + b) elementwise on bf16 (fp32 add, one rounding = torch's bf16 add): `timesteps_emb + guidance_emb`, `+ pooled_projections`
 * of CombinedTimestepGuidanceTextProjEmbeddings [EXT] (call site inplace.py:476-480).  y may alias a or b. */
int rgn_add_bf16(const void* a, const void* b, void* y, size_t n, void* stream);

/* out[i] = i (i < T), out[T + k] = T + edited_ids[k]: the cache rows [text ; T + edited ids] a region step rewrites -
 * `selection = torch.cat((arange(txt_len), edited_ids + txt_len))`, inplace.py:732-733.  out: int64 [T + K]. */
int rgn_sel_rows(const int64_t* edited_ids, int K, int T, int64_t* out, void* stream);

/* hipMemsetAsync(ptr, 0, bytes) on `stream`: zero-initialises a K / V^T cache slab (the padding rows up to skv_pad must be finite;
 * the reference's caches have no padding) without a torch fill kernel. */
int rgn_fill_zero(void* ptr, size_t bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * LayerNorm(eps, no affine) * (1 + scale) + shift over rows of width d (AdaLN-Zero modulate).
 * Rows < split_row use (shift0, scale0), the others (shift1, scale1) - the text / image streams
 * of a double block in one launch.  Rounding points follow the eager bf16 op sequence.
 */
int rgn_ln_modulate(const void* x, int ldx, void* out, int ldo, int M, int d, float eps, int split_row,
                    const void* shift0, const void* scale0, const void* shift1, const void* scale1,
                    void* stream);

/* The same with up to FOUR row segments: rows [seg_end[i-1], seg_end[i]) use (shift[i], scale[i]); seg_end[nseg-1] == M.
 * `seg_end_host`, `shift_host`, `scale_host` are HOST arrays (of ints / device pointers) read at call time.  Used by the
 * batched CFG forward: [text_cond ; image_cond ; text_uncond ; image_uncond] rows with per-stream, per-branch AdaLN vectors
 * (reference: the B = 2 forward, Step1XEdit/inplace.py:381-399, where `temb` has one row per branch). */
int rgn_ln_modulate_segs(const void* x, int ldx, void* out, int ldo, int M, int d, float eps, int nseg,
                         const int* seg_end_host, const void* const* shift_host, const void* const* scale_host, void* stream);

/* ------------------------------------------------------------------------------------------
 * Region-Instruction KV-cache write (RegoionEFluxAttnProcessor2_0, inplace.py:717-763,792-794):
 * per-head RMSNorm(q), RMSNorm(k) + RoPE on raw projections and placement into the K / V^T slab.
 *   qkv     [M, ld] bf16 with column blocks k_raw @ k_col, v_raw @ v_col, q_raw @ q_col (H*128 each)
 *   q is normalised + rotated IN PLACE (cos/sin row = rope_q_rows ? rope_q_rows[m] : m of table q);
 *   k row m goes to slab row kv_rows ? kv_rows[m] : m, rotated with the FULL-id table at that row
 *     (MANAGER.image_rotary_emb, inplace.py:499);  v row m goes to column kvpos(row) of V^T.
 *   rows < split_row use (wq0, wk0) RMSNorm weights (text stream: norm_added_q/k), others (wq1, wk1).
 * K slab: [Skv_pad, H*128] bf16.  V^T slab: [H*128, Skv_pad] bf16 with the kv index permuted inside
 * every 16-group (bits 2<->3 swapped) so that the attention kernel's PV operand is one 16-byte read.
 */
int rgn_qk_norm_rope_store(void* qkv, int ld, int k_col, int v_col, int q_col, int M, int H,
                           int split_row, const void* wq0, const void* wk0, const void* wq1, const void* wk1,
                           float eps, const float* cos_q, const float* sin_q, const float* cos_k,
                           const float* sin_k, const int64_t* kv_rows, void* k_slab, void* vt_slab,
                           int skv_pad, void* stream);

/* ------------------------------------------------------------------------------------------
 * Joint attention over the compacted query set against the full K/V cache; replaces
 * flash_attn_func / SDPA (inplace.py:796-806).  softmax(Q K^T / sqrt(128)) V, non-causal,
 * Sq != Skv allowed, head_dim 128, fp32 online softmax, bf16 MFMA.
 *   Q [Sq, H*128] (row stride ldq) ; K slab / V^T slab as written by rgn_qk_norm_rope_store;
 *   O [Sq, H*128] (row stride ldo), may alias Q (each workgroup reads its Q tile before writing;
 *   with a workspace, split items write O only in the final combine pass).
 */
int rgn_attention(const void* Q, int ldq, const void* k_slab, const void* vt_slab, int skv_pad, void* O,
                  int ldo, int Sq, int Skv, int H, float scale, void* workspace, size_t workspace_bytes,
                  void* stream);
/* The same with a caller-provided bound on the scores: the caller GUARANTEES |q . k| * scale <= score_bound for every (query,
 * key) pair of the call - e.g. q and k are RMS-normalised per head (norm_q / norm_k, inplace.py:760-763) and rotated, so
 * |q . k| <= 128 * max|w_q| * max|w_k|.  With score_bound * log2(e) <= 96 the hand-scheduled kernel then computes
 * P = exp2(s * log2 e) without a running row maximum (no per-tile max, no rescale: ~11 % fewer VALU instructions per KV tile); the
 * result is the same softmax up to fp32 / bf16 rounding.  A bound of 0 (or one that is too large) = rgn_attention.  A violated
 * bound is the caller's error (overflow). */
int rgn_attention_bounded(const void* Q, int ldq, const void* k_slab, const void* vt_slab, int skv_pad, void* O, int ldo,
                          int Sq, int Skv, int H, float scale, float score_bound, void* workspace, size_t workspace_bytes,
                          void* stream);
/* Optional fp32 scratch for the round-aware schedule: (head, q-block) items that do not fill a whole
 * round of the chip's workgroup slots are cut along KV (equal pieces or stream-K runs, chosen per launch) and merged by a
 * combine kernel.  NULL disables the split (results are identical up to fp32 summation order).  The returned size (128 MiB)
 * is a shape-independent upper bound; a smaller buffer limits the piece count.  One workspace per stream. */
size_t rgn_attention_workspace_bytes(int Sq, int H);
/* Introspection of the round-aware schedule (tests, traces; no reference counterpart): the plan of the last rgn_attention* call on this
 * thread / the plan the scheduler WOULD choose for (Sq, Skv, H) with a workspace of `workspace_bytes` (0 = none) - pure host arithmetic,
 * no launch, no GPU.  Bits 0-3 = equal KV pieces of the remainder items (1 = not split), bit 4 = stream-K remainder, bit 5 = 8-wave
 * workgroups (256 query rows; clear = 4 waves x 128 rows for tiny query sets). */
int rgn_attention_last_plan(void);
int rgn_attention_plan_query(int Sq, int Skv, int H, size_t workspace_bytes);

/* ---- f4: VAE decode, the host-side step behind the loop ------------------------------------------------------
 * Replaces `self.vae.decode(latents, return_dict=False)[0]` (reference FluxKontext/inplace.py:396-402; Step1XEdit / Step1XEditV1P2
 * the same call) for the [EXT] AutoencoderKL of the public FLUX.1 / Step1X-Edit checkpoints (16 latent channels, block widths
 * 128-256-512-512, 3 ResNet blocks per up level, one mid-block attention head of width 512, GroupNorm(32, eps 1e-6) + SiLU).
 * Activation layout of every entry: a ZERO-BORDERED pixel-major image [Hp * Wp, C] bf16, Hp = H + 2, Wp = W + 2, row = y * Wp + x of
 * the PADDED image, channels contiguous.  The caller allocates >= Wp + 1 rows of readable memory in front of row 0 and behind the
 * last row (guard rows: a 3 x 3 window reads them for border outputs, which are then written as zeros).
 *
 * rgn_conv_bf16: Y = conv(X, Wt) + bias (+ resid), border rows of Y written as ZEROS.  taps = 9: 3 x 3, stride 1, zero padding 1, as an
 * implicit GEMM on the hand-scheduled MFMA loop (K = 9 Cin in (ky, kx, c) order: inside one kernel row the three taps' channels are
 * contiguous in this layout, so the A tile of a K step is the plain GEMM's at another byte offset); Wt = [Cout, 3, 3, Cin] (the
 * checkpoint's [Cout, Cin, 3, 3] permuted once at load), needs ldx == Cin.  taps = 1: 1 x 1 (ResNet shortcut, attention projections),
 * Wt = [Cout, Cin].  Cin % 64 == 0; resid (or NULL) has Y's layout; fp32 accumulation, one bf16 rounding of acc + bias, one of the
 * residual sum.
 * group > 1 (narrow outputs: Cout = 128, the RGB head): one GEMM row = `group` consecutive pixels, its output row = group x ldy columns, and
 * Wt is the caller-built block-Toeplitz matrix [group * ldy, 3 * (group + 2) * Cin] (taps = 9; row p * ldy + c holds W[c, ky, kx] at
 * window pixel p + kx of kernel row ky, zeros elsewhere and in the padding channels) or [group * ldy, group * Cin] (taps = 1, block
 * diagonal); bias = [group * ldy].  The 256-wide tile is then full: MFMA work (group + 2) / 3 of the ideal instead of 256 / Cout.  Rows of Y
 * up to group - 1 past the image are written (zeros): the caller's guard rows.
 * gn_partial != NULL: the epilogue also leaves the GroupNorm(32) statistics of the STORED image in the caller's groupnorm workspace - per tile
 * 32 x (sum, sum of squares), folded in a fixed order - and writes the tile count to *gn_blocks_host (a HOST int); passing that count as
 * rgn_groupnorm_silu's `precomputed_blocks` skips its statistics pass (one read of the image less per ResNet half). */
int rgn_conv_bf16(const void* X, int ldx, const void* Wt, const void* bias, const void* resid, void* Y, int ldy, int Hp, int Wp,
                  int Cin, int Cout, int taps, int group, float* gn_partial, int* gn_blocks_host, void* stream);
/* The encoder's downsampling convolution (VAE encode of the condition image, inside the host's `prepare_latents`, reference call site
 * FluxKontext/inplace.py:210-226): 3 x 3, stride 2, F.pad(x, (0, 1, 0, 1)) + padding 0.  X = padded image [Hp * Wp, Cin] (even H, W);
 * GEMM row m = yo * Wp + xo on the INPUT pitch (the A row is image row 2 m + Wp + 1: the same loop with a row stride of two pixels);
 * out_rows[m] (int64, (Hp - 2) / 2 * Wp entries, device) = the row of Y it is stored to - the caller's table: the padded row of output
 * pixel (yo, xo) for xo < W / 2, a scratch row for the unused columns of the wide grid.  Wt = [Cout, 3, 3, Cin]; no border test. */
int rgn_conv_s2_bf16(const void* X, const void* Wt, const void* bias, void* Y, int ldy, int Hp, int Wp, int Cin, int Cout,
                     const int64_t* out_rows, void* stream);
/* The decoder's `upsamplers.0` (nearest 2 x upsample, then a 3 x 3 convolution) WITHOUT the upsampled image: every output phase (a, b) of
 * pixel (2y + a, 2x + b) is a 2 x 2 convolution of the LOW-resolution image X [Hp * Wp, Cin] with tap-summed weights (16 / 36 of the FLOPs).
 * Wt4 = [4 phases (a * 2 + b)][Cout, 2, 2, Cin] (caller: kernel row 0 / 1 of phase a = 0 holds w[0] / w[1] + w[2], of a = 1 w[0] + w[1] / w[2];
 * columns alike with b); out_rows4 = [4][Hp * Wp] int64 (device): the row of the high-resolution padded image Y each low-resolution padded pixel's
 * phase is stored to (border pixels: a scratch row; their values are zeroed); Y's own border rows are not written (the caller keeps them
 * zero).  One launch of four problems.  gn_partial / gn_blocks_host as rgn_conv_bf16 (the statistics of all of Y). */
int rgn_conv_up2_bf16(const void* X, const void* Wt4, const void* bias, void* Y, int ldy, int Hp, int Wp, int Cin, int Cout,
                      const int64_t* out_rows4, float* gn_partial, int* gn_blocks_host, void* stream);
/* GroupNorm(32 groups) over the valid pixels + optional SiLU: Y = silu((X - mean_g) * rstd_g * gamma + beta), border rows of Y = 0.
 * C in {128, 256, 512}.  Statistics: per-block fp32 partial sums folded in a fixed order + one double-precision pass (no atomics:
 * bit-reproducible).  `workspace`: rgn_groupnorm_workspace_bytes() bytes, 16-byte aligned, one per stream.  precomputed_blocks > 0: X was
 * written by rgn_conv_bf16 with gn_partial = this workspace, which left that many tiles' sums there: no statistics pass. */
size_t rgn_groupnorm_workspace_bytes(void);
size_t rgn_groupnorm_partial_bytes(void);          /* the leading part of the workspace a convolution's gn_partial may fill */
int rgn_groupnorm_silu(const void* X, void* Y, int Hp, int Wp, int C, const void* gamma, const void* beta, float eps, int silu,
                       void* workspace, int precomputed_blocks, void* stream);
/* Nearest-neighbour 2 x upsample of a padded image [Hp * Wp, C] into the padded image [(2 Hp - 2) * (2 Wp - 2), C] (border zero). */
int rgn_upsample2x(const void* X, void* Y, int Hp, int Wp, int C, void* stream);
/* Mid-block attention = three GEMMs (rgn_gemm_bf16) + this pass: S [Hp * Wp, ld] holds q . k for every (query, key) pixel of the
 * padded image; in place P = softmax(scale * S) per row over the VALID key columns (border pixels and the padding columns
 * [Hp * Wp, ld) get probability 0).  ld % 8 == 0, ld <= 24576. */
int rgn_softmax_rows(void* S, int ld, int Hp, int Wp, float scale, void* stream);
/* z [Cz, H, W] bf16 (one NCHW image) -> padded pixel-major [Hp * Wp, Cpad], channels [Cz, Cpad) and the border zero; and back:
 * the first Co channels of a padded image with row stride ld -> [Co, H, W] bf16. */
int rgn_nchw_to_padded(const void* Z, void* Y, int Cz, int H, int W, int Cpad, void* stream);
int rgn_padded_to_nchw(const void* X, int ld, void* O, int Co, int H, int W, void* stream);

/* Device properties the host side needs for roofline reporting (no torch types). */
int rgn_device_info(int* cu_count, int* clock_khz, size_t* hbm_bytes);

#ifdef __cplusplus
}
#endif
#endif /* REGIONE_HIP_H */
