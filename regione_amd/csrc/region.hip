// Region ops of RegionE's hot path: Adaptive Region Partition, row gather/scatter, split Euler
// step, Adaptive-Velocity-Decay cache hit.  HBM-streaming, O(L*64) bytes: latency-bound, so the
// design goal is FEW launches and zero host syncs, with bit-exact rounding vs the reference.
#include "common.h"
#include <stdlib.h>
#include <string.h>

namespace rgn {

thread_local char g_err[256] = "";

// ------------------------------------------------------------------------------------------------
// ARP phase 1: one wave per token, lane = latent channel (D == 64).
//   reference: inplace.py:650 (estimate), utils.py:310-312 (cosine), utils.py:333 (threshold)
// ------------------------------------------------------------------------------------------------
template <typename T> __device__ __forceinline__ float ld_as_f32(const T* p, size_t i);
template <> __device__ __forceinline__ float ld_as_f32<float>(const float* p, size_t i) { return p[i]; }
template <> __device__ __forceinline__ float ld_as_f32<uint16_t>(const uint16_t* p, size_t i) { return bf2f(p[i]); }

// torch-CPU reduction trees over a 64-element row held one element per lane.  The reference runs on torch eager; its CPU
// kernels (the parity oracle) reduce a contiguous last dimension of 64 with FIXED trees, reproduced here bit for bit
// (tests/test_host_logic.py::test_torch_cpu_reduction_trees pins them against torch itself):
//   * torch.sum (SumKernel.cpp vectorized_inner_sum, 8-float vectors, 4 interleaved accumulators):
//       t[k][j] = p[8k+j] + p[8(k+4)+j] (k < 4, j < 8);  a[j] = ((t[0][j] + t[1][j]) + t[2][j]) + t[3][j];
//       sum = (((a[0] + a[1]) + a[2]) + ... + a[7])
//   * fp32 norm (F.normalize -> linalg_vector_norm): acc[j] = x[j]^2, then acc[j] = fma(x[8b+j], x[8b+j], acc[j]) for
//       b = 1..7 (the compiler contracts the accumulate), sum = ((acc[0] + acc[1]) + ... + acc[7]), sqrt
//   * bf16 norm (fp32 accumulate, squares exact): xor butterfly 32, 16, 8, 4, 2, 1 == wave_sum()
__device__ __forceinline__ float row_sum_torch(float p, int lane) {
    const int j = lane & 7;
    const float t = p + __shfl_xor(p, 32, 64);
    const float a = ((__shfl(t, j, 64) + __shfl(t, 8 + j, 64)) + __shfl(t, 16 + j, 64)) + __shfl(t, 24 + j, 64);
    float s = __shfl(a, 0, 64);
#pragma unroll
    for (int l = 1; l < 8; ++l) s = s + __shfl(a, l, 64);
    return s;
}

__device__ __forceinline__ float row_sumsq_torch_f32(float x, int lane) {
    const int j = lane & 7;
    const float x0 = __shfl(x, j, 64);
    float acc = __fmul_rn(x0, x0);
#pragma unroll
    for (int b = 1; b < 8; ++b) {
        const float xb = __shfl(x, 8 * b + j, 64);
        acc = __builtin_fmaf(xb, xb, acc);
    }
    float s = __shfl(acc, 0, 64);
#pragma unroll
    for (int l = 1; l < 8; ++l) s = s + __shfl(acc, l, 64);
    return s;
}

template <typename TS, typename TM, typename TC>
__global__ __launch_bounds__(256) void arp_sim_kernel(const TS* __restrict__ sample, const TM* __restrict__ mo,
                                                      const TC* __restrict__ cond, float dt_final, float thr,
                                                      int L, uint8_t* __restrict__ raw, float* __restrict__ sim_out) {
    const int lane = threadIdx.x & 63;
    const int tok = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tok >= L) return;
    const size_t i = (size_t)tok * 64 + lane;
    float est = ld_as_f32<TS>(sample, i);                     // sample.to(float32), inplace.py:610
    if (mo != nullptr) {
        float p;
        if constexpr (sizeof(TM) == 2) {
            // 0-dim fp32 dt times bf16 tensor: dt is cast to bf16, product rounded to bf16 (quirk A-2)
            p = rbf(__fmul_rn(rbf(dt_final), bf2f(mo[i])));
        } else {
            p = __fmul_rn(dt_final, mo[i]);
        }
        est = __fadd_rn(est, p);
    }
    // F.normalize(est): fp32 norm, clamp_min(1e-12), divide
    float n1 = sqrtf(row_sumsq_torch_f32(est, lane));
    float a = est / fmaxf(n1, 1e-12f);
    float b;
    if constexpr (sizeof(TC) == 2) {
        // bf16 operand normalised in bf16: norm rounded to bf16, quotient rounded to bf16
        float c = bf2f(cond[i]);
        float n2 = rbf(sqrtf(wave_sum(__fmul_rn(c, c))));
        n2 = fmaxf(n2, rbf(1e-12f));
        b = rbf(c / n2);
    } else {
        float c = cond[i];
        float n2 = sqrtf(row_sumsq_torch_f32(c, lane));
        b = c / fmaxf(n2, 1e-12f);
    }
    float s = row_sum_torch(__fmul_rn(a, b), lane);
    if (lane == 0) {
        raw[tok] = (s <= thr) ? 1 : 0;                        // utils.py:333
        if (sim_out) sim_out[tok] = s;
    }
}

// ------------------------------------------------------------------------------------------------
// ARP phase 2: ONE workgroup (1024 threads): 3x3-cross erosion, 5x5 dilation (separable) in LDS,
// then wavefront ballot + popcount prefix-sum stream compaction -> ascending ids.
//   reference: utils.py:152-237 (morphology), utils.py:346-352 (boolean-mask compaction)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void arp_morph_compact_kernel(const uint8_t* __restrict__ raw, int H, int W,
                                                                 int ed, int64_t* __restrict__ edited,
                                                                 int64_t* __restrict__ unedited,
                                                                 uint8_t* __restrict__ mask_out,
                                                                 int32_t* __restrict__ count) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int L = H * W;
    uint8_t* A = smem;
    uint8_t* B = smem + ((L + 15) & ~15);
    __shared__ int wave_cnt[16];
    const int tid = threadIdx.x;
    for (int p = tid; p < L; p += 1024) A[p] = raw[p] ? 1 : 0;
    __syncthreads();
    const uint8_t* F = A;
    if (ed) {
        // erosion with the 3x3 cross, zero padding (border ring always eroded, quirk A-6)
        for (int p = tid; p < L; p += 1024) {
            int y = p / W, x = p - y * W;
            uint8_t v = A[p];
            v &= (y > 0) ? A[p - W] : 0;
            v &= (y < H - 1) ? A[p + W] : 0;
            v &= (x > 0) ? A[p - 1] : 0;
            v &= (x < W - 1) ? A[p + 1] : 0;
            B[p] = v;
        }
        __syncthreads();
        // 5x5 square dilation = horizontal 5-OR then vertical 5-OR
        for (int p = tid; p < L; p += 1024) {
            int y = p / W, x = p - y * W;
            uint8_t v = 0;
#pragma unroll
            for (int dx = -2; dx <= 2; ++dx) {
                int xx = x + dx;
                if (xx >= 0 && xx < W) v |= B[p + dx];
            }
            A[p] = v;
        }
        __syncthreads();
        for (int p = tid; p < L; p += 1024) {
            int y = p / W;
            uint8_t v = 0;
#pragma unroll
            for (int dy = -2; dy <= 2; ++dy) {
                int yy = y + dy;
                if (yy >= 0 && yy < H) v |= A[p + dy * W];
            }
            B[p] = v;
        }
        __syncthreads();
        F = B;
    }
    // compaction
    const int lane = tid & 63, wid = tid >> 6;
    int base_e = 0, base_u = 0;
    for (int c0 = 0; c0 < L; c0 += 1024) {
        const int p = c0 + tid;
        const bool valid = p < L;
        const bool f = valid && F[p];
        const unsigned long long bal = __ballot(f);
        if (lane == 0) wave_cnt[wid] = __popcll(bal);
        __syncthreads();
        int before = 0, total = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) {
            int c = wave_cnt[w];
            before += (w < wid) ? c : 0;
            total += c;
        }
        const int lane_before = __popcll(bal & ((1ull << lane) - 1ull));
        const int e_pos = before + lane_before;
        if (valid) {
            mask_out[p] = f ? 1 : 0;
            if (f) edited[base_e + e_pos] = p;
            else unedited[base_u + tid - e_pos] = p;
        }
        const int nvalid = min(1024, L - c0);
        base_e += total;
        base_u += nvalid - total;
        __syncthreads();
    }
    if (tid == 0) *count = base_e;
}

// ------------------------------------------------------------------------------------------------
// a3 gather / scatter of byte rows (utils.py:240-279)
// ------------------------------------------------------------------------------------------------
template <typename V, bool SCATTER>
__global__ void rows_kernel(const V* __restrict__ src, const int64_t* __restrict__ ids, V* __restrict__ dst,
                            int K, int vec_per_row) {
    const size_t total = (size_t)K * vec_per_row;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int k = (int)(i / vec_per_row), j = (int)(i - (size_t)k * vec_per_row);
        const size_t r = (size_t)ids[k];
        if (SCATTER) dst[r * vec_per_row + j] = src[(size_t)k * vec_per_row + j];
        else dst[(size_t)k * vec_per_row + j] = src[r * vec_per_row + j];
    }
}

// ------------------------------------------------------------------------------------------------
// a5 Euler / split-Euler step (inplace.py:641-686)
// ------------------------------------------------------------------------------------------------
template <typename TS, typename TV>
__global__ void euler_kernel(const TS* __restrict__ sample, const TV* __restrict__ v, TV* __restrict__ out,
                             const uint8_t* __restrict__ mask, float dt, float dt_direct, size_t n, int D) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float d = dt;
        if (mask != nullptr && !mask[i / D]) d = dt_direct;
        float x = ld_as_f32<TS>(sample, i);
        if constexpr (sizeof(TV) == 2) {
            float p = rbf(__fmul_rn(rbf(d), bf2f(v[i])));
            out[i] = f2bf(__fadd_rn(x, p));
        } else {
            out[i] = __fadd_rn(x, __fmul_rn(d, v[i]));
        }
    }
}

// ------------------------------------------------------------------------------------------------
// a6 AVD cache hit (inplace.py:315-318)
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void avd_kernel(const T* __restrict__ cache, const int64_t* __restrict__ ids, float ratio,
                           int round_ratio, T* __restrict__ out, size_t n, int D) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        size_t src = i;
        if (ids != nullptr) {
            size_t k = i / D;
            src = (size_t)ids[k] * D + (i - k * D);
        }
        if constexpr (sizeof(T) == 2) out[i] = f2bf(__fmul_rn(round_ratio ? rbf(ratio) : ratio, bf2f(cache[src])));
        else out[i] = __fmul_rn(ratio, cache[src]);
    }
}

// ------------------------------------------------------------------------------------------------
// a11 classifier-free-guidance combine, one wave per token row (D == 64), eager rounding points
//   mode 0  FLUX true-CFG          neg + s*(pos-neg)                                inplace.py:364
//   mode 1  Step1X norm-rescale    neg + s*(pos-neg)/f(||pos-neg||)                 Step1XEdit/inplace.py:401-410
//           f(n) = n>1 ? n^k : (n<1 ? 1 : n)   [EXT Step1XEditPipeline.process_diff_norm]
//   mode 2  Qwen norm-preserving   c = neg + s*(pos-neg); c * (||pos|| / ||c||)     QwenImageEdit/inplace.py:401-405
// ------------------------------------------------------------------------------------------------
template <typename T> __device__ __forceinline__ float rnd(float x);
template <> __device__ __forceinline__ float rnd<uint16_t>(float x) { return rbf(x); }
template <> __device__ __forceinline__ float rnd<float>(float x) { return x; }
template <typename T> __device__ __forceinline__ void st_from_f32(T* p, size_t i, float v);
template <> __device__ __forceinline__ void st_from_f32<uint16_t>(uint16_t* p, size_t i, float v) { p[i] = f2bf(v); }
template <> __device__ __forceinline__ void st_from_f32<float>(float* p, size_t i, float v) { p[i] = v; }

template <typename T>
__global__ __launch_bounds__(256) void cfg_combine_kernel(const T* __restrict__ pos, const T* __restrict__ neg,
                                                          T* __restrict__ out, float scale, int mode, float power,
                                                          int K) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= K) return;
    const size_t i = (size_t)row * 64 + lane;
    const float p = ld_as_f32<T>(pos, i), n = ld_as_f32<T>(neg, i);
    const float d = rnd<T>(__fsub_rn(p, n));                       // (pos - neg) in the tensor dtype
    float sd = rnd<T>(__fmul_rn(scale, d));                        // python-float scale: fp32 opmath, one rounding
    // torch.norm(x, dim=-1) on the CPU: bf16 rows reduce as the xor butterfly (= wave_sum), fp32 rows as 8 FMA lanes added
    // in order (row_sumsq_torch_f32) - the same trees the partition's cosine uses
    auto row_norm = [&](float x) {
        if constexpr (sizeof(T) == 2) return rnd<T>(sqrtf(wave_sum(__fmul_rn(x, x))));
        else return sqrtf(row_sumsq_torch_f32(x, lane));
    };
    if (mode == 1) {
        float nrm = row_norm(d);                                   // torch.norm(diff, dim=2, keepdim=True)
        float f = nrm;
        // torch.pow(tensor, python_float): the exponent is cast to the TENSOR dtype first (0.4 -> bf16 0.400390625)
        // (pow evaluated in double and rounded once: torch's CPU kernel is Sleef's 1-ulp powf, which a correctly rounded value
        // matches far more often than the device powf does; exact for bf16 tensors after their rounding)
        if (nrm > 1.0f) f = rnd<T>((float)pow((double)nrm, (double)rnd<T>(power)));
        else if (nrm < 1.0f) f = 1.0f;
        sd = rnd<T>(sd / f);
    }
    float c = rnd<T>(__fadd_rn(n, sd));
    if (mode == 2) {
        const float cn = row_norm(p);
        const float nn = row_norm(c);
        c = rnd<T>(__fmul_rn(c, rnd<T>(cn / nn)));
    }
    st_from_f32<T>(out, i, c);
}

static inline int grid_for(size_t n, int block) {
    size_t g = (n + block - 1) / block;
    return (int)(g < 1 ? 1 : (g > 2048 ? 2048 : g));
}

template <typename TS, typename TM>
static int launch_sim(const void* sample, const void* mo, const void* cond, int cond_dtype, float dt_final,
                      float thr, int L, uint8_t* raw, float* sim, hipStream_t st) {
    dim3 g((L + 3) / 4), b(256);
    if (cond_dtype == RGN_BF16)
        hipLaunchKernelGGL((arp_sim_kernel<TS, TM, uint16_t>), g, b, 0, st, (const TS*)sample, (const TM*)mo,
                           (const uint16_t*)cond, dt_final, thr, L, raw, sim);
    else
        hipLaunchKernelGGL((arp_sim_kernel<TS, TM, float>), g, b, 0, st, (const TS*)sample, (const TM*)mo,
                           (const float*)cond, dt_final, thr, L, raw, sim);
    return check_launch("arp_sim_kernel");
}

static int launch_morph(const uint8_t* raw, int H, int W, int ed, int64_t* e, int64_t* u, uint8_t* mask,
                        int32_t* count, hipStream_t st) {
    const int L = H * W;
    const size_t lds = 2 * (size_t)((L + 15) & ~15);
    if (lds > 150 * 1024) return fail(RGN_E_UNSUPPORTED, "arp: token grid larger than 76800 tokens");
    if (lds > 64 * 1024)          // the opt-in is per device and costs nothing next to the launch: no process-wide flag
        (void)hipFuncSetAttribute((const void*)arp_morph_compact_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    hipLaunchKernelGGL(arp_morph_compact_kernel, dim3(1), dim3(1024), lds, st, raw, H, W, ed, e, u, mask, count);
    return check_launch("arp_morph_compact_kernel");
}

// the launch-plan override table (common.h): all -1 unless RGN_PLAN_OVERRIDE="key=value,..." names fields - the library's only
// environment read, once per process
// sel[i] = i for the T text rows, sel[T + k] = T + edited_ids[k]: the cache rows a region step rewrites (`selection` of
// inplace.py:732-733: torch.cat((arange(T), edited_ids + T)) - three eager launches in the reference)
__global__ void sel_rows_kernel(const int64_t* __restrict__ ids, int K, int T, int64_t* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < T) out[i] = i;
    else if (i < T + K) out[i] = (int64_t)T + ids[i - T];
}

static int* plan_field(PlanOverride& o, const char* key) {
    struct { const char* name; int* field; } tab[] = {
        {"gemm_pieces", &o.gemm_pieces}, {"gemm_geometry", &o.gemm_geometry}, {"gemm_asm", &o.gemm_asm}, {"gemm_quarter", &o.gemm_quarter},
        {"attn_waves", &o.attn_waves}, {"attn_split", &o.attn_split}, {"attn_streamk", &o.attn_streamk}, {"attn_asm", &o.attn_asm}};
    for (auto& t : tab)
        if (strcmp(t.name, key) == 0) return t.field;
    return nullptr;
}

static bool plan_set(PlanOverride& o, const char* key, int value) {
    int* f = plan_field(o, key);
    if (f != nullptr) *f = value;
    return f != nullptr;
}

PlanOverride& plan_override() {
    static PlanOverride o = [] {
        PlanOverride v{-1, -1, -1, -1, -1, -1, -1, -1};
        const char* e = getenv("RGN_PLAN_OVERRIDE");
        if (e != nullptr) {
            char buf[256];
            snprintf(buf, sizeof(buf), "%s", e);
            char* save = nullptr;
            for (char* tok = strtok_r(buf, ",", &save); tok != nullptr; tok = strtok_r(nullptr, ",", &save)) {
                char* eq = strchr(tok, '=');
                if (eq != nullptr) *eq = 0;
                if (eq == nullptr || !plan_set(v, tok, atoi(eq + 1))) fprintf(stderr, "regione_hip: RGN_PLAN_OVERRIDE: bad entry '%s'\n", tok);
            }
        }
        return v;
    }();
    return o;
}

}  // namespace rgn

using namespace rgn;

extern "C" {

int rgn_version(void) { return RGN_ABI_VERSION; }
size_t rgn_abi_struct_bytes(void) { return sizeof(rgn_qkv_epilogue) * 1000 + sizeof(rgn_gemm_problem); }

int rgn_sel_rows(const int64_t* edited_ids, int K, int T, int64_t* out, void* stream) {
    if (K < 0 || T < 0 || (K > 0 && !edited_ids) || !out) return fail(RGN_E_BADARG, "sel_rows: bad argument");
    if (T + K == 0) return 0;
    hipLaunchKernelGGL(sel_rows_kernel, dim3((T + K + 255) / 256), dim3(256), 0, (hipStream_t)stream, edited_ids, K, T, out);
    return check_launch("sel_rows_kernel");
}

int rgn_fill_zero(void* ptr, size_t bytes, void* stream) {
    if (bytes == 0) return 0;
    if (!ptr) return fail(RGN_E_BADARG, "fill_zero: null pointer");
    hipError_t e = hipMemsetAsync(ptr, 0, bytes, (hipStream_t)stream);
    return e == hipSuccess ? 0 : fail((int)e, hipGetErrorString(e));
}

int rgn_plan_override(const char* key, int value) {
    PlanOverride& o = plan_override();
    if (key == nullptr) { o = PlanOverride{-1, -1, -1, -1, -1, -1, -1, -1}; return 0; }
    return plan_set(o, key, value) ? 0 : fail(RGN_E_BADARG, "plan_override: unknown key");
}
int rgn_plan_override_get(const char* key, int* value) {
    int* f = key != nullptr ? plan_field(plan_override(), key) : nullptr;
    if (f == nullptr || value == nullptr) return fail(RGN_E_BADARG, "plan_override_get: unknown key");
    *value = *f;
    return 0;
}
const char* rgn_last_error(void) { return g_err; }

int rgn_device_info(int* cu_count, int* clock_khz, size_t* hbm_bytes) {
    hipDeviceProp_t p;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e == hipSuccess) e = hipGetDeviceProperties(&p, dev);
    if (e != hipSuccess) return fail((int)e, hipGetErrorString(e));
    if (cu_count) *cu_count = p.multiProcessorCount;
    if (clock_khz) *clock_khz = p.clockRate;
    if (hbm_bytes) *hbm_bytes = p.totalGlobalMem;
    return 0;
}

int rgn_arp_partition(const void* sample, int sample_dtype, const void* model_output, int mo_dtype,
                      const void* cond, int cond_dtype, float dt_final, float threshold, int L, int D, int h_tok,
                      int w_tok, int erosion_dilation, int64_t* edited_ids, int64_t* unedited_ids,
                      uint8_t* raw_mask, uint8_t* mask, float* sim_out, int32_t* count, void* stream) {
    if (D != 64) return fail(RGN_E_UNSUPPORTED, "arp: latent width must be 64");
    if (L <= 0 || h_tok * w_tok != L) return fail(RGN_E_BADARG, "arp: h_tok*w_tok != L");
    if (!sample || !cond || !edited_ids || !unedited_ids || !raw_mask || !mask || !count)
        return fail(RGN_E_BADARG, "arp: null pointer");
    hipStream_t st = (hipStream_t)stream;
    int rc;
    const bool sb = sample_dtype == RGN_BF16, mb = mo_dtype == RGN_BF16;
    if (sb && mb) rc = launch_sim<uint16_t, uint16_t>(sample, model_output, cond, cond_dtype, dt_final, threshold, L, raw_mask, sim_out, st);
    else if (sb) rc = launch_sim<uint16_t, float>(sample, model_output, cond, cond_dtype, dt_final, threshold, L, raw_mask, sim_out, st);
    else if (mb) rc = launch_sim<float, uint16_t>(sample, model_output, cond, cond_dtype, dt_final, threshold, L, raw_mask, sim_out, st);
    else rc = launch_sim<float, float>(sample, model_output, cond, cond_dtype, dt_final, threshold, L, raw_mask, sim_out, st);
    if (rc) return rc;
    return launch_morph(raw_mask, h_tok, w_tok, erosion_dilation, edited_ids, unedited_ids, mask, count, st);
}

int rgn_morph_compact(const uint8_t* raw_mask, int h_tok, int w_tok, int erosion_dilation, int64_t* edited_ids,
                      int64_t* unedited_ids, uint8_t* mask, int32_t* count, void* stream) {
    if (!raw_mask || !edited_ids || !unedited_ids || !mask || !count || h_tok <= 0 || w_tok <= 0)
        return fail(RGN_E_BADARG, "morph_compact: bad argument");
    return launch_morph(raw_mask, h_tok, w_tok, erosion_dilation, edited_ids, unedited_ids, mask, count,
                        (hipStream_t)stream);
}

static int rows_op(const void* src, const int64_t* ids, void* dst, int K, int row_bytes, bool scatter, void* stream) {
    if (K == 0) return 0;
    if (!src || !ids || !dst || K < 0 || row_bytes <= 0 || (row_bytes & 3)) return fail(RGN_E_BADARG, "rows: bad argument");
    hipStream_t st = (hipStream_t)stream;
    const bool v16 = (row_bytes % 16 == 0) && (((uintptr_t)src | (uintptr_t)dst) % 16 == 0);
    if (v16) {
        int vpr = row_bytes / 16;
        int g = grid_for((size_t)K * vpr, 256);
        if (scatter) hipLaunchKernelGGL((rows_kernel<uint4, true>), dim3(g), dim3(256), 0, st, (const uint4*)src, ids, (uint4*)dst, K, vpr);
        else hipLaunchKernelGGL((rows_kernel<uint4, false>), dim3(g), dim3(256), 0, st, (const uint4*)src, ids, (uint4*)dst, K, vpr);
    } else {
        int vpr = row_bytes / 4;
        int g = grid_for((size_t)K * vpr, 256);
        if (scatter) hipLaunchKernelGGL((rows_kernel<uint32_t, true>), dim3(g), dim3(256), 0, st, (const uint32_t*)src, ids, (uint32_t*)dst, K, vpr);
        else hipLaunchKernelGGL((rows_kernel<uint32_t, false>), dim3(g), dim3(256), 0, st, (const uint32_t*)src, ids, (uint32_t*)dst, K, vpr);
    }
    return check_launch("rows_kernel");
}

int rgn_gather_rows(const void* src, const int64_t* ids, void* dst, int K, int row_bytes, void* stream) {
    return rows_op(src, ids, dst, K, row_bytes, false, stream);
}
int rgn_scatter_rows(const void* src, const int64_t* ids, void* dst, int K, int row_bytes, void* stream) {
    return rows_op(src, ids, dst, K, row_bytes, true, stream);
}

int rgn_euler_step(const void* sample, int sample_dtype, const void* v, int v_dtype, void* out, const uint8_t* mask,
                   float dt, float dt_direct, int L, int D, void* stream) {
    if (L == 0) return 0;
    if (!sample || !v || !out || L < 0 || D <= 0) return fail(RGN_E_BADARG, "euler: bad argument");
    hipStream_t st = (hipStream_t)stream;
    size_t n = (size_t)L * D;
    int g = grid_for(n, 256);
    const bool sb = sample_dtype == RGN_BF16, vb = v_dtype == RGN_BF16;
    if (sb && vb) hipLaunchKernelGGL((euler_kernel<uint16_t, uint16_t>), dim3(g), dim3(256), 0, st, (const uint16_t*)sample, (const uint16_t*)v, (uint16_t*)out, mask, dt, dt_direct, n, D);
    else if (sb) hipLaunchKernelGGL((euler_kernel<uint16_t, float>), dim3(g), dim3(256), 0, st, (const uint16_t*)sample, (const float*)v, (float*)out, mask, dt, dt_direct, n, D);
    else if (vb) hipLaunchKernelGGL((euler_kernel<float, uint16_t>), dim3(g), dim3(256), 0, st, (const float*)sample, (const uint16_t*)v, (uint16_t*)out, mask, dt, dt_direct, n, D);
    else hipLaunchKernelGGL((euler_kernel<float, float>), dim3(g), dim3(256), 0, st, (const float*)sample, (const float*)v, (float*)out, mask, dt, dt_direct, n, D);
    return check_launch("euler_kernel");
}

int rgn_cfg_combine(const void* pos, const void* neg, void* out, int dtype, float scale, int mode, float power, int K,
                    int D, void* stream) {
    if (K == 0) return 0;
    if (!pos || !neg || !out || K < 0 || mode < 0 || mode > 2) return fail(RGN_E_BADARG, "cfg_combine: bad argument");
    if (D != 64) return fail(RGN_E_UNSUPPORTED, "cfg_combine: latent width must be 64");
    hipStream_t st = (hipStream_t)stream;
    dim3 g((K + 3) / 4), b(256);
    if (dtype == RGN_BF16) hipLaunchKernelGGL((cfg_combine_kernel<uint16_t>), g, b, 0, st, (const uint16_t*)pos, (const uint16_t*)neg, (uint16_t*)out, scale, mode, power, K);
    else hipLaunchKernelGGL((cfg_combine_kernel<float>), g, b, 0, st, (const float*)pos, (const float*)neg, (float*)out, scale, mode, power, K);
    return check_launch("cfg_combine_kernel");
}

int rgn_avd_apply(const void* cache, int dtype, const int64_t* ids, float ratio, int round_ratio, void* out, int K,
                  int D, void* stream) {
    if (K == 0) return 0;
    if (!cache || !out || K < 0 || D <= 0) return fail(RGN_E_BADARG, "avd: bad argument");
    hipStream_t st = (hipStream_t)stream;
    size_t n = (size_t)K * D;
    int g = grid_for(n, 256);
    if (dtype == RGN_BF16) hipLaunchKernelGGL((avd_kernel<uint16_t>), dim3(g), dim3(256), 0, st, (const uint16_t*)cache, ids, ratio, round_ratio, (uint16_t*)out, n, D);
    else hipLaunchKernelGGL((avd_kernel<float>), dim3(g), dim3(256), 0, st, (const float*)cache, ids, ratio, round_ratio, (float*)out, n, D);
    return check_launch("avd_kernel");
}

}  // extern "C"
