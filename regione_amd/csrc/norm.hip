// Row-wise kernels of the masked MMDiT block: AdaLN modulate, per-head RMSNorm + RoPE on Q/K with
// placement of K / V^T into the Region-Instruction KV-cache slab.  All HBM-streaming; rounding
// points follow the eager bf16 op sequence of the [EXT] diffusers modules the reference calls
// (inplace.py:518-555 blocks, :760-763 norm_q/norm_k, :792-794 apply_rotary_emb).
#include "common.h"

namespace rgn {

// ------------------------------------------------------------------------------------------------
// y = bf16(bf16(bf16(LN(x)) * bf16(1 + scale)) + shift), one 256-thread block per row, two-pass stats
// ------------------------------------------------------------------------------------------------
// Row segments with their own modulation vectors: rows [end[i-1], end[i]) use (shift[i], scale[i]) - the text / image
// streams of a double block, times the CFG branches of a batched forward (per-branch AdaLN vectors).
constexpr int LN_MAXSEG = 4;
struct LnSegs {
    int end[LN_MAXSEG];
    const uint16_t* shift[LN_MAXSEG];
    const uint16_t* scale[LN_MAXSEG];
};

__global__ __launch_bounds__(256) void ln_modulate_kernel(const uint16_t* __restrict__ x, int ldx,
                                                          uint16_t* __restrict__ out, int ldo, int d, float eps,
                                                          const LnSegs segs) {
    __shared__ float red[8];
    const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint16_t* xr = x + (size_t)row * ldx;
    int si = 0;
#pragma unroll
    for (int i = 0; i < LN_MAXSEG - 1; ++i) si += (row >= segs.end[i]) ? 1 : 0;
    const uint16_t* sh = segs.shift[si];
    const uint16_t* sc = segs.scale[si];
    const int nvec = d >> 3;
    constexpr int MAXV = 4;                       // d <= 8192
    float v[MAXV][8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int vi = tid + i * 256;
        if (vi < nvec) {
            uint16_t t[8];
            *(uint4*)t = *(const uint4*)(xr + vi * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) { v[i][e] = bf2f(t[e]); s += v[i][e]; }
        }
    }
    s = wave_sum(s);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    const float mean = (red[0] + red[1] + red[2] + red[3]) / (float)d;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int vi = tid + i * 256;
        if (vi < nvec) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float c = v[i][e] - mean; q += c * c; }
        }
    }
    q = wave_sum(q);
    if (lane == 0) red[4 + wave] = q;
    __syncthreads();
    const float var = (red[4] + red[5] + red[6] + red[7]) / (float)d;
    const float rstd = 1.0f / sqrtf(var + eps);
    uint16_t* orow = out + (size_t)row * ldo;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int vi = tid + i * 256;
        if (vi < nvec) {
            uint16_t a[8], b[8], o[8];
            *(uint4*)a = *(const uint4*)(sc + vi * 8);
            *(uint4*)b = *(const uint4*)(sh + vi * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float y0 = rbf((v[i][e] - mean) * rstd);          // LayerNorm output (bf16)
                const float s1 = rbf(1.0f + bf2f(a[e]));                // (1 + scale) (bf16)
                const float y1 = rbf(y0 * s1);
                o[e] = f2bf(y1 + bf2f(b[e]));
            }
            *(uint4*)(orow + vi * 8) = *(const uint4*)o;
        }
    }
}

// Round 5: the same arithmetic with ONE WAVE per row (4 rows per 256-thread block), for d a multiple of 512.  The row stays in registers
// as the bf16 bits it arrived in (NV uint4 per lane: 24 VGPRs at d = 3072 - the round-4 attempt kept 48 fp32 values per lane and lost
// its occupancy) and is widened on the fly in each of the three passes; no LDS, no block barriers, every thread has the same amount of
// work (the block-per-row kernel above gives threads 0-127 two vectors of a 3072-wide row and threads 128-255 one).  Sums run per lane
// over its NV x 8 elements, then across the wave (xor butterfly): another fp32 summation order than the block kernel's (same box A/B,
// profiles/r05_ln_wave_ab.txt: 8704 x 3072 rows 25.8 -> 20.7 us = 4.15 -> 5.16 TB/s, 33280 rows 92 -> 74 us; 4 outputs in a million
// differ, by one bf16 ulp).
template <int NV>
__global__ __launch_bounds__(256) void ln_modulate_wave_kernel(const uint16_t* __restrict__ x, int ldx, uint16_t* __restrict__ out, int ldo,
                                                               int M, float eps, const LnSegs segs) {
    constexpr int d = NV * 512;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    int si = 0;
#pragma unroll
    for (int i = 0; i < LN_MAXSEG - 1; ++i) si += (row >= segs.end[i]) ? 1 : 0;
    const uint16_t* sh = segs.shift[si];
    const uint16_t* sc = segs.scale[si];
    const uint16_t* xr = x + (size_t)row * ldx + lane * 8;
    uint4 xv[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) xv[i] = *(const uint4*)(xr + i * 512);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const uint32_t w[4] = {xv[i].x, xv[i].y, xv[i].z, xv[i].w};
#pragma unroll
        for (int k = 0; k < 4; ++k) { s += __uint_as_float(w[k] << 16); s += __uint_as_float(w[k] & 0xffff0000u); }
    }
    const float mean = wave_sum(s) / (float)d;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const uint32_t w[4] = {xv[i].x, xv[i].y, xv[i].z, xv[i].w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float c0 = __uint_as_float(w[k] << 16) - mean, c1 = __uint_as_float(w[k] & 0xffff0000u) - mean;
            q += c0 * c0;
            q += c1 * c1;
        }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)d + eps);
    uint16_t* orow = out + (size_t)row * ldo + lane * 8;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const uint4 a4 = *(const uint4*)(sc + lane * 8 + i * 512), b4 = *(const uint4*)(sh + lane * 8 + i * 512);
        const uint32_t w[4] = {xv[i].x, xv[i].y, xv[i].z, xv[i].w}, a[4] = {a4.x, a4.y, a4.z, a4.w}, b[4] = {b4.x, b4.y, b4.z, b4.w};
        uint32_t o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float y[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const float xe = __uint_as_float(h ? (w[k] & 0xffff0000u) : (w[k] << 16));
                const float ae = __uint_as_float(h ? (a[k] & 0xffff0000u) : (a[k] << 16));
                const float be = __uint_as_float(h ? (b[k] & 0xffff0000u) : (b[k] << 16));
                const float y0 = rbf((xe - mean) * rstd);          // LayerNorm output (bf16)
                const float s1 = rbf(1.0f + ae);                   // (1 + scale) (bf16)
                y[h] = rbf(y0 * s1) + be;
            }
            o[k] = f2bf_pk(y[0], y[1]);
        }
        *(uint4*)(orow + i * 512) = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

static int launch_ln(const uint16_t* x, int ldx, uint16_t* out, int ldo, int M, int d, float eps, const LnSegs& segs, hipStream_t st) {
    if (d % 512 == 0) {          // the wave-per-row kernel; other widths (toy trunks, d = 256): the block-per-row kernel below
        const dim3 grid((M + 3) / 4), blk(256);
        switch (d / 512) {
            case 1: hipLaunchKernelGGL(ln_modulate_wave_kernel<1>, grid, blk, 0, st, x, ldx, out, ldo, M, eps, segs); return check_launch("ln_modulate_wave_kernel");
            case 2: hipLaunchKernelGGL(ln_modulate_wave_kernel<2>, grid, blk, 0, st, x, ldx, out, ldo, M, eps, segs); return check_launch("ln_modulate_wave_kernel");
            case 3: hipLaunchKernelGGL(ln_modulate_wave_kernel<3>, grid, blk, 0, st, x, ldx, out, ldo, M, eps, segs); return check_launch("ln_modulate_wave_kernel");
            case 4: hipLaunchKernelGGL(ln_modulate_wave_kernel<4>, grid, blk, 0, st, x, ldx, out, ldo, M, eps, segs); return check_launch("ln_modulate_wave_kernel");
            case 6: hipLaunchKernelGGL(ln_modulate_wave_kernel<6>, grid, blk, 0, st, x, ldx, out, ldo, M, eps, segs); return check_launch("ln_modulate_wave_kernel");
            case 8: hipLaunchKernelGGL(ln_modulate_wave_kernel<8>, grid, blk, 0, st, x, ldx, out, ldo, M, eps, segs); return check_launch("ln_modulate_wave_kernel");
            default: break;
        }
    }
    hipLaunchKernelGGL(ln_modulate_kernel, dim3(M), dim3(256), 0, st, x, ldx, out, ldo, d, eps, segs);
    return check_launch("ln_modulate_kernel");
}

// ------------------------------------------------------------------------------------------------
// Q/K: per-head RMSNorm + RoPE.  One wave per token row, lane = one rotary pair (2 of 128 dims),
// loop over heads so the cos/sin pairs are loaded once per row.
//   q: in place.   k: -> K slab row (kv_rows ? kv_rows[m] : m), rotated with the FULL-id table.
// ------------------------------------------------------------------------------------------------
// Sum of the 64 lanes' values in the order the fused Q/K/V epilogue of the GEMM uses (gemm.hip: qk_norm_rope_vec - there a lane holds
// FOUR rotary pairs and adds them before its 16-lane butterfly): neighbours 2 and 1 apart first, then 32, 16, 8, 4.  Every lane ends
// with the same value; the two paths agree bit for bit (tests/test_gpu_kernels.py).
__device__ __forceinline__ float head_sum(float v) {
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 32, 64);
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 8, 64);
    v += __shfl_xor(v, 4, 64);
    return v;
}

__global__ __launch_bounds__(256) void qk_norm_rope_kernel(uint16_t* __restrict__ qkv, int ld, int k_col, int q_col,
                                                           int M, int H, int split_row,
                                                           const uint16_t* __restrict__ wq0, const uint16_t* __restrict__ wk0,
                                                           const uint16_t* __restrict__ wq1, const uint16_t* __restrict__ wk1,
                                                           float eps, const float* __restrict__ cos_q,
                                                           const float* __restrict__ sin_q, const float* __restrict__ cos_k,
                                                           const float* __restrict__ sin_k,
                                                           const int64_t* __restrict__ kv_rows,
                                                           uint16_t* __restrict__ k_slab) {
    const int lane = threadIdx.x & 63;
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    const size_t kr = kv_rows ? (size_t)kv_rows[m] : (size_t)m;
    const uint16_t* wq = m < split_row ? wq0 : wq1;
    const uint16_t* wk = m < split_row ? wk0 : wk1;
    const float2 cq = *(const float2*)(cos_q + (size_t)m * 128 + lane * 2);
    const float2 sq = *(const float2*)(sin_q + (size_t)m * 128 + lane * 2);
    const float2 ck = *(const float2*)(cos_k + kr * 128 + lane * 2);
    const float2 sk = *(const float2*)(sin_k + kr * 128 + lane * 2);
    const uint32_t wqp = *(const uint32_t*)(wq + lane * 2), wkp = *(const uint32_t*)(wk + lane * 2);
    const float wq_a = bf2f(wqp & 0xffff), wq_b = bf2f(wqp >> 16), wk_a = bf2f(wkp & 0xffff), wk_b = bf2f(wkp >> 16);
    uint16_t* row = qkv + (size_t)m * ld;
    const size_t HD = (size_t)H * 128;
#pragma unroll 4
    for (int h = 0; h < H; ++h) {
        uint32_t* qp = (uint32_t*)(row + q_col + h * 128 + lane * 2);
        const uint32_t qv = *qp;
        const uint32_t kv = *(const uint32_t*)(row + k_col + h * 128 + lane * 2);
        float qa = bf2f(qv & 0xffff), qb = bf2f(qv >> 16), ka = bf2f(kv & 0xffff), kb = bf2f(kv >> 16);
        // RMSNorm: fp32 variance, x*rsqrt in fp32, round to bf16, times bf16 weight (round)
        const float rq = 1.0f / sqrtf(head_sum(qa * qa + qb * qb) * (1.0f / 128.0f) + eps);
        const float rk = 1.0f / sqrtf(head_sum(ka * ka + kb * kb) * (1.0f / 128.0f) + eps);
        qa = rbf(rbf(qa * rq) * wq_a); qb = rbf(rbf(qb * rq) * wq_b);
        ka = rbf(rbf(ka * rk) * wk_a); kb = rbf(rbf(kb * rk) * wk_b);
        // RoPE (fp32): out[2i] = x[2i]*cos - x[2i+1]*sin ; out[2i+1] = x[2i+1]*cos + x[2i]*sin
        const float q0 = __fadd_rn(__fmul_rn(qa, cq.x), __fmul_rn(-qb, sq.x));
        const float q1 = __fadd_rn(__fmul_rn(qb, cq.y), __fmul_rn(qa, sq.y));
        const float k0 = __fadd_rn(__fmul_rn(ka, ck.x), __fmul_rn(-kb, sk.x));
        const float k1 = __fadd_rn(__fmul_rn(kb, ck.y), __fmul_rn(ka, sk.y));
        *qp = (uint32_t)f2bf(q0) | ((uint32_t)f2bf(q1) << 16);
        *(uint32_t*)(k_slab + kr * HD + h * 128 + lane * 2) = (uint32_t)f2bf(k0) | ((uint32_t)f2bf(k1) << 16);
    }
}

// ------------------------------------------------------------------------------------------------
// V: transpose 64 rows x 128 dims (one head) through LDS into the V^T slab, kv index permuted inside
// every 16-group (bits 2<->3 swapped): that is the order in which the attention kernel's S^T
// accumulator registers line up as the P operand, so its V^T operand is one 16-byte LDS read.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ size_t kvpos(size_t r) { return (r & ~(size_t)12) | ((r & 4) << 1) | ((r & 8) >> 1); }

__global__ __launch_bounds__(256) void v_transpose_store_kernel(const uint16_t* __restrict__ qkv, int ld, int v_col,
                                                                int M, const int64_t* __restrict__ kv_rows,
                                                                uint16_t* __restrict__ vt_slab, int skv_pad) {
    __shared__ uint16_t tile[64][130];
    __shared__ size_t dpos[64];
    const int tid = threadIdx.x, m0 = blockIdx.x * 64, h = blockIdx.y;
    // load: 64 rows x 256 B, 16 B per thread per pass
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int r = (tid >> 4) + it * 16, c = (tid & 15) * 8;
        const int m = min(m0 + r, M - 1);
        uint16_t t[8];
        *(uint4*)t = *(const uint4*)(qkv + (size_t)m * ld + v_col + h * 128 + c);
#pragma unroll
        for (int e = 0; e < 8; ++e) tile[r][c + e] = t[e];
    }
    if (tid < 64) {
        const int m = m0 + tid;
        dpos[tid] = (m < M) ? kvpos(kv_rows ? (size_t)kv_rows[m] : (size_t)m) : (size_t)-1;
    }
    __syncthreads();
    const int r = tid & 63;
    const size_t p = dpos[r];
    if (p == (size_t)-1) return;
#pragma unroll 8
    for (int it = 0; it < 32; ++it) {
        const int dd = (tid >> 6) + it * 4;
        vt_slab[((size_t)h * 128 + dd) * skv_pad + p] = tile[r][dd];
    }
}

// Row RMSNorm with affine weight (Qwen `txt_norm` on the prompt embeddings, [EXT] RMSNorm semantics:
// fp32 variance, x*rsqrt in fp32, round to bf16, times bf16 weight).
__global__ __launch_bounds__(256) void rms_norm_rows_kernel(const uint16_t* __restrict__ x, int ldx,
                                                            const uint16_t* __restrict__ w, uint16_t* __restrict__ out,
                                                            int ldo, int d, float eps) {
    __shared__ float red[4];
    const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint16_t* xr = x + (size_t)row * ldx;
    float s = 0.f;
    for (int i = tid; i < d; i += 256) { const float v = bf2f(xr[i]); s += v * v; }
    s = wave_sum(s);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    const float r = 1.0f / sqrtf((red[0] + red[1] + red[2] + red[3]) / (float)d + eps);
    for (int i = tid; i < d; i += 256) out[(size_t)row * ldo + i] = f2bf(rbf(bf2f(xr[i]) * r) * bf2f(w[i]));
}

__global__ void silu_bf16_kernel(const uint16_t* __restrict__ x, uint16_t* __restrict__ y, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float v = bf2f(x[i]);
        y[i] = f2bf(v / (1.0f + expf(-v)));
    }
}

// y = bf16(a + b) elementwise on bf16 (fp32 add, one rounding: what torch's bf16 add computes) - the two adds of
// CombinedTimestepGuidanceTextProjEmbeddings [EXT]; the result may alias an input
__global__ void add_bf16_kernel(const uint16_t* __restrict__ a, const uint16_t* __restrict__ b, uint16_t* __restrict__ y, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        y[i] = f2bf(bf2f(a[i]) + bf2f(b[i]));
}

}  // namespace rgn

using namespace rgn;

extern "C" {

int rgn_add_bf16(const void* a, const void* b, void* y, size_t n, void* stream) {
    if (n == 0) return 0;
    if (!a || !b || !y) return fail(RGN_E_BADARG, "add: null pointer");
    size_t g = (n + 255) / 256;
    hipLaunchKernelGGL(add_bf16_kernel, dim3((unsigned)(g > 1024 ? 1024 : g)), dim3(256), 0, (hipStream_t)stream,
                       (const uint16_t*)a, (const uint16_t*)b, (uint16_t*)y, n);
    return check_launch("add_bf16_kernel");
}

int rgn_ln_modulate(const void* x, int ldx, void* out, int ldo, int M, int d, float eps, int split_row,
                    const void* shift0, const void* scale0, const void* shift1, const void* scale1, void* stream) {
    if (M == 0) return 0;
    if (!x || !out || M < 0 || d <= 0 || (d % 8) || d > 8192 || (ldx % 8) || (ldo % 8) || !shift1 || !scale1)
        return fail(RGN_E_BADARG, "ln_modulate: bad argument");
    if (split_row > 0 && (!shift0 || !scale0)) return fail(RGN_E_BADARG, "ln_modulate: stream-0 modulation missing");
    LnSegs segs;
    for (int i = 0; i < LN_MAXSEG; ++i) {
        segs.end[i] = i == 0 ? (split_row > 0 ? split_row : 0) : M;
        segs.shift[i] = (const uint16_t*)(i == 0 && split_row > 0 ? shift0 : shift1);
        segs.scale[i] = (const uint16_t*)(i == 0 && split_row > 0 ? scale0 : scale1);
    }
    return launch_ln((const uint16_t*)x, ldx, (uint16_t*)out, ldo, M, d, eps, segs, (hipStream_t)stream);
}

int rgn_ln_modulate_segs(const void* x, int ldx, void* out, int ldo, int M, int d, float eps, int nseg, const int* seg_end_host,
                         const void* const* shift_host, const void* const* scale_host, void* stream) {
    if (M == 0) return 0;
    if (!x || !out || M < 0 || d <= 0 || (d % 8) || d > 8192 || (ldx % 8) || (ldo % 8) || nseg < 1 || nseg > LN_MAXSEG ||
        !seg_end_host || !shift_host || !scale_host)
        return fail(RGN_E_BADARG, "ln_modulate_segs: bad argument");
    LnSegs segs;
    int prev = 0;
    for (int i = 0; i < LN_MAXSEG; ++i) {
        const int j = i < nseg ? i : nseg - 1;
        if (!shift_host[j] || !scale_host[j] || seg_end_host[j] < prev) return fail(RGN_E_BADARG, "ln_modulate_segs: segment table");
        segs.end[i] = i < nseg - 1 ? seg_end_host[i] : M;
        segs.shift[i] = (const uint16_t*)shift_host[j];
        segs.scale[i] = (const uint16_t*)scale_host[j];
        prev = seg_end_host[j];
    }
    if (seg_end_host[nseg - 1] != M) return fail(RGN_E_BADARG, "ln_modulate_segs: the last segment must end at M");
    return launch_ln((const uint16_t*)x, ldx, (uint16_t*)out, ldo, M, d, eps, segs, (hipStream_t)stream);
}

int rgn_rms_norm_rows(const void* x, int ldx, const void* w, void* out, int ldo, int M, int d, float eps, void* stream) {
    if (M == 0) return 0;
    if (!x || !w || !out || M < 0 || d <= 0) return fail(RGN_E_BADARG, "rms_norm_rows: bad argument");
    hipLaunchKernelGGL(rms_norm_rows_kernel, dim3(M), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)x, ldx,
                       (const uint16_t*)w, (uint16_t*)out, ldo, d, eps);
    return check_launch("rms_norm_rows_kernel");
}

int rgn_silu_bf16(const void* x, void* y, size_t n, void* stream) {
    if (n == 0) return 0;
    if (!x || !y) return fail(RGN_E_BADARG, "silu: null pointer");
    size_t g = (n + 255) / 256;
    hipLaunchKernelGGL(silu_bf16_kernel, dim3((unsigned)(g > 1024 ? 1024 : g)), dim3(256), 0, (hipStream_t)stream,
                       (const uint16_t*)x, (uint16_t*)y, n);
    return check_launch("silu_bf16_kernel");
}

int rgn_qk_norm_rope_store(void* qkv, int ld, int k_col, int v_col, int q_col, int M, int H, int split_row,
                           const void* wq0, const void* wk0, const void* wq1, const void* wk1, float eps,
                           const float* cos_q, const float* sin_q, const float* cos_k, const float* sin_k,
                           const int64_t* kv_rows, void* k_slab, void* vt_slab, int skv_pad, void* stream) {
    if (M == 0) return 0;
    if (!qkv || !wq1 || !wk1 || !cos_q || !sin_q || !cos_k || !sin_k || !k_slab || !vt_slab || M < 0 || H <= 0 ||
        (ld % 8) || (k_col % 8) || (v_col % 8) || (q_col % 8) || (skv_pad % 64))
        return fail(RGN_E_BADARG, "qk_norm_rope_store: bad argument");
    if (split_row > 0 && (!wq0 || !wk0)) return fail(RGN_E_BADARG, "qk_norm_rope_store: stream-0 norm weights missing");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(qk_norm_rope_kernel, dim3((M + 3) / 4), dim3(256), 0, st, (uint16_t*)qkv, ld, k_col, q_col, M, H,
                       split_row, (const uint16_t*)wq0, (const uint16_t*)wk0, (const uint16_t*)wq1,
                       (const uint16_t*)wk1, eps, cos_q, sin_q, cos_k, sin_k, kv_rows, (uint16_t*)k_slab);
    int rc = check_launch("qk_norm_rope_kernel");
    if (rc) return rc;
    hipLaunchKernelGGL(v_transpose_store_kernel, dim3((M + 63) / 64, H), dim3(256), 0, st, (const uint16_t*)qkv, ld,
                       v_col, M, kv_rows, (uint16_t*)vt_slab, skv_pad);
    return check_launch("v_transpose_store_kernel");
}

}  // extern "C"
