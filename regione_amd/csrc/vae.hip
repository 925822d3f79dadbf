// Row kernels of the VAE decoder (reference call site FluxKontext/inplace.py:396-402 `self.vae.decode(latents)`,
// [EXT] AutoencoderKL of the public FLUX.1 / Step1X-Edit checkpoints) - HBM-streaming passes around the implicit-GEMM convolutions
// (rgn_conv_bf16, gemm.hip).
//
// Activation layout: a ZERO-BORDERED, pixel-major image [Hp * Wp, C] bf16 with Hp = H + 2, Wp = W + 2 (row = y * Wp + x of the padded
// image, channels contiguous): a 3 x 3 tap is then a constant row shift, a convolution a GEMM whose A tile moves by a byte offset per
// K step, and the zero border IS the convolution's zero padding.  Every kernel here writes zeros to the border rows it produces.
//   gn_stats_kernel + gn_finalize_kernel   GroupNorm(32 groups) statistics: per-block fp32 partial sums, one double-precision
//                                           finalize block -> (mean, rstd) per group; no atomics: bit-reproducible
//   gn_apply_kernel                         y = (x - mean) * rstd * gamma + beta, optional SiLU, border rows -> 0
//   upsample2x_kernel                       nearest-neighbour 2 x upsample into the next level's padded image
//   softmax_rows_kernel                     mid-block attention: P = softmax(scale * S) per row over the valid (non-border) key columns
//   nchw_to_padded_kernel / padded_to_nchw_kernel   the host tensor layouts on either side of the decoder
#include "common.h"

namespace rgn {

__device__ __forceinline__ uint32_t pk_bf16(float a, float b) { return f2bf_pk(a, b); }     // v_cvt_pk_bf16_f32 (RNE)
__device__ __forceinline__ float lo_bf(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float hi_bf(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }

// ---- GroupNorm statistics -------------------------------------------------------------------------------------------------------
// One thread = one 8-channel vector of a row; 256 / (C / 8) rows per block pass, grid-stride over the rows.  A vector's two 4-channel
// halves are accumulated apart (C = 128: 4 channels per group); a block folds its threads to 32 groups x (sum, sum of squares) in LDS.
__global__ __launch_bounds__(256) void gn_stats_kernel(const uint16_t* __restrict__ X, int rows, int C, float* __restrict__ partial) {
    const int vpr = C >> 3, rpb = 256 / vpr;
    const int tid = threadIdx.x, v = tid % vpr, rl = tid / vpr;
    float s0 = 0.f, q0 = 0.f, s1 = 0.f, q1 = 0.f;
    for (int r = blockIdx.x * rpb + rl; r < rows; r += gridDim.x * rpb) {
        const uint4 w = *(const uint4*)(X + (size_t)r * C + v * 8);
        const float a0 = lo_bf(w.x), a1 = hi_bf(w.x), a2 = lo_bf(w.y), a3 = hi_bf(w.y);
        const float b0 = lo_bf(w.z), b1 = hi_bf(w.z), b2 = lo_bf(w.w), b3 = hi_bf(w.w);
        s0 += (a0 + a1) + (a2 + a3);
        q0 += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
        s1 += (b0 + b1) + (b2 + b3);
        q1 += (b0 * b0 + b1 * b1) + (b2 * b2 + b3 * b3);
    }
    // block fold in a FIXED order (no atomics: the statistics are bit-reproducible): thread t < 64 owns (group t >> 1, quantity t & 1)
    // and walks the 256 threads' values of that quantity, taking the halves that belong to its group
    __shared__ float sh[4][256];
    sh[0][tid] = s0; sh[1][tid] = q0; sh[2][tid] = s1; sh[3][tid] = q1;
    __syncthreads();
    if (tid < 64) {
        const int cpg = C >> 5, g = tid >> 1, qn = tid & 1;         // channels per group (4, 8, 16)
        float acc = 0.f;
        for (int i = 0; i < 256; ++i) {
            const int vi = i % vpr;
            if ((vi * 8) / cpg == g) acc += sh[qn][i];
            if ((vi * 8 + 4) / cpg == g) acc += sh[2 + qn][i];
        }
        partial[(size_t)blockIdx.x * 64 + tid] = acc;
    }
}

__global__ __launch_bounds__(1024) void gn_finalize_kernel(const float* __restrict__ partial, int nblocks, double count, float eps,
                                                           float* __restrict__ stats) {
    // thread: quantity t = tid & 63 (group t >> 1, sum / sum of squares t & 1), slice tid >> 6 of the blocks; 16 slices folded in a
    // fixed order in double precision (a single 64-thread block walking 1024 partials took 200 us: 30 such launches per decode)
    const int t = threadIdx.x & 63, part = threadIdx.x >> 6;
    double acc = 0.0;
    for (int b = part; b < nblocks; b += 16) acc += (double)partial[(size_t)b * 64 + t];
    __shared__ double sh[16][64];
    sh[part][t] = acc;
    __syncthreads();
    if (threadIdx.x < 64) {
        double tot = 0.0;
#pragma unroll
        for (int i = 0; i < 16; ++i) tot += sh[i][t];
        const double other = __shfl_xor(tot, 1, 64);
        if ((t & 1) == 0) {
            const double mean = tot / count, var = other / count - mean * mean;
            stats[t] = (float)mean;
            stats[t + 1] = (float)(1.0 / sqrt((var > 0.0 ? var : 0.0) + (double)eps));
        }
    }
}

__global__ __launch_bounds__(256) void gn_apply_kernel(const uint16_t* __restrict__ X, uint16_t* __restrict__ Y, int Hp, int Wp, int C,
                                                       const float* __restrict__ stats, const uint16_t* __restrict__ gamma,
                                                       const uint16_t* __restrict__ beta, int silu) {
    const int vpr = C >> 3, rpb = 256 / vpr, rows = Hp * Wp;
    const int tid = threadIdx.x, v = tid % vpr, rl = tid / vpr;
    const int cpg = C >> 5;
    const int g0 = (v * 8) / cpg, g1 = (v * 8 + 4) / cpg;
    const float m0 = stats[g0 * 2], r0 = stats[g0 * 2 + 1], m1 = stats[g1 * 2], r1 = stats[g1 * 2 + 1];
    const uint4 gw = *(const uint4*)(gamma + v * 8), bw = *(const uint4*)(beta + v * 8);
    const uint32_t gws[4] = {gw.x, gw.y, gw.z, gw.w}, bws[4] = {bw.x, bw.y, bw.z, bw.w};
    float sc[8], sf[8];                              // y = x * sc + sf
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float ga = (e & 1) ? hi_bf(gws[e >> 1]) : lo_bf(gws[e >> 1]);
        const float be = (e & 1) ? hi_bf(bws[e >> 1]) : lo_bf(bws[e >> 1]);
        const float mean = e < 4 ? m0 : m1, rstd = e < 4 ? r0 : r1;
        sc[e] = rstd * ga;
        sf[e] = be - mean * rstd * ga;
    }
    for (int r = blockIdx.x * rpb + rl; r < rows; r += gridDim.x * rpb) {
        const int py = r / Wp, px = r - py * Wp;
        uint4 o = make_uint4(0u, 0u, 0u, 0u);
        if (px != 0 && px != Wp - 1 && py != 0 && py != Hp - 1) {
            const uint4 w = *(const uint4*)(X + (size_t)r * C + v * 8);
            const uint32_t ws[4] = {w.x, w.y, w.z, w.w};
            float y[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float x = (e & 1) ? hi_bf(ws[e >> 1]) : lo_bf(ws[e >> 1]);
                float t = __builtin_fmaf(x, sc[e], sf[e]);
                if (silu) t = t * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * t));
                y[e] = t;
            }
            o = make_uint4(pk_bf16(y[0], y[1]), pk_bf16(y[2], y[3]), pk_bf16(y[4], y[5]), pk_bf16(y[6], y[7]));
        }
        *(uint4*)(Y + (size_t)r * C + v * 8) = o;
    }
}

// ---- nearest-neighbour 2 x upsample: output-driven, one thread = one 16-byte vector ---------------------------------------------------
__global__ __launch_bounds__(256) void upsample2x_kernel(const uint16_t* __restrict__ X, uint16_t* __restrict__ Y, int Hp, int Wp, int C) {
    const int Ho = 2 * (Hp - 2) + 2, Wo = 2 * (Wp - 2) + 2, vpr = C >> 3;
    const uint32_t total = (uint32_t)Ho * Wo * vpr;                   // < 2^31 (checked by the launcher)
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
        const int r = (int)(i / (uint32_t)vpr), v = (int)(i - (uint32_t)r * vpr), oy = r / Wo, ox = r - oy * Wo;
        uint4 o = make_uint4(0u, 0u, 0u, 0u);
        if (ox != 0 && ox != Wo - 1 && oy != 0 && oy != Ho - 1)
            o = *(const uint4*)(X + ((size_t)(((oy - 1) >> 1) + 1) * Wp + ((ox - 1) >> 1) + 1) * C + v * 8);
        *(uint4*)(Y + (size_t)r * C + v * 8) = o;
    }
}

// ---- row softmax of the mid-block attention (one head of width C over every pixel: 3 GEMMs + this pass) -----------------------------
// S [rows, ld] bf16 holds q . k for every (query pixel, key pixel) of the PADDED image; key columns on the border (and the K padding
// columns [rows, ld)) get probability 0.  In place: P = softmax(scale * S) as bf16.  One block per query row.
__global__ __launch_bounds__(256) void softmax_rows_kernel(uint16_t* __restrict__ S, int ld, int Hp, int Wp, float scale_log2e) {
    const int rows = Hp * Wp, tid = threadIdx.x;
    uint16_t* row = S + (size_t)blockIdx.x * ld;
    constexpr int MAXV = 12;                          // 256 threads x 12 vectors x 8 = 24576 columns
    uint4 w[MAXV];
    uint32_t okm[MAXV];                               // bit e: column vi * 8 + e is a valid (non-border, non-padding) key
    float mx = -INFINITY;
    const int nvec = ld >> 3;
#pragma unroll
    for (int j = 0; j < MAXV; ++j) {
        const int vi = tid + j * 256;
        if (vi >= nvec) break;
        w[j] = *(const uint4*)(row + vi * 8);
        const uint32_t ws[4] = {w[j].x, w[j].y, w[j].z, w[j].w};
        int cy = (vi * 8) / Wp, cx = vi * 8 - cy * Wp;            // one division per vector, then stepped
        uint32_t m = 0;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const bool ok = vi * 8 + e < rows && cx != 0 && cx != Wp - 1 && cy != 0 && cy != Hp - 1;
            const float x = (e & 1) ? hi_bf(ws[e >> 1]) : lo_bf(ws[e >> 1]);
            if (ok) { mx = fmaxf(mx, x); m |= 1u << e; }
            if (++cx == Wp) { cx = 0; ++cy; }
        }
        okm[j] = m;
    }
    __shared__ float red[4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if ((tid & 63) == 0) red[tid >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float sum = 0.f;
    float p[MAXV][8];
#pragma unroll
    for (int j = 0; j < MAXV; ++j) {
        const int vi = tid + j * 256;
        if (vi >= nvec) break;
        const uint32_t ws[4] = {w[j].x, w[j].y, w[j].z, w[j].w};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float x = (e & 1) ? hi_bf(ws[e >> 1]) : lo_bf(ws[e >> 1]);
            p[j][e] = ((okm[j] >> e) & 1u) ? __builtin_amdgcn_exp2f((x - mx) * scale_log2e) : 0.f;
            sum += p[j][e];
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    if ((tid & 63) == 0) red[tid >> 6] = sum;
    __syncthreads();
    sum = (red[0] + red[1]) + (red[2] + red[3]);
    const float inv = sum > 0.f ? 1.0f / sum : 0.f;
#pragma unroll
    for (int j = 0; j < MAXV; ++j) {
        const int vi = tid + j * 256;
        if (vi >= nvec) break;
        *(uint4*)(row + vi * 8) = make_uint4(pk_bf16(p[j][0] * inv, p[j][1] * inv), pk_bf16(p[j][2] * inv, p[j][3] * inv),
                                             pk_bf16(p[j][4] * inv, p[j][5] * inv), pk_bf16(p[j][6] * inv, p[j][7] * inv));
    }
}

// ---- host layouts ---------------------------------------------------------------------------------------------------------------
// z [Cz, H, W] (bf16, NCHW of one image) -> padded pixel-major [Hp * Wp, Cpad] with channels [Cz, Cpad) and the border zero
__global__ __launch_bounds__(256) void nchw_to_padded_kernel(const uint16_t* __restrict__ Z, uint16_t* __restrict__ Y, int Cz, int H, int W, int Cpad) {
    const int Hp = H + 2, Wp = W + 2;
    const uint32_t total = (uint32_t)Hp * Wp * Cpad;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
        const int r = (int)(i / (uint32_t)Cpad), c = (int)(i - (uint32_t)r * Cpad), py = r / Wp, px = r - py * Wp;
        uint16_t o = 0;
        if (c < Cz && px != 0 && px != Wp - 1 && py != 0 && py != Hp - 1) o = Z[((size_t)c * H + (py - 1)) * W + (px - 1)];
        Y[i] = o;
    }
}

// padded pixel-major [Hp * Wp, ld] (first Co channels) -> [Co, H, W] bf16
__global__ __launch_bounds__(256) void padded_to_nchw_kernel(const uint16_t* __restrict__ X, int ld, uint16_t* __restrict__ O, int Co, int H, int W) {
    const int Wp = W + 2;
    const uint32_t total = (uint32_t)Co * H * W;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
        const int t = (int)(i / (uint32_t)W), x = (int)(i - (uint32_t)t * W), c = t / H, y = t - c * H;
        O[i] = X[((size_t)(y + 1) * Wp + (x + 1)) * ld + c];
    }
}

}  // namespace rgn

using namespace rgn;

extern "C" {

static inline int grid_for(size_t work_items, int per_block, int cap) {
    const size_t g = (work_items + per_block - 1) / per_block;
    return (int)(g < 1 ? 1 : (g > (size_t)cap ? (size_t)cap : g));
}

// workspace = [GN_MAX_BLOCKS][64] partial sums (the statistics kernel's blocks, or the tiles of the convolution that produced X) + 64 stats
constexpr int GN_MAX_BLOCKS = 32768;
size_t rgn_groupnorm_partial_bytes(void) { return (size_t)GN_MAX_BLOCKS * 64 * sizeof(float); }
size_t rgn_groupnorm_workspace_bytes(void) { return rgn_groupnorm_partial_bytes() + 64 * sizeof(float); }

int rgn_groupnorm_silu(const void* X, void* Y, int Hp, int Wp, int C, const void* gamma, const void* beta, float eps, int silu,
                       void* workspace, int precomputed_blocks, void* stream) {
    if (!X || !Y || !gamma || !beta || !workspace || Hp < 3 || Wp < 3 || precomputed_blocks < 0 || precomputed_blocks > GN_MAX_BLOCKS)
        return fail(RGN_E_BADARG, "groupnorm: bad argument");
    if (C != 128 && C != 256 && C != 512) return fail(RGN_E_UNSUPPORTED, "groupnorm: 32 groups over 128 / 256 / 512 channels");
    if ((((uintptr_t)X | (uintptr_t)Y | (uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)workspace) & 15) != 0)
        return fail(RGN_E_UNSUPPORTED, "groupnorm: pointers must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const int rows = Hp * Wp, rpb = 256 / (C / 8);
    float* partial = (float*)workspace;
    float* stats = partial + (size_t)GN_MAX_BLOCKS * 64;
    int nb = precomputed_blocks;
    if (nb == 0) {           // else: the convolution that wrote X left its tiles' sums in `partial` (rgn_conv_bf16 gn_partial)
        nb = grid_for((size_t)rows, rpb * 8, 1024);
        hipLaunchKernelGGL(gn_stats_kernel, dim3(nb), dim3(256), 0, st, (const uint16_t*)X, rows, C, partial);
    }
    const double count = (double)(Hp - 2) * (Wp - 2) * (C / 32);
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(1), dim3(1024), 0, st, partial, nb, count, eps, stats);
    const int nb2 = grid_for((size_t)rows, rpb * 4, 4096);
    hipLaunchKernelGGL(gn_apply_kernel, dim3(nb2), dim3(256), 0, st, (const uint16_t*)X, (uint16_t*)Y, Hp, Wp, C, stats,
                       (const uint16_t*)gamma, (const uint16_t*)beta, silu);
    return check_launch("groupnorm kernels");
}

int rgn_upsample2x(const void* X, void* Y, int Hp, int Wp, int C, void* stream) {
    if (!X || !Y || Hp < 3 || Wp < 3 || C <= 0 || (C % 8)) return fail(RGN_E_BADARG, "upsample2x: bad argument");
    const size_t total = (size_t)(2 * Hp - 2) * (2 * Wp - 2) * (C / 8);
    if (total >= ((size_t)1 << 31)) return fail(RGN_E_UNSUPPORTED, "upsample2x: image too large");
    hipLaunchKernelGGL(upsample2x_kernel, dim3(grid_for(total, 256 * 4, 16384)), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)X,
                       (uint16_t*)Y, Hp, Wp, C);
    return check_launch("upsample2x_kernel");
}

int rgn_softmax_rows(void* S, int ld, int Hp, int Wp, float scale, void* stream) {
    if (!S || Hp < 3 || Wp < 3 || (ld % 8) || ld < Hp * Wp) return fail(RGN_E_BADARG, "softmax_rows: bad argument");
    if (ld > 256 * 12 * 8) return fail(RGN_E_UNSUPPORTED, "softmax_rows: at most 24576 columns");
    hipLaunchKernelGGL(softmax_rows_kernel, dim3(Hp * Wp), dim3(256), 0, (hipStream_t)stream, (uint16_t*)S, ld, Hp, Wp,
                       scale * 1.4426950408889634f);
    return check_launch("softmax_rows_kernel");
}

int rgn_nchw_to_padded(const void* Z, void* Y, int Cz, int H, int W, int Cpad, void* stream) {
    if (!Z || !Y || Cz <= 0 || H <= 0 || W <= 0 || Cpad < Cz) return fail(RGN_E_BADARG, "nchw_to_padded: bad argument");
    const size_t total = (size_t)(H + 2) * (W + 2) * Cpad;
    hipLaunchKernelGGL(nchw_to_padded_kernel, dim3(grid_for(total, 256 * 4, 8192)), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)Z,
                       (uint16_t*)Y, Cz, H, W, Cpad);
    return check_launch("nchw_to_padded_kernel");
}

int rgn_padded_to_nchw(const void* X, int ld, void* O, int Co, int H, int W, void* stream) {
    if (!X || !O || Co <= 0 || H <= 0 || W <= 0 || ld < Co) return fail(RGN_E_BADARG, "padded_to_nchw: bad argument");
    const size_t total = (size_t)Co * H * W;
    hipLaunchKernelGGL(padded_to_nchw_kernel, dim3(grid_for(total, 256 * 4, 8192)), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)X,
                       ld, (uint16_t*)O, Co, H, W);
    return check_launch("padded_to_nchw_kernel");
}

}  // extern "C"
