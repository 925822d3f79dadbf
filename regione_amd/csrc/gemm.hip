// bf16 MFMA GEMM for gfx950:  C = epilogue(A[M,K] @ W[N,K]^T + bias), fp32 accumulate.
//
// Both operands are K-contiguous ("NT"), which is exactly what an MFMA 16x16x32 fragment wants:
// every lane reads 8 consecutive k (16 B) of one row.  Structure (cdna_hip_programming.md section 5,
// "step 3" + T1/T2):
//   * 128x128x64 tile, 256 threads = 4 waves (2x2), each wave 64x64 = 4x4 MFMA 16x16x32 tiles;
//   * global -> LDS by `global_load_lds_dwordx4` (1 KiB per wave-instruction, no VGPR round trip),
//     double buffered, one barrier per K-step;
//   * LDS tile is [128 rows][64 k] bf16 (128 B rows).  The DMA destination is lane-linear, so the
//     bank-conflict swizzle (16-B slot ^= row&7) is applied to the per-lane SOURCE address and to
//     the ds_read_b128 address (rule 21: both sides or neither);
//   * XCD-aware, bijective blockIdx -> tile map with GROUP_M ordering so that the tiles resident on
//     one XCD share A / W panels in that XCD's private L2;
//   * epilogue: accumulators + bias -> bf16 -> LDS C tile -> 16-byte row-contiguous stores with
//     GELU-tanh / gated residual / row scatter fused in.
//
// Replaces: nn.Linear calls of the [EXT] MMDiT blocks and, with out_rows, the Triton index-scatter
// GEMM `_partially_linear` (RegionE/FluxKontext/fused_kernels.py:9-101).  Unlike the Triton kernel
// the result is rounded ONCE to the cache dtype (no fp16 round trip, quirk A-3).
#include "common.h"
#include <stdlib.h>
#include <math.h>
#include <utility>
#include "gemm_loop_asm.inc"

namespace rgn {

typedef __attribute__((ext_vector_type(8))) __bf16 bf8_t;
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

// RGN_EPI_QKV: what the separate qk_norm_rope / v_transpose_store kernels do, applied to the staged C tile
struct QkvEpi {
    const uint16_t* wq;          // RMSNorm weights [128] of this problem's stream
    const uint16_t* wk;
    const float* cos_q;          // [rows][128] fp32 tables: q rows are indexed by the sequence row, k rows by kv row
    const float* sin_q;
    const float* cos_k;
    const float* sin_k;
    const int64_t* kv_rows;      // sequence row -> K/V cache row (nullptr = identity)
    uint16_t* k_slab;            // [kv rows][heads * 128]
    uint16_t* vt_slab;           // [heads * 128][skv_pad], kv index permuted inside 16-groups
    int row_base;                // sequence row of this problem's row 0
    int skv_pad, k_col, v_col, q_col, hd;     // hd = heads * 128
    int fp16_roundtrip;          // K / V columns round fp32 -> fp16 -> bf16 (the reference's partial-update kernel)
    float eps;
};

struct GemmArgs {
    QkvEpi qkv;
    const uint16_t* A;
    const uint16_t* W;
    const uint16_t* bias;
    uint16_t* C;
    const uint16_t* gate;
    const uint16_t* resid;
    const int64_t* out_rows;
    int lda, ldw, ldc;
    int M, N, K;
    int gelu_from_col;
    const float* wscale;         // per-output-channel scale, applied to the fp32 accumulator (fp8 weights); else nullptr
    int w8;                      // W is OCP e4m3fn bytes (ldw in bytes); 0 = bf16 (also: fp8 weights widened into the workspace)
};

constexpr int BK = 64;

__device__ __forceinline__ float gelu_tanh(float x) {
    // torch GELU(approximate='tanh'): 0.5*x*(1+tanh(u)), u = sqrt(2/pi)*(x+0.044715 x^3)
    //   = x * sigmoid(2u) = x / (1 + 2^(-2u*log2(e)))      (one v_exp_f32 + one v_rcp_f32)
    const float kBeta = 0.7978845608028654f, kKappa = 0.044715f;
    const float u = kBeta * (x + kKappa * x * x * x);
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-2.8853900817779268f * u));
}

typedef __attribute__((ext_vector_type(2))) __bf16 bf2_t;
typedef __attribute__((ext_vector_type(2))) float f32x2;
__device__ __forceinline__ uint32_t cvt_pk_bf16(float a, float b) {
    bf2_t r = __builtin_convertvector(f32x2{a, b}, bf2_t);     // v_cvt_pk_bf16_f32 (RNE)
    return *(uint32_t*)&r;
}

__device__ __forceinline__ size_t kvpos(size_t r) { return (r & ~(size_t)12) | ((r & 4) << 1) | ((r & 8) >> 1); }

// value of lane (l ^ X) for X in {1, 2, 4, 8}, exchanged inside the 16-lane DPP row
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xF, 0xF, true));
}
template <int X>
__device__ __forceinline__ float lane_xor(float x) {
    if constexpr (X == 1) return dpp_mov<0xB1>(x);                      // quad_perm:[1,0,3,2]
    else if constexpr (X == 2) return dpp_mov<0x4E>(x);                 // quad_perm:[2,3,0,1]
    else if constexpr (X == 4) return dpp_mov<0x1B>(dpp_mov<0x141>(x)); // row_half_mirror (l ^ 7) then quad_perm:[3,2,1,0] (l ^ 3)
    else return dpp_mov<0x128>(x);                                      // row_ror:8 == l ^ 8 inside a 16-lane row
}

// Per-head RMSNorm + RoPE of 8 consecutive columns of one row (16 lanes = one 128-wide head).  Same
// arithmetic and the SAME summation tree as qk_norm_rope_kernel (norm.hip), so both paths agree bit for bit.
struct RopeTab { float4 c0, c1, s0, s1; };

__device__ __forceinline__ RopeTab load_rope(const float* __restrict__ cos8, const float* __restrict__ sin8) {
    RopeTab t;
    t.c0 = *(const float4*)cos8; t.c1 = *(const float4*)(cos8 + 4);
    t.s0 = *(const float4*)sin8; t.s1 = *(const float4*)(sin8 + 4);
    return t;
}

__device__ __forceinline__ void qk_norm_rope_vec(uint16_t (&v)[8], const uint16_t (&wv)[8], float eps, const RopeTab& t) {
#pragma clang fp contract(off)
    float x[8], p[4];
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = bf2f(v[e]);
#pragma unroll
    for (int j = 0; j < 4; ++j) p[j] = x[2 * j] * x[2 * j] + x[2 * j + 1] * x[2 * j + 1];
    // butterfly over the 16 lanes of a head (partners lane ^ 8, ^ 4, ^ 2, ^ 1) as DPP moves inside the 16-lane row -
    // the generic __shfl_xor goes through the LDS crossbar (ds_bpermute): 256 of them per thread and tile
#pragma unroll
    for (int j = 0; j < 4; ++j) p[j] += lane_xor<8>(p[j]);
#pragma unroll
    for (int j = 0; j < 4; ++j) p[j] += lane_xor<4>(p[j]);
#pragma unroll
    for (int j = 0; j < 4; ++j) p[j] += lane_xor<2>(p[j]);
#pragma unroll
    for (int j = 0; j < 4; ++j) p[j] += lane_xor<1>(p[j]);
    const float ss = (p[0] + p[2]) + (p[1] + p[3]);
    const float r = 1.0f / sqrtf(ss * (1.0f / 128.0f) + eps);
    const float cc[8] = {t.c0.x, t.c0.y, t.c0.z, t.c0.w, t.c1.x, t.c1.y, t.c1.z, t.c1.w};
    const float sn[8] = {t.s0.x, t.s0.y, t.s0.z, t.s0.w, t.s1.x, t.s1.y, t.s1.z, t.s1.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float a = rbf(rbf(x[2 * j] * r) * bf2f(wv[2 * j]));
        const float b = rbf(rbf(x[2 * j + 1] * r) * bf2f(wv[2 * j + 1]));
        const float o0 = a * cc[2 * j] + (-b) * sn[2 * j];
        const float o1 = b * cc[2 * j + 1] + a * sn[2 * j + 1];
        v[2 * j] = f2bf(o0);
        v[2 * j + 1] = f2bf(o1);
    }
}

struct GemmGroup {
    GemmArgs p[2];      // up to two problems per launch (e.g. text + image stream of a double block)
    int nt0;            // tiles of problem 0; tiles >= nt0 belong to problem 1
    int nt;             // total tiles
    // round-aware launch: this launch covers tiles [tile_offset, tile_offset + nt_launch).
    // MODE 1 cuts each tile's K range into nsplit pieces (fp32 partial fragments -> ws);
    // MODE 2 sums the partials of each tile and runs the normal epilogue.
    int tile_offset, nt_launch, nsplit;
    float* ws;
    int part4w;         // MODE 2: the partials were written by the hand-scheduled 4-wave geometry (its lane-linear layout)
    int dbg;            // timing experiments only (RGN_GEMM_DBG): bit 0 = skip the epilogue (results are garbage)
};

constexpr int MODE_FULL = 0, MODE_PARTIAL = 1, MODE_REDUCE = 2;

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
// 16-byte streaming (`nt`) store of a row-major output vector.  A round of 256 tiles writes 32 MiB - the size of all eight
// L2s - at the same moment; as ordinary stores they evict the A / W tiles the next round is about to read.  Same-box A/B:
// +3...6 % on the isolated FLUX projections, +1.2...1.5 % on the whole edit.  NOT used for the K / V^T cache placement
// of the fused Q/K/V epilogue: the V^T tile is written as 16-byte pieces of different cache lines, which as streaming
// stores become partial-line writes (QKV pair 1046 -> 912 TFLOP/s); Q / K rows measured neutral.
__device__ __forceinline__ void store_nt16(uint16_t* dst, const uint16_t (&v)[8]) {
    __builtin_nontemporal_store(*(const u32x4_t*)v, (u32x4_t*)dst);
}



// ---- helpers of the hand-scheduled 4-wave variant (accumulators live in AGPRs a[0:255], tools/gen_gemm_loop.py) ----
template <int B>
__device__ __forceinline__ void agpr_read4(float (&c)[4]) {
    asm volatile("v_accvgpr_read_b32 %0, a[%4]\n\tv_accvgpr_read_b32 %1, a[%5]\n\tv_accvgpr_read_b32 %2, a[%6]\n\tv_accvgpr_read_b32 %3, a[%7]"
                 : "=v"(c[0]), "=v"(c[1]), "=v"(c[2]), "=v"(c[3])
                 : "n"(B), "n"(B + 1), "n"(B + 2), "n"(B + 3));
}
template <class F, int... Is>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Is...>) {
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_s;

// Tile configurations:
//   <128,128,2,2>: 4 waves, wave tile 64x64, 64 KiB LDS, 2 blocks/CU  - small / ragged problems
//   <256,256,2,4>: 8 waves, wave tile 128x64, 128 KiB LDS, 1 block/CU - large problems (half the
//                  global->LDS traffic and 25 % less LDS read traffic per FLOP)
//   <256,256,2,2>: 4 waves, wave tile 128x128, accumulators in 256 AGPRs, K loop = ONE hand-scheduled asm statement
//                  (gemm_loop_asm.inc); MODE_FULL only, K >= 128, operands < 4 GiB
template <int EPI, int BM, int BN, int WM, int WN, int MODE, int AV = 0>
__global__ __launch_bounds__(64 * WM * WN, (BM * BN >= 256 * 256) ? 1 : 2) void gemm_bf16_kernel(const GemmGroup gg) {
    constexpr int NW = WM * WN, NT = 64 * NW;
    constexpr int TM = BM / WM / 16, TN = BN / WN / 16;        // 16x16 MFMA tiles per wave
    constexpr bool ASM4W = (BM == 256 && BN == 256 && WM == 2 && WN == 2);
    // AV == 8: W is fp8 (OCP e4m3fn), one byte per element: its LDS tile has 64-byte rows, a 1 KiB DMA piece covers 16 rows,
    // a fragment is one ds_read_b64 (8 values) converted to bf16 in registers (v_cvt_scalef32_pk_bf16_fp8, exact); the
    // per-output-channel scale multiplies the fp32 accumulator in the epilogue.  Half the weight bytes from HBM / L2.
    constexpr bool W8 = (AV == 8);
    static_assert(!(W8 && ASM4W), "fp8 weights run on the compiler-scheduled geometries");
    constexpr int A_BYTES = BM * BK * 2, B_BYTES = W8 ? BN * BK : BN * BK * 2, STAGE = A_BYTES + B_BYTES;
    constexpr int PA = BM / 8 / NW, PB = (W8 ? BN / 16 : BN / 8) / NW;          // 1 KiB DMA pieces per wave per stage
    constexpr int CT_LD = BN + 8;                              // padded bf16 row of the C staging tile
    // C rows staged per epilogue chunk: one wave row at a time (the staging tile then fits inside the K-loop
    // stages), except for the fused Q/K/V epilogue, which stages the whole tile at once so that no wave keeps its
    // 128 accumulator registers alive while the others run the (register-hungry) norm / RoPE passes
    // ... and for the 4-wave asm variant (each wave holds 64 fragments: with per-wave-row chunks only half the waves
    // stage at a time and the epilogue costs ~9 us per tile instead of ~4; measured with RGN_GEMM_DBG=1 K sweeps)
    // (the gated-residual epilogue keeps per-wave-row chunks: its residual prefetch would need 128 registers for a whole tile)
    constexpr int NCH = (EPI == RGN_EPI_QKV || (BM == 256 && WM * WN == 4 && EPI != RGN_EPI_GATE_RESID)) ? 1 : WM;
    constexpr int CROWS = BM / NCH;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;

    // ---- XCD-aware bijective tile map (T1) over both problems + grouped ordering ------------------
    int t, split = 0, unit;
    {
        const int nb = (MODE == MODE_PARTIAL) ? gg.nt_launch * gg.nsplit : gg.nt_launch;
        const int bid = blockIdx.x, xcd = bid & 7, loc = bid >> 3;
        const int q = nb >> 3, r = nb & 7;
        unit = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
        t = unit;
        if (MODE == MODE_PARTIAL) { split = t % gg.nsplit; t /= gg.nsplit; }
        unit = t;                       // launch-local tile index (workspace slot)
        t += gg.tile_offset;
    }
    const int pi = (t >= gg.nt0) ? 1 : 0;
    const GemmArgs& g = gg.p[pi];
    t -= pi ? gg.nt0 : 0;
    const int ntm = (g.M + BM - 1) / BM, ntn = (g.N + BN - 1) / BN;
    constexpr int GROUP_M = (BM >= 256) ? 4 : 8;
    const int per_group = GROUP_M * ntn;
    const int group = t / per_group, first_m = group * GROUP_M;
    const int gsize = min(ntm - first_m, GROUP_M);
    const int tm = first_m + (t % per_group) % gsize;
    const int tn = (t % per_group) / gsize;
    const int m0 = tm * BM, n0 = tn * BN;

    // ---- staging addresses ---------------------------------------------------------------------
    const int srow = lane >> 3;                    // row inside an 8-row piece
    const int schunk = (lane & 7) ^ srow;          // source 16-B chunk for LDS slot (lane&7)
    const uint8_t* a_src[PA];
    const uint8_t* b_src[PB];
#pragma unroll
    for (int q = 0; q < PA; ++q) {
        const int ar = min(m0 + (wave * PA + q) * 8 + srow, g.M - 1);
        a_src[q] = (const uint8_t*)(g.A + (size_t)ar * g.lda) + schunk * 16;
    }
#pragma unroll
    for (int q = 0; q < PB; ++q) {
        if constexpr (W8) {
            // 16 rows x 64 bytes per piece: lane -> (row lane>>2, 16-byte chunk lane&3); bank swizzle: the 8-byte slot index
            // ^= ((row >> 2) & 3) << 1, i.e. the 16-byte chunk ^= (row >> 2) & 3 - applied on the source, like the bf16 tiles
            const int r16 = lane >> 2;
            const int br = min(n0 + (wave * PB + q) * 16 + r16, g.N - 1);
            b_src[q] = (const uint8_t*)g.W + (size_t)br * g.ldw + (((lane & 3) ^ ((r16 >> 2) & 3)) * 16);
        } else {
            const int br = min(n0 + (wave * PB + q) * 8 + srow, g.N - 1);
            b_src[q] = (const uint8_t*)(g.W + (size_t)br * g.ldw) + schunk * 16;
        }
    }
    auto stage = [&](int kt, int buf) {
        uint8_t* base = smem + buf * STAGE;
#pragma unroll
        for (int q = 0; q < PA; ++q)
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(a_src[q] + (size_t)kt * (BK * 2)),
                                             (lds_ptr_t)(base + (wave * PA + q) * 1024), 16, 0, 0);
#pragma unroll
        for (int q = 0; q < PB; ++q)
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(b_src[q] + (size_t)kt * (W8 ? BK : BK * 2)),
                                             (lds_ptr_t)(base + A_BYTES + (wave * PB + q) * 1024), 16, 0, 0);
    };

    // ---- fragment read offsets (swizzled) -------------------------------------------------------
    const int frow = lane & 15, fk = lane >> 4;
    int a_off[2], b_off[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const int slot = (kk * 4 + fk) ^ (frow & 7);
        a_off[kk] = (wm * (BM / WM) + frow) * 128 + slot * 16;
        if constexpr (W8) b_off[kk] = A_BYTES + (wn * (BN / WN) + frow) * 64 + (((kk * 4 + fk) ^ (((frow >> 2) & 3) << 1)) * 8);
        else b_off[kk] = A_BYTES + (wn * (BN / WN) + frow) * 128 + slot * 16;
    }

    f32x4 acc[TM][TN];
    if constexpr (!ASM4W) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    const int nk_all = g.K / BK;
    int k_begin = 0, nk = nk_all;
    if (MODE == MODE_PARTIAL) {
        const int per = (nk_all + gg.nsplit - 1) / gg.nsplit;
        k_begin = split * per;
        nk = max(0, min(per, nk_all - k_begin));
    }
    // 4-wave variant: the bias of this lane's 8 x 4 columns is requested BEFORE the K loop (one L2 round trip hidden under
    // the loop); fetched inside the staging loop, hipcc waits vmcnt(0) behind each of the 8 loads - 8 serialised round
    // trips, ~5 of the ~9 us this variant's epilogue used to cost
    uint2 bias_pre[ASM4W ? TN : 1];
    if constexpr (ASM4W) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + wn * (BN / WN) + j * 16 + (lane >> 4) * 4;
            bias_pre[j] = (g.bias != nullptr && col + 4 <= g.N) ? *(const uint2*)(g.bias + col) : make_uint2(0u, 0u);
        }
    }
    if constexpr (ASM4W) {
        // ---- hand-scheduled K loop (tools/gen_gemm_loop.py): per-lane SOURCE byte offsets of this wave's 8 + 8 DMA pieces
        // (same swizzled image as stage()), LDS fragment addresses of stage 0, buffer resources, loop count ----
        uint32_t oa[PA], ob[PB];
#pragma unroll
        for (int q = 0; q < PA; ++q) {
            const int ar = min(m0 + (wave * PA + q) * 8 + srow, g.M - 1);
            oa[q] = (uint32_t)ar * (uint32_t)(g.lda * 2) + schunk * 16;
        }
#pragma unroll
        for (int q = 0; q < PB; ++q) {
            const int br = min(n0 + (wave * PB + q) * 8 + srow, g.N - 1);
            ob[q] = (uint32_t)br * (uint32_t)(g.ldw * 2) + schunk * 16;
        }
        const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)smem;
        uint32_t la0 = lds0 + a_off[0], la1 = lds0 + a_off[1];
        uint32_t lb0 = lds0 + b_off[0] - A_BYTES, lb1 = lds0 + b_off[1] - A_BYTES;      // the asm adds A_BYTES as an immediate
        auto rsrc = [](const void* p) {
            const uint64_t a = (uint64_t)p;
            u32x4_s r;
            r[0] = __builtin_amdgcn_readfirstlane((uint32_t)a);
            r[1] = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32) & 0xffffu);      // stride 0: raw buffer
            r[2] = 0xffffffffu;                                                          // rows are clamped by the offsets
            r[3] = 0x00020000u;
            return r;
        };
        const u32x4_s pa = rsrc(g.A), pw = rsrc(g.W);
        uint32_t stg = __builtin_amdgcn_readfirstlane(lds0 + wave * (PA * 1024));
        uint32_t koff = __builtin_amdgcn_readfirstlane(k_begin * (BK * 2));       // byte offset of the first K tile (split-K piece)
        if constexpr (MODE == MODE_REDUCE) {
            // no K loop: the accumulators are the sums of the partial fragments, read by the epilogue below
        } else if constexpr (AV == 0) {
            uint32_t cnt = __builtin_amdgcn_readfirstlane(nk - 2);
            asm volatile(RGN_GEMM_LOOP4W_ASM
                         : [la0] "+&v"(la0), [la1] "+&v"(la1), [lb0] "+&v"(lb0), [lb1] "+&v"(lb1), [cnt] "+&s"(cnt), [stg] "+&s"(stg),
                           [koff] "+&s"(koff)
                         : [oa0] "v"(oa[0]), [oa1] "v"(oa[1]), [oa2] "v"(oa[2]), [oa3] "v"(oa[3]), [oa4] "v"(oa[4]), [oa5] "v"(oa[5]),
                           [oa6] "v"(oa[6]), [oa7] "v"(oa[7]), [ob0] "v"(ob[0]), [ob1] "v"(ob[1]), [ob2] "v"(ob[2]), [ob3] "v"(ob[3]),
                           [ob4] "v"(ob[4]), [ob5] "v"(ob[5]), [ob6] "v"(ob[6]), [ob7] "v"(ob[7]), [pa] "s"(pa), [pw] "s"(pw)
                         : RGN_GEMM_LOOP4W_CLOBBERS);
        } else {
            // ring variant: A slots at 0 / 32 K, W slots at 64 K / 96 K / 128 K (all 160 KiB), W two tiles ahead
            uint32_t cnt = __builtin_amdgcn_readfirstlane(nk - 3);
            const uint32_t lbo0 = lb0 - lds0, lbo1 = lb1 - lds0;
            const uint32_t wwrap = __builtin_amdgcn_readfirstlane(lds0 + 65536 + wave * (PB * 1024));
            uint32_t was = wwrap, wrd = __builtin_amdgcn_readfirstlane(lds0 + 65536), kofw = koff;
            asm volatile(RGN_GEMM_LOOP4W_RING_ASM
                         : [la0] "+&v"(la0), [la1] "+&v"(la1), [lb0] "=&v"(lb0), [lb1] "=&v"(lb1), [cnt] "+&s"(cnt), [stg] "+&s"(stg),
                           [koff] "+&s"(koff), [was] "+&s"(was), [wrd] "+&s"(wrd), [kofw] "+&s"(kofw)
                         : [oa0] "v"(oa[0]), [oa1] "v"(oa[1]), [oa2] "v"(oa[2]), [oa3] "v"(oa[3]), [oa4] "v"(oa[4]), [oa5] "v"(oa[5]),
                           [oa6] "v"(oa[6]), [oa7] "v"(oa[7]), [ob0] "v"(ob[0]), [ob1] "v"(ob[1]), [ob2] "v"(ob[2]), [ob3] "v"(ob[3]),
                           [ob4] "v"(ob[4]), [ob5] "v"(ob[5]), [ob6] "v"(ob[6]), [ob7] "v"(ob[7]), [pa] "s"(pa), [pw] "s"(pw),
                           [lbo0] "v"(lbo0), [lbo1] "v"(lbo1), [wwrap] "s"(wwrap)
                         : RGN_GEMM_LOOP4W_CLOBBERS);
        }
    }
    if constexpr (!ASM4W) {
    if (MODE != MODE_REDUCE && nk > 0) {
        stage(k_begin, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            const int cur = kt & 1;
            if (kt + 1 < nk) stage(k_begin + kt + 1, cur ^ 1);
            const uint8_t* sb = smem + cur * STAGE;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                bf8_t af[TM], bfr[TN];
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if constexpr (W8) {
                        const uint2 raw = *(const uint2*)(sb + b_off[kk] + j * 1024);        // 8 fp8 values of one row
                        const bf2_t c0 = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(raw.x, 1.0f, false);
                        const bf2_t c1 = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(raw.x, 1.0f, true);
                        const bf2_t c2 = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(raw.y, 1.0f, false);
                        const bf2_t c3 = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(raw.y, 1.0f, true);
                        bfr[j] = bf8_t{c0[0], c0[1], c1[0], c1[1], c2[0], c2[1], c3[0], c3[1]};
                    } else {
                        bfr[j] = *(const bf8_t*)(sb + b_off[kk] + j * 2048);
                    }
                }
#pragma unroll
                for (int i = 0; i < TM; ++i) af[i] = *(const bf8_t*)(sb + a_off[kk] + i * 2048);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);   // C^T fragment
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    }
    }
    if (gg.dbg & 1) return;
    if constexpr (MODE == MODE_PARTIAL && ASM4W) {
        float4* w = (float4*)gg.ws + ((size_t)unit * gg.nsplit + split) * (size_t)(TM * TN * NT) + tid;
        static_for<TM * TN>([&](auto Fc) {
            constexpr int F = decltype(Fc)::value;
            float c[4];
            agpr_read4<F * 4>(c);
            w[(size_t)F * NT] = make_float4(c[0], c[1], c[2], c[3]);
        });
        return;
    }
    if (MODE == MODE_PARTIAL) {
        // lane-linear fp32 fragment dump: [unit][split][fragment (i,j)][thread] float4 - fully coalesced
        float4* w = (float4*)gg.ws + ((size_t)unit * gg.nsplit + split) * (size_t)(TM * TN * NT) + tid;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                w[(size_t)(i * TN + j) * NT] = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
        return;
    }
    if (MODE == MODE_REDUCE && !ASM4W) {
        if (BM == 256 && WN == 4 && gg.part4w) {
            // the pieces ran on the hand-scheduled 4-wave geometry (wave tile 128 x 128, 64 fragments x 256 threads per tile):
            // this wave's 128 x 64 tile is the left / right half of 4-wave wave (wm, wn / 2)'s - same lane layout per fragment
            const float4* w = (const float4*)gg.ws + (size_t)unit * gg.nsplit * (size_t)(64 * 256) + (wm * 2 + (wn >> 1)) * 64 + lane;
            for (int sp = 0; sp < gg.nsplit; ++sp) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const float4 v = w[(size_t)(sp * 64 + i * 8 + (wn & 1) * 4 + j) * 256];
                        acc[i][j][0] += v.x; acc[i][j][1] += v.y; acc[i][j][2] += v.z; acc[i][j][3] += v.w;
                    }
            }
        } else {
        const float4* w = (const float4*)gg.ws + (size_t)unit * gg.nsplit * (size_t)(TM * TN * NT) + tid;
        for (int sp = 0; sp < gg.nsplit; ++sp) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const float4 v = w[(size_t)(sp * TM * TN + i * TN + j) * NT];
                    acc[i][j][0] += v.x; acc[i][j][1] += v.y; acc[i][j][2] += v.z; acc[i][j][3] += v.w;
                }
        }
        }
    }

    // ---- epilogue: per row-chunk (one wave row): acc + bias -> bf16 -> LDS -> 16-byte stores ------
    uint16_t* ct = (uint16_t*)smem;                 // CROWS x CT_LD bf16 (stages are free now)
    int* krow = (int*)(smem + CROWS * CT_LD * 2);   // CROWS cache-row indices (RGN_EPI_QKV)
    constexpr int VPR = BN / 8;                     // 8-column vectors per tile row
    constexpr int RPP = NT / VPR;                   // rows per store pass
    const int vc = tid % VPR;
    const int ncol = n0 + vc * 8;
    const bool full_vec = (ncol + 8 <= g.N);
    uint16_t gv[8];
    if (EPI == RGN_EPI_GATE_RESID) {
#pragma unroll
        for (int e = 0; e < 8; ++e) gv[e] = (ncol + e < g.N) ? g.gate[ncol + e] : 0;
    }
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        // gated residual: the residual rows of this chunk are requested BEFORE the accumulators are staged, so the
        // global latency hides under the staging pass instead of sitting in front of every store
        constexpr int NITG = CROWS / RPP;
        uint4 rpre[(EPI == RGN_EPI_GATE_RESID) ? NITG : 1];
        const bool pre_ok = (EPI == RGN_EPI_GATE_RESID) && full_vec && g.out_rows == nullptr;
        if (EPI == RGN_EPI_GATE_RESID && pre_ok) {
#pragma unroll
            for (int it = 0; it < NITG; ++it) {
                const int m = min(m0 + ch * CROWS + tid / VPR + it * RPP, g.M - 1);
                rpre[it] = *(const uint4*)(g.resid + (size_t)m * g.ldc + ncol);
            }
        }
        if (NCH == 1 || wm == ch) {
            const int crow0 = (NCH == 1) ? wm * (BM / WM) : 0;
            // operands are fed swapped (W fragment as the MFMA row operand), so a lane's 4 accumulator
            // registers are 4 CONSECUTIVE COLUMNS of one output row: one 8-byte LDS write per fragment
            const int mrow = lane & 15, ncol4 = (lane >> 4) * 4;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int nl = wn * (BN / WN) + j * 16 + ncol4;
                float bv[4] = {0.f, 0.f, 0.f, 0.f};
                if (g.bias != nullptr) {
                    if (n0 + nl + 4 <= g.N) {
                        uint2 b2;
                        if constexpr (ASM4W) b2 = bias_pre[j];
                        else b2 = *(const uint2*)(g.bias + n0 + nl);
                        bv[0] = bf2f(b2.x & 0xffff); bv[1] = bf2f(b2.x >> 16); bv[2] = bf2f(b2.y & 0xffff); bv[3] = bf2f(b2.y >> 16);
                    } else {
                        for (int r = 0; r < 4; ++r) if (n0 + nl + r < g.N) bv[r] = bf2f(g.bias[n0 + nl + r]);
                    }
                }
                // the reference's partial K/V update (_triton_matmul_index_kernel) stores `accumulator.to(tl.float16)` into
                // the bf16 cache (fused_kernels.py:80, quirk A-3): K / V columns of a partial-update problem take the same
                // fp32 -> fp16 -> bf16 double rounding; every other column rounds once, like F.linear
                const bool f16rt = (EPI == RGN_EPI_QKV) && g.qkv.fp16_roundtrip && n0 < g.qkv.q_col;
                float sv[4] = {1.f, 1.f, 1.f, 1.f};
                if (g.wscale != nullptr) {               // fp8 weights: per-output-channel scale on the fp32 accumulator
#pragma unroll
                    for (int r = 0; r < 4; ++r) sv[r] = (n0 + nl + r < g.N) ? g.wscale[n0 + nl + r] : 1.f;
                }
                auto put = [&](int i, float (&c)[4]) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) c[r] = c[r] * sv[r] + bv[r];
                    if (f16rt) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) c[r] = (float)(_Float16)c[r];
                    }
                    const uint2 w = make_uint2(cvt_pk_bf16(c[0], c[1]), cvt_pk_bf16(c[2], c[3]));
                    *(uint2*)(ct + (crow0 + i * 16 + mrow) * CT_LD + nl) = w;
                };
                if constexpr (ASM4W) {
                    // j is a runtime index of the (fully unrolled) outer loop: AGPR numbers must be literal -> dispatch
                    static_for<TN>([&](auto Jc) {
                        constexpr int J = decltype(Jc)::value;
                        if (j == J) {
                            static_for<TM>([&](auto Ic) {
                                constexpr int I = decltype(Ic)::value;
                                float c[4];
                                agpr_read4<(I * TN + J) * 4>(c);
                                put(I, c);
                            });
                        }
                    });
                } else {
#pragma unroll
                    for (int i = 0; i < TM; ++i) {
                        float c[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) c[r] = acc[i][j][r];
                        put(i, c);
                    }
                }
            }
        }
        if (EPI == RGN_EPI_QKV && tid < CROWS) {
            // cache rows of this chunk's sequence rows: one coalesced load per chunk instead of a dependent
            // global load in front of every K / V^T store
            const int ml = m0 + ch * CROWS + tid;
            const int sr = g.qkv.row_base + ml;
            krow[tid] = (ml < g.M) ? (g.qkv.kv_rows ? (int)g.qkv.kv_rows[sr] : sr) : -1;
        }
        __syncthreads();
        if (EPI == RGN_EPI_QKV && n0 >= g.qkv.v_col && n0 < g.qkv.v_col + g.qkv.hd) {
            // ---- V tile: transposed into the V^T slab (kv index bits 2<->3 swapped inside 16-groups) ----
            // item = (column d, 16-row group a, half p): rows 16a + {4p..4p+3, 8+4p..8+4p+3} are 8 CONSECUTIVE
            // slab positions 16a + 8p .. +7  ->  one 16-byte store when the kv rows are the identity
            const QkvEpi& q = g.qkv;
            const int rbase = q.row_base + m0 + ch * CROWS;                 // sequence row of chunk row 0
            const bool ident = (q.kv_rows == nullptr) && ((rbase & 15) == 0) && (m0 + (ch + 1) * CROWS <= g.M);
            if (ident) {
                constexpr int ITEMS = BN * (CROWS / 8) / NT;
#pragma unroll
                for (int j = 0; j < ITEMS; ++j) {
                    const int item = tid + j * NT;
                    const int col = item % BN, grp = item / BN;
                    const int a = grp >> 1, ph = grp & 1;
                    const int hd_col = n0 - q.v_col + col;                  // h * 128 + d
                    if (hd_col >= q.hd) continue;
                    uint16_t t[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) t[e] = ct[(16 * a + 4 * ph + (e & 3) + 8 * (e >> 2)) * CT_LD + col];
                    *(uint4*)(q.vt_slab + (size_t)hd_col * q.skv_pad + rbase + 16 * a + 8 * ph) = *(const uint4*)t;
                }
            } else {
                // gathered cache rows (region step) or a ragged chunk: lanes run over ROWS, so one store instruction
                // writes 64 kv positions of one (h, d) row of V^T - contiguous wherever the cache rows are (text rows,
                // runs of edited tokens) - instead of 64 different rows
                static_assert(NT % CROWS == 0 || CROWS % NT == 0, "row-major V scatter mapping");
                constexpr int RPT = (CROWS > NT) ? CROWS / NT : 1;          // rows per thread
                constexpr int CSTEP = (NT > CROWS) ? NT / CROWS : 1;        // threads sharing a row step through the columns
#pragma unroll
                for (int rr = 0; rr < RPT; ++rr) {
                    const int row = (tid % CROWS) + rr * NT;
                    const int kr = krow[row];
                    if (kr < 0) continue;
                    uint16_t* dcol = q.vt_slab + kvpos((size_t)kr);
                    const uint16_t* src = ct + row * CT_LD;
                    for (int col = tid / CROWS; col < BN; col += CSTEP) {
                        const int hd_col = n0 - q.v_col + col;
                        if (hd_col < q.hd) dcol[(size_t)hd_col * q.skv_pad] = src[col];
                    }
                }
            }
            if (ch + 1 < NCH) __syncthreads();
            continue;
        }
        const bool qk_tile = (EPI == RGN_EPI_QKV) && n0 < g.qkv.q_col + g.qkv.hd &&
                             (n0 >= g.qkv.q_col || (n0 >= g.qkv.k_col && n0 < g.qkv.k_col + g.qkv.hd));
        if (EPI == RGN_EPI_QKV && qk_tile) {
            // ---- Q / K tile: per-head RMSNorm + RoPE; Q goes back to C in place, K to its cache row.  The
            // cos / sin rows of the next two passes are in flight while this pass computes. ----
            const QkvEpi& q = g.qkv;
            const bool k_tile = !(n0 >= q.q_col);
            const int hc = ncol - (k_tile ? q.k_col : q.q_col);            // h * 128 + c, c = 8 * (lane in head)
            const float* cos_t = (k_tile ? q.cos_k : q.cos_q) + (hc & 127);
            const float* sin_t = (k_tile ? q.sin_k : q.sin_q) + (hc & 127);
            uint16_t wv[8];
            *(uint4*)wv = *(const uint4*)((k_tile ? q.wk : q.wq) + (hc & 127));
            constexpr int NIT = CROWS / RPP, DEPTH = 2;
            auto tab_row = [&](int it) -> size_t {                          // clamped: rows >= M load row 0, never stored
                const int row = tid / VPR + it * RPP;
                const int kr = krow[row];
                return kr < 0 ? 0 : (k_tile ? (size_t)kr : (size_t)(q.row_base + m0 + ch * CROWS + row));
            };
            RopeTab tab[DEPTH];
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                const size_t tr = tab_row(d);
                tab[d] = load_rope(cos_t + tr * 128, sin_t + tr * 128);
            }
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int row = tid / VPR + it * RPP;
                const int m = m0 + ch * CROWS + row;
                uint16_t v[8];
                *(uint4*)v = *(const uint4*)(ct + row * CT_LD + vc * 8);
                const RopeTab t = tab[it % DEPTH];
                if (it + DEPTH < NIT) {
                    const size_t tr = tab_row(it + DEPTH);
                    tab[it % DEPTH] = load_rope(cos_t + tr * 128, sin_t + tr * 128);
                }
                qk_norm_rope_vec(v, wv, q.eps, t);
                if (m < g.M) {
                    uint16_t* dst = k_tile ? q.k_slab + (size_t)krow[row] * q.hd + hc : g.C + (size_t)m * g.ldc + ncol;
                    *(uint4*)dst = *(const uint4*)v;
                }
            }
            if (ch + 1 < NCH) __syncthreads();
            continue;
        }
#pragma unroll
        for (int it = 0; it < CROWS / RPP; ++it) {
            const int row = tid / VPR + it * RPP;
            const int m = m0 + ch * CROWS + row;
            if (m >= g.M || ncol >= g.N) continue;
            const size_t orow = g.out_rows ? (size_t)g.out_rows[m] : (size_t)m;
            uint16_t v[8];
            *(uint4*)v = *(const uint4*)(ct + row * CT_LD + vc * 8);
            uint16_t* dst = g.C + orow * g.ldc + ncol;
            if (EPI == RGN_EPI_GELU || EPI == RGN_EPI_QKV) {          // QKV: the fused MLP half of a single-stream block
                // one decision per 8-column vector (gelu_from_col is a multiple of 8) and all 8 lanes of math
                // independent: a per-element branch serialises eight 12-deep dependency chains (exp, rcp) per vector
                if (ncol >= g.gelu_from_col) {
                    float y[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) y[e] = gelu_tanh(bf2f(v[e]));
                    uint32_t* pv = (uint32_t*)v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) pv[e] = cvt_pk_bf16(y[2 * e], y[2 * e + 1]);
                }
            } else if (EPI == RGN_EPI_GATE_RESID) {
                const uint16_t* rs = g.resid + orow * g.ldc + ncol;
                uint16_t rv[8];
                if (pre_ok) *(uint4*)rv = rpre[it];
                else if (full_vec) *(uint4*)rv = *(const uint4*)rs;
                else
                    for (int e = 0; e < 8; ++e) rv[e] = (ncol + e < g.N) ? rs[e] : 0;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float gated = rbf(bf2f(gv[e]) * bf2f(v[e]));      // gate.unsqueeze(1) * out (bf16)
                    v[e] = f2bf(bf2f(rv[e]) + gated);                         // residual + ... (bf16)
                }
            }
            if (full_vec) store_nt16(dst, v);
            else
                for (int e = 0; e < 8; ++e)
                    if (ncol + e < g.N) dst[e] = v[e];
        }
        if (ch + 1 < NCH) __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// Skinny GEMV: one wave per output row n, lanes stride K in 16-byte vectors.  HBM-bound on W.
// ------------------------------------------------------------------------------------------------
template <int B>
__global__ __launch_bounds__(256) void gemv_bf16_kernel(const uint16_t* __restrict__ x, int ldx,
                                                        const uint16_t* __restrict__ W,
                                                        const uint16_t* __restrict__ bias,
                                                        uint16_t* __restrict__ y, int ldy, int N, int K,
                                                        int silu_input) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    float acc[B];
#pragma unroll
    for (int b = 0; b < B; ++b) acc[b] = 0.f;
    const uint16_t* wr = W + (size_t)n * K;
    for (int k = lane * 8; k < K; k += 64 * 8) {
        uint16_t wv[8];
        *(uint4*)wv = *(const uint4*)(wr + k);
#pragma unroll
        for (int b = 0; b < B; ++b) {
            uint16_t xv[8];
            *(uint4*)xv = *(const uint4*)(x + (size_t)b * ldx + k);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float xf = bf2f(xv[e]);
                if (silu_input) xf = rbf(xf / (1.0f + expf(-xf)));     // F.silu in bf16: fp32 math, bf16 out
                acc[b] += xf * bf2f(wv[e]);
            }
        }
    }
#pragma unroll
    for (int b = 0; b < B; ++b) {
        float s = wave_sum(acc[b]);
        if (lane == 0) y[(size_t)b * ldy + n] = f2bf(s + (bias ? bf2f(bias[n]) : 0.f));
    }
}

}  // namespace rgn

using namespace rgn;

template <int BM, int BN, int WM, int WN, int MODE, int AV = 0>
static int launch_gemm(const GemmGroup& gg, int epilogue, hipStream_t st) {
    constexpr int QKV_TILE = BM * (BN + 8) * 2 + BM * 4;            // whole staged C tile + cache-row table
    constexpr bool ASM4W = (BM == 256 && WM * WN == 4);
    constexpr int LDS_LOOP = (AV == 1) ? 160 * 1024 : (AV == 8) ? 2 * (BM * BK * 2 + BN * BK) : 2 * (BM + BN) * BK * 2;
    constexpr int LDS = (ASM4W && QKV_TILE > LDS_LOOP) ? QKV_TILE : LDS_LOOP;     // the asm variant stages the whole C tile
    constexpr int LDS_QKV = QKV_TILE > LDS ? QKV_TILE : LDS;
    constexpr int NT = 64 * WM * WN;
    // the opt-in is per device (one process may drive several GPUs): remember which devices have it
    static bool attr[64] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64 || !attr[dev]) {
        (void)hipFuncSetAttribute((const void*)gemm_bf16_kernel<RGN_EPI_BIAS, BM, BN, WM, WN, MODE, AV>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        (void)hipFuncSetAttribute((const void*)gemm_bf16_kernel<RGN_EPI_GELU, BM, BN, WM, WN, MODE, AV>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        (void)hipFuncSetAttribute((const void*)gemm_bf16_kernel<RGN_EPI_GATE_RESID, BM, BN, WM, WN, MODE, AV>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        (void)hipFuncSetAttribute((const void*)gemm_bf16_kernel<RGN_EPI_QKV, BM, BN, WM, WN, MODE, AV>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_QKV);
        if (dev >= 0 && dev < 64) attr[dev] = true;
    }
    const int nb = (MODE == MODE_PARTIAL) ? gg.nt_launch * gg.nsplit : gg.nt_launch;
    if (nb == 0) return 0;
    if (MODE == MODE_PARTIAL) {       // partial fragments carry no epilogue
        hipLaunchKernelGGL((gemm_bf16_kernel<RGN_EPI_BIAS, BM, BN, WM, WN, MODE, AV>), dim3(nb), dim3(NT), LDS, st, gg);
        return check_launch("gemm_bf16_kernel(partial)");
    }
    switch (epilogue) {
        case RGN_EPI_BIAS: hipLaunchKernelGGL((gemm_bf16_kernel<RGN_EPI_BIAS, BM, BN, WM, WN, MODE, AV>), dim3(nb), dim3(NT), LDS, st, gg); break;
        case RGN_EPI_GELU: hipLaunchKernelGGL((gemm_bf16_kernel<RGN_EPI_GELU, BM, BN, WM, WN, MODE, AV>), dim3(nb), dim3(NT), LDS, st, gg); break;
        case RGN_EPI_GATE_RESID: hipLaunchKernelGGL((gemm_bf16_kernel<RGN_EPI_GATE_RESID, BM, BN, WM, WN, MODE, AV>), dim3(nb), dim3(NT), LDS, st, gg); break;
        case RGN_EPI_QKV: hipLaunchKernelGGL((gemm_bf16_kernel<RGN_EPI_QKV, BM, BN, WM, WN, MODE, AV>), dim3(nb), dim3(NT), LDS_QKV, st, gg); break;
        default: return fail(RGN_E_BADARG, "gemm: unknown epilogue");
    }
    return check_launch("gemm_bf16_kernel");
}

static int check_problem(const void* A, int lda, const void* W, int ldw, const void* C, int ldc, int M, int N, int K,
                         int epilogue, const void* gate, const void* resid) {
    if (!A || !W || !C || M < 0 || N <= 0 || K <= 0) return fail(RGN_E_BADARG, "gemm: bad argument");
    if (K % BK != 0) return fail(RGN_E_UNSUPPORTED, "gemm: K must be a multiple of 64");
    if ((lda % 8) || (ldw % 8) || (ldc % 8)) return fail(RGN_E_UNSUPPORTED, "gemm: row strides must be multiples of 8");
    if ((((uintptr_t)A | (uintptr_t)W | (uintptr_t)C | (uintptr_t)resid) & 15) != 0)
        return fail(RGN_E_UNSUPPORTED, "gemm: pointers must be 16-byte aligned");
    if (epilogue == RGN_EPI_GATE_RESID && (!gate || !resid)) return fail(RGN_E_BADARG, "gemm: gate/resid missing");
    return 0;
}

static inline int tiles(int M, int N, int b) { return ((M + b - 1) / b) * ((N + b - 1) / b); }

// ------------------------------------------------------------------------------------------------
// Launch planning.  Both tile configurations are modelled in microseconds with constants measured on
// MI355X (tools/bench_kernels.py gemm small; uniform random operands, sustained clocks):
//   256x256 (8 waves, 1 block/CU, 256 slots): a K step of 64 takes ~1.5 us (the chip is then at its power
//            cap, ~1.2-1.3 PFLOP/s) plus ~10 us of prologue + epilogue per tile.
//   128x128 (4 waves, 2 blocks/CU): ~0.33 us per K step for one block on a CU; two co-resident blocks
//            share the CU's MFMA rate, so a launch takes ceil(tiles/256) "rounds" of (nk x 0.36 + 8) us,
//            and never runs above ~1.05 PFLOP/s chip-wide.
// Round-aware schedule (same idea as attention_schedule): tiles that fill whole rounds of the chip's
// workgroup slots run as they are; the remainder is cut along K into nsplit pieces spread over all
// CUs (fp32 fragment partials in `ws`, ~0.07 us of L2/MALL traffic per partial tile) and finished by
// a reduce pass that owns the epilogue (~10 us for the two extra launches).
// ------------------------------------------------------------------------------------------------
struct GemmPlan { int nsplit; float cost_us; bool asm_pieces; };

static GemmPlan plan256(int nt, int K, bool can_split, size_t ws_bytes, bool asm4w, double wbytes) {
    const int slots = 256, nk = K / BK;
    const size_t tile_bytes = (size_t)256 * 256 * 4;
    const int full_rounds = nt / slots, left = nt - full_rounds * slots;
    // microseconds per K step of 64 and per tile (prologue + epilogue), tools/bench_kernels.py K sweeps:
    //   8-wave compiler-scheduled: 1.5 + 10 (FULL), 1.45 + 5 (PARTIAL: fp32 fragment dump instead of the epilogue), two extra
    //   launches (partial, reduce) ~10 and ~0.07 per partial tile of L2 / MALL traffic;
    //   4-wave hand-scheduled: 1.25 + 9 (FULL), 1.25 + 8 per piece (PARTIAL); the reduce launch reads 256 KiB per piece
    //   (~0.11 per piece, measured from 216 to 720 pieces) and runs the epilogue (~6 with the launch gap).
    const float ks = asm4w ? 1.25f : 1.5f, tile_us = asm4w ? 9.0f : 10.0f;
    const float full_cost = (float)full_rounds * ((float)nk * ks + tile_us);
    GemmPlan p{1, full_cost + (left > 0 ? (float)nk * ks + tile_us : 0.0f), false};
    if (left == 0 || !can_split || ws_bytes == 0) return p;
    const float base = p.cost_us;
    static const int asm_split_on = [] { const char* e = getenv("RGN_GEMM_ASM_SPLIT"); return e ? atoi(e) : 1; }();
    bool any_asm = false;
    for (int S = 2; S <= 8; ++S) {
        if (nk / S < 6) break;
        if ((size_t)left * S * tile_bytes > ws_bytes) break;
        const int blocks = left * S, rounds = (blocks + slots - 1) / slots;
        const float per = (float)((nk + S - 1) / S);
        // pieces on the hand-scheduled geometry where it applies (faster at every piece length measured, tools/probes/
        // fixup_ab.sh: R proj_out 162 -> 137 us, R ff2 137 -> 115), else on the 8-wave one; the reduce pass is the 8-wave one
        float cost;
        const bool asm_pieces = asm4w && asm_split_on;
        // small-M long-K problems are bound by the cold weight stream plus the partial dump, not by the K loop: ~3.5 TB/s
        // effective (R 5 % ff2 pair: 151 MB of W, S = 6 -> 93 us measured where the loop alone would take 48)
        const float hbm_us = (float)((wbytes + (double)blocks * 262144.0) / 3.5e6);
        if (asm_pieces) cost = full_cost + fmaxf((float)rounds * (per * 1.25f + 8.0f), full_rounds == 0 ? hbm_us : 0.0f) + 0.11f * (float)blocks + 6.0f;
        else if (S <= 6 && nk / S >= 8) cost = full_cost + (float)rounds * (per * 1.45f + 5.0f) + 0.07f * (float)blocks + 10.0f;
        else continue;
        if (cost < p.cost_us) { p.cost_us = cost; p.nsplit = S; p.asm_pieces = asm_pieces; any_asm = asm_pieces; }
    }
    static const float min_gain = [] { const char* e = getenv("RGN_GEMM_SPLIT_MIN_US"); return e ? (float)atof(e) : -1.0f; }();
    static const float min_frac = [] { const char* e = getenv("RGN_GEMM_SPLIT_MIN_FRAC"); return e ? (float)atof(e) : 0.05f; }();
    const float need = fmaxf(min_gain >= 0.f ? min_gain : (any_asm ? 8.0f : 30.0f), min_frac * base);
    if (base - p.cost_us < need) { p.nsplit = 1; p.cost_us = base; p.asm_pieces = false; }   // not worth the extra launches / the partial traffic
    return p;
}

// 128x128 remainder split: few tiles with a long K (text stream, small K_e) spread over the idle CUs
static int split128(int nt, int K, bool can_split, size_t ws_bytes) {
    const int slots = 512, nk = K / BK;
    const int left = nt % slots;
    if (left == 0 || !can_split) return 1;
    const size_t tile_bytes = (size_t)128 * 128 * 4;
    const float t_tile = 2.0f * 128 * 128 * (float)K / 4.3e12f * 1e6f;
    const float base = (float)((left + slots - 1) / slots) * t_tile;
    const float total = (float)((nt + slots - 1) / slots) * t_tile;
    float best_cost = base;
    int best = 1;
    for (int S = 2; S <= 6; ++S) {
        if (nk / S < 8) break;
        if ((size_t)left * S * tile_bytes > ws_bytes) break;
        const float traffic = (float)((size_t)left * S * tile_bytes * 2) / 4.0e12f * 1e6f;
        const float cost = (float)((left * S + slots - 1) / slots) * (t_tile / (float)S + 5.0f) + traffic + 10.0f;
        if (cost < best_cost) { best_cost = cost; best = S; }
    }
    if (base - best_cost < fmaxf(30.0f, 0.05f * total)) best = 1;
    return best;
}

static float estimate128(int nt, int K, double flops) {
    const int nk = K / BK, rounds = (nt + 255) / 256;
    // re-fitted in round 2 on COLD weights (every layer's W comes from HBM in the pipeline): R out pair 63 us, 144-tile long-K 255 us
    const float t = (float)rounds * ((float)nk * 0.40f + 12.0f);
    return fmaxf(t, (float)(flops / 1.05e15 * 1e6));
}

// launches of the 256x256 configuration: the hand-scheduled 4-wave geometry where it applies (a split-K remainder's
// PARTIAL and REDUCE launches must use the SAME geometry: the partial fragments are stored lane-linear)
template <int BM, int BN, int WM, int WN, int MODE>
static int launch_mode(const GemmGroup& gg, int epilogue, bool asm4w, hipStream_t st) {
    if (gg.p[0].w8) {                                 // fp8 W tiles: compiler-scheduled geometries, AV = 8
        if constexpr (MODE == MODE_REDUCE) return launch_gemm<BM, BN, WM, WN, MODE>(gg, epilogue, st);   // W is not touched
        else return launch_gemm<BM, BN, WM, WN, MODE, 8>(gg, epilogue, st);
    }
    if constexpr (BM == 256 && BN == 256) {
        // RGN_GEMM_ASMV: 0 = two 64 KiB stages, 1 = A ring of two + W ring of three 32 KiB slots (needs >= 4 K tiles)
        static const int asmv = [] { const char* e = getenv("RGN_GEMM_ASMV"); return e ? atoi(e) : 1; }();
        int nk = gg.p[0].K / BK;
        if (MODE == MODE_PARTIAL) {                       // shortest split piece
            const int per = (nk + gg.nsplit - 1) / gg.nsplit;
            nk = nk - (gg.nsplit - 1) * per;
        }
        if constexpr (MODE != MODE_REDUCE) {
            if (asm4w && asmv == 1 && nk >= 4) return launch_gemm<256, 256, 2, 2, MODE, 1>(gg, epilogue, st);
            if (asm4w) return launch_gemm<256, 256, 2, 2, MODE, 0>(gg, epilogue, st);
        }
    }
    return launch_gemm<BM, BN, WM, WN, MODE>(gg, epilogue, st);
}

template <int BM, int BN, int WM, int WN>
static int gemm_schedule(GemmGroup& gg, int epilogue, int slots, int nsplit, void* ws, hipStream_t st, bool asm4w = false,
                         bool asm_pieces = false) {
    const int nt = gg.nt;
    const int full = (nt / slots) * slots, left = nt - full;
    const char* v = getenv("RGN_GEMM_SPLIT");
    if ((v && v[0] == '0') || left == 0) nsplit = 1;
    gg.ws = (float*)ws;
    gg.part4w = 0;
    gg.nsplit = 1;
    static const int dbg = [] { const char* e = getenv("RGN_GEMM_DBG"); return e ? atoi(e) : 0; }();
    gg.dbg = dbg;
    int rc;
    if (nsplit == 1) {
        gg.tile_offset = 0; gg.nt_launch = nt;
        return launch_mode<BM, BN, WM, WN, MODE_FULL>(gg, epilogue, asm4w, st);
    }
    if (full > 0) {
        gg.tile_offset = 0; gg.nt_launch = full;
        if ((rc = launch_mode<BM, BN, WM, WN, MODE_FULL>(gg, epilogue, asm4w, st))) return rc;
    }
    gg.tile_offset = full; gg.nt_launch = left; gg.nsplit = nsplit;
    const int nk_all = gg.p[0].K / BK, per = (nk_all + nsplit - 1) / nsplit, shortest = nk_all - (nsplit - 1) * per;
    // hand-scheduled geometry for the pieces where the plan asks for it (long pieces; the asm K loop needs >= 2 K tiles in the
    // shortest piece, >= 4 for the ring variant).  The reduce pass stays on the 8-wave geometry and reads that layout (an
    // in-kernel finish by the last-arriving piece was measured and dropped).
    // RGN_GEMM_ASM_SPLIT=0 (A/B switch, read by the planner): 8-wave pieces.
    const bool asm_split = asm4w && asm_pieces && shortest >= 2;
    gg.part4w = asm_split ? 1 : 0;
    if ((rc = launch_mode<BM, BN, WM, WN, MODE_PARTIAL>(gg, epilogue, asm_split, st))) return rc;
    return launch_mode<BM, BN, WM, WN, MODE_REDUCE>(gg, epilogue, false, st);
}

// fp8 (e4m3fn) -> bf16, exact (every e4m3 value is a bf16 value): 16 bytes in, 32 bytes out per thread and step
__global__ __launch_bounds__(256) void widen_w8_kernel(const uint8_t* __restrict__ src, uint16_t* __restrict__ dst, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
        const uint4 raw = ((const uint4*)src)[i];
        const uint32_t w[4] = {raw.x, raw.y, raw.z, raw.w};
        uint32_t o[8];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const bf2_t lo = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w[k], 1.0f, false);
            const bf2_t hi = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w[k], 1.0f, true);
            o[2 * k] = *(const uint32_t*)&lo;
            o[2 * k + 1] = *(const uint32_t*)&hi;
        }
        ((uint4*)dst)[2 * i] = make_uint4(o[0], o[1], o[2], o[3]);
        ((uint4*)dst)[2 * i + 1] = make_uint4(o[4], o[5], o[6], o[7]);
    }
}

// Large-M problems on fp8 weights: the GEMM is MFMA-bound whatever the weight format (bf16 activations -> bf16 MFMA), and the
// hand-scheduled bf16 loop is ~15 % faster than the fp8-tile kernel, so the weight matrix is widened ONCE per call into the
// tail of the caller's workspace (exact conversion, N x K x 3 bytes of traffic = ~3 % of such a GEMM's time) and the bf16
// path runs on it; the per-channel scale still multiplies the fp32 accumulator in the epilogue, so the result is
// bit-identical to the fp8-tile kernel's.  Weights stay fp8 in HBM.  RGN_W8_WIDEN_MIN_M (default 2048 rows in total; 0 = never).
static int widen_w8(GemmGroup& gg, int nprob, void* ws, size_t& ws_bytes, hipStream_t st) {
    const char* mm = getenv("RGN_W8_WIDEN_MIN_M");
    const int min_m = mm ? atoi(mm) : 2048;
    if (!gg.p[0].w8 || min_m <= 0 || ws == nullptr) return 0;
    int m = 0;
    size_t need = 0;
    for (int i = 0; i < nprob; ++i) {
        m += gg.p[i].M;
        if (gg.p[i].ldw != gg.p[i].K || (gg.p[i].K % 16)) return 0;        // contiguous rows only
        need += ((size_t)gg.p[i].N * gg.p[i].K * 2 + 255) & ~(size_t)255;
    }
    if (m < min_m || need + ((size_t)64 << 20) > ws_bytes) return 0;         // keep >= 64 MiB for split-K partials
    uint8_t* dst = (uint8_t*)ws + ws_bytes - need;
    ws_bytes -= need;
    for (int i = 0; i < nprob; ++i) {
        const size_t n16 = (size_t)gg.p[i].N * gg.p[i].K / 16;
        hipLaunchKernelGGL(widen_w8_kernel, dim3((unsigned)((n16 + 255) / 256 < 4096 ? (n16 + 255) / 256 : 4096)), dim3(256), 0, st,
                           (const uint8_t*)gg.p[i].W, (uint16_t*)dst, n16);
        gg.p[i].W = (const uint16_t*)dst;
        gg.p[i].w8 = 0;
        dst += ((size_t)gg.p[i].N * gg.p[i].K * 2 + 255) & ~(size_t)255;
    }
    return check_launch("widen_w8_kernel");
}

static int gemm_dispatch(GemmGroup& gg, int nprob, int epilogue, void* ws, size_t ws_bytes, hipStream_t st) {
    if (int rc = widen_w8(gg, nprob, ws, ws_bytes, st)) return rc;
    int big = 0, small_ = 0;
    double flops = 0.0;
    for (int i = 0; i < nprob; ++i) {
        big += tiles(gg.p[i].M, gg.p[i].N, 256);
        small_ += tiles(gg.p[i].M, gg.p[i].N, 128);
        flops += 2.0 * gg.p[i].M * (double)gg.p[i].N * gg.p[i].K;
    }
    const int K = gg.p[0].K;
    const char* v = getenv("RGN_GEMM_VARIANT");
    // hand-scheduled 4-wave K loop for the 256x256 tiles: needs >= 2 K tiles and 32-bit operand offsets
    bool asm4w = K >= 2 * BK;
    for (int i = 0; i < nprob; ++i)
        asm4w = asm4w && (size_t)gg.p[i].M * gg.p[i].lda * 2 < ((size_t)1 << 32) && (size_t)gg.p[i].N * gg.p[i].ldw * 2 < ((size_t)1 << 32);
    static const int asm_default = [] { const char* e = getenv("RGN_GEMM_ASM"); return e ? atoi(e) : 1; }();   // RGN_GEMM_ASM=0: A/B switch
    if (!(asm_default || (v && v[0] == '3')) || (v && v[0] == '2')) asm4w = false;
    if (gg.p[0].w8) asm4w = false;
    double wbytes = 0.0;
    for (int i = 0; i < nprob; ++i) wbytes += (double)gg.p[i].N * gg.p[i].K * 2.0;
    const GemmPlan p256 = plan256(big, K, ws != nullptr, ws_bytes, asm4w, wbytes);
    const float e128 = estimate128(small_, K, flops);
    // the model is coarse: leave the habitual choice (256x256 from ~200 tiles up) only for a clear predicted win
    bool use_big = (big >= 200) ? !(e128 < 0.90f * p256.cost_us) : (p256.cost_us < e128);
    if (v && v[0] == '1') use_big = false;
    if (v && (v[0] == '2' || v[0] == '3')) use_big = true;
    const int b = use_big ? 256 : 128;
    int ns256 = p256.nsplit;
    bool asm_pieces = p256.asm_pieces;
    if (const char* f = getenv("RGN_GEMM_NSPLIT")) {          // measurement / test switch: piece count forced, asm pieces unless switched off
        if (atoi(f) > 0 && ws != nullptr) {
            ns256 = atoi(f);
            while (ns256 > 1 && (size_t)(big % 256) * ns256 * ((size_t)256 * 256 * 4) > ws_bytes) --ns256;      // must fit the workspace
            const char* e = getenv("RGN_GEMM_ASM_SPLIT");
            asm_pieces = asm4w && !(e && atoi(e) == 0);
        }
    }
    gg.nt0 = tiles(gg.p[0].M, gg.p[0].N, b);
    gg.nt = gg.nt0 + (nprob > 1 ? tiles(gg.p[1].M, gg.p[1].N, b) : 0);
    if (gg.nt == 0) return 0;
    return use_big ? gemm_schedule<256, 256, 2, 4>(gg, epilogue, 256, ns256, ws, st, asm4w, asm_pieces)
                   : gemm_schedule<128, 128, 2, 2>(gg, epilogue, 512, split128(gg.nt, K, ws != nullptr, ws_bytes), ws, st);
}

static void fill(GemmArgs& g, const void* A, int lda, const void* W, int ldw, const void* bias, void* C, int ldc, int M,
                 int N, int K, int gelu_from_col, const void* gate, const void* resid, const int64_t* out_rows) {
    g.A = (const uint16_t*)A; g.W = (const uint16_t*)W; g.bias = (const uint16_t*)bias; g.C = (uint16_t*)C;
    g.gate = (const uint16_t*)gate; g.resid = (const uint16_t*)resid; g.out_rows = out_rows;
    g.qkv = QkvEpi{};
    g.lda = lda; g.ldw = ldw; g.ldc = ldc; g.M = M; g.N = N; g.K = K;
    g.gelu_from_col = (gelu_from_col + 7) & ~7;      // the GELU boundary is honoured per 8-column vector (callers pass multiples of 8)
    g.wscale = nullptr;
    g.w8 = 0;
}

static int fill_qkv(GemmArgs& g, const rgn_qkv_epilogue* e, int N) {
    if (!e || !e->wq || !e->wk || !e->cos_q || !e->sin_q || !e->cos_k || !e->sin_k || !e->k_slab || !e->vt_slab)
        return fail(RGN_E_BADARG, "gemm_qkv: null pointer in rgn_qkv_epilogue");
    const int hd = e->heads * 128;
    if (e->heads <= 0 || (e->k_col % 256) || (e->v_col % 256) || (e->q_col % 256) || (hd % 256) || (e->skv_pad % 64) ||
        e->k_col + hd > N || e->v_col + hd > N || e->q_col + hd > N || e->row_base < 0)
        return fail(RGN_E_UNSUPPORTED, "gemm_qkv: k/v/q column blocks must be 256-aligned blocks of heads*128 columns inside N");
    if ((((uintptr_t)e->wq | (uintptr_t)e->wk | (uintptr_t)e->cos_q | (uintptr_t)e->sin_q | (uintptr_t)e->cos_k |
          (uintptr_t)e->sin_k | (uintptr_t)e->k_slab | (uintptr_t)e->vt_slab) & 15) != 0)
        return fail(RGN_E_UNSUPPORTED, "gemm_qkv: pointers must be 16-byte aligned");
    QkvEpi& q = g.qkv;
    q.wq = (const uint16_t*)e->wq; q.wk = (const uint16_t*)e->wk;
    q.cos_q = e->cos_q; q.sin_q = e->sin_q; q.cos_k = e->cos_k; q.sin_k = e->sin_k;
    q.kv_rows = e->kv_rows; q.k_slab = (uint16_t*)e->k_slab; q.vt_slab = (uint16_t*)e->vt_slab;
    q.row_base = e->row_base; q.skv_pad = e->skv_pad; q.k_col = e->k_col; q.v_col = e->v_col; q.q_col = e->q_col;
    q.hd = hd; q.eps = e->eps; q.fp16_roundtrip = e->fp16_roundtrip;
    return 0;
}

// fp8 weights (ws0 / ws1 = per-output-channel fp32 scales, nullptr = bf16 weights): W is [N, K] bytes (OCP e4m3fn), ldw in
// bytes and a multiple of 16
static int w8_check(const void* W, int ldw, const float* wscale) {
    if (wscale == nullptr) return 0;
    if ((ldw % 16) || (((uintptr_t)W | (uintptr_t)wscale) & 15)) return fail(RGN_E_UNSUPPORTED, "gemm_w8: W rows and the scale vector must be 16-byte aligned");
    return 0;
}

static int gemm1(const void* A, int lda, const void* W, int ldw, const float* wsc, const void* bias, void* C, int ldc, int M, int N,
                 int K, int epilogue, int gelu_from_col, const void* gate, const void* resid, const int64_t* out_rows,
                 void* workspace, size_t workspace_bytes, void* stream) {
    if (M == 0) return 0;
    int rc = check_problem(A, lda, W, wsc ? 8 : ldw, C, ldc, M, N, K, epilogue, gate, resid);
    if (rc || (rc = w8_check(W, ldw, wsc))) return rc;
    GemmGroup gg;
    fill(gg.p[0], A, lda, W, ldw, bias, C, ldc, M, N, K, gelu_from_col, gate, resid, out_rows);
    gg.p[0].wscale = wsc; gg.p[0].w8 = wsc != nullptr;
    gg.p[1] = gg.p[0];
    return gemm_dispatch(gg, 1, epilogue, workspace, workspace_bytes, (hipStream_t)stream);
}

static int gemm2(const void* A0, int lda0, const void* W0, const float* ws0, const void* bias0, void* C0, int ldc0, int M0,
                 const void* gate0, const void* resid0, const void* A1, int lda1, const void* W1, const float* ws1,
                 const void* bias1, void* C1, int ldc1, int M1, const void* gate1, const void* resid1, int N, int K,
                 int epilogue, int gelu_from_col, void* workspace, size_t workspace_bytes, void* stream) {
    if (M0 == 0 && M1 == 0) return 0;
    if ((ws0 == nullptr) != (ws1 == nullptr)) return fail(RGN_E_UNSUPPORTED, "gemm pair: both weight matrices must have the same format");
    if (M0 == 0) return gemm1(A1, lda1, W1, K, ws1, bias1, C1, ldc1, M1, N, K, epilogue, gelu_from_col, gate1, resid1, nullptr, workspace, workspace_bytes, stream);
    if (M1 == 0) return gemm1(A0, lda0, W0, K, ws0, bias0, C0, ldc0, M0, N, K, epilogue, gelu_from_col, gate0, resid0, nullptr, workspace, workspace_bytes, stream);
    int rc = check_problem(A0, lda0, W0, ws0 ? 8 : K, C0, ldc0, M0, N, K, epilogue, gate0, resid0);
    if (rc) return rc;
    rc = check_problem(A1, lda1, W1, ws1 ? 8 : K, C1, ldc1, M1, N, K, epilogue, gate1, resid1);
    if (rc || (rc = w8_check(W0, K, ws0)) || (rc = w8_check(W1, K, ws1))) return rc;
    GemmGroup gg;
    fill(gg.p[0], A0, lda0, W0, K, bias0, C0, ldc0, M0, N, K, gelu_from_col, gate0, resid0, nullptr);
    fill(gg.p[1], A1, lda1, W1, K, bias1, C1, ldc1, M1, N, K, gelu_from_col, gate1, resid1, nullptr);
    gg.p[0].wscale = ws0; gg.p[1].wscale = ws1; gg.p[0].w8 = gg.p[1].w8 = ws0 != nullptr;
    return gemm_dispatch(gg, 2, epilogue, workspace, workspace_bytes, (hipStream_t)stream);
}

static int gemm_qkv1(const void* A, int lda, const void* W, int ldw, const float* wsc, const void* bias, void* C, int ldc, int M,
                     int N, int K, int gelu_from_col, const rgn_qkv_epilogue* e, void* workspace, size_t workspace_bytes,
                     void* stream) {
    if (M == 0) return 0;
    int rc = check_problem(A, lda, W, wsc ? 8 : ldw, C, ldc, M, N, K, RGN_EPI_QKV, nullptr, nullptr);
    if (rc || (rc = w8_check(W, ldw, wsc))) return rc;
    GemmGroup gg;
    fill(gg.p[0], A, lda, W, ldw, bias, C, ldc, M, N, K, gelu_from_col, nullptr, nullptr, nullptr);
    if ((rc = fill_qkv(gg.p[0], e, N))) return rc;
    gg.p[0].wscale = wsc; gg.p[0].w8 = wsc != nullptr;
    gg.p[1] = gg.p[0];
    return gemm_dispatch(gg, 1, RGN_EPI_QKV, workspace, workspace_bytes, (hipStream_t)stream);
}

static int gemm_qkv2(const void* A0, int lda0, const void* W0, const float* ws0, const void* bias0, void* C0, int ldc0, int M0,
                     const rgn_qkv_epilogue* e0, const void* A1, int lda1, const void* W1, const float* ws1, const void* bias1,
                     void* C1, int ldc1, int M1, const rgn_qkv_epilogue* e1, int N, int K, void* workspace,
                     size_t workspace_bytes, void* stream) {
    if (M0 == 0 && M1 == 0) return 0;
    if ((ws0 == nullptr) != (ws1 == nullptr)) return fail(RGN_E_UNSUPPORTED, "gemm pair: both weight matrices must have the same format");
    if (M0 == 0) return gemm_qkv1(A1, lda1, W1, K, ws1, bias1, C1, ldc1, M1, N, K, N, e1, workspace, workspace_bytes, stream);
    if (M1 == 0) return gemm_qkv1(A0, lda0, W0, K, ws0, bias0, C0, ldc0, M0, N, K, N, e0, workspace, workspace_bytes, stream);
    int rc = check_problem(A0, lda0, W0, ws0 ? 8 : K, C0, ldc0, M0, N, K, RGN_EPI_QKV, nullptr, nullptr);
    if (rc) return rc;
    if ((rc = check_problem(A1, lda1, W1, ws1 ? 8 : K, C1, ldc1, M1, N, K, RGN_EPI_QKV, nullptr, nullptr))) return rc;
    if ((rc = w8_check(W0, K, ws0)) || (rc = w8_check(W1, K, ws1))) return rc;
    GemmGroup gg;
    fill(gg.p[0], A0, lda0, W0, K, bias0, C0, ldc0, M0, N, K, N, nullptr, nullptr, nullptr);
    fill(gg.p[1], A1, lda1, W1, K, bias1, C1, ldc1, M1, N, K, N, nullptr, nullptr, nullptr);
    if ((rc = fill_qkv(gg.p[0], e0, N))) return rc;
    if ((rc = fill_qkv(gg.p[1], e1, N))) return rc;
    gg.p[0].wscale = ws0; gg.p[1].wscale = ws1; gg.p[0].w8 = gg.p[1].w8 = ws0 != nullptr;
    return gemm_dispatch(gg, 2, RGN_EPI_QKV, workspace, workspace_bytes, (hipStream_t)stream);
}

extern "C" {

int rgn_gemm_bf16(const void* A, int lda, const void* W, int ldw, const void* bias, void* C, int ldc, int M, int N,
                  int K, int epilogue, int gelu_from_col, const void* gate, const void* resid,
                  const int64_t* out_rows, void* workspace, size_t workspace_bytes, void* stream) {
    return gemm1(A, lda, W, ldw, nullptr, bias, C, ldc, M, N, K, epilogue, gelu_from_col, gate, resid, out_rows, workspace,
                 workspace_bytes, stream);
}

int rgn_gemm_bf16_pair(const void* A0, int lda0, const void* W0, const void* bias0, void* C0, int ldc0, int M0,
                       const void* gate0, const void* resid0, const void* A1, int lda1, const void* W1,
                       const void* bias1, void* C1, int ldc1, int M1, const void* gate1, const void* resid1, int N,
                       int K, int epilogue, int gelu_from_col, void* workspace, size_t workspace_bytes, void* stream) {
    return gemm2(A0, lda0, W0, nullptr, bias0, C0, ldc0, M0, gate0, resid0, A1, lda1, W1, nullptr, bias1, C1, ldc1, M1, gate1,
                 resid1, N, K, epilogue, gelu_from_col, workspace, workspace_bytes, stream);
}

int rgn_gemm_bf16_qkv(const void* A, int lda, const void* W, int ldw, const void* bias, void* C, int ldc, int M, int N,
                      int K, int gelu_from_col, const rgn_qkv_epilogue* e, void* workspace, size_t workspace_bytes,
                      void* stream) {
    return gemm_qkv1(A, lda, W, ldw, nullptr, bias, C, ldc, M, N, K, gelu_from_col, e, workspace, workspace_bytes, stream);
}

int rgn_gemm_bf16_qkv_pair(const void* A0, int lda0, const void* W0, const void* bias0, void* C0, int ldc0, int M0,
                           const rgn_qkv_epilogue* e0, const void* A1, int lda1, const void* W1, const void* bias1,
                           void* C1, int ldc1, int M1, const rgn_qkv_epilogue* e1, int N, int K, void* workspace,
                           size_t workspace_bytes, void* stream) {
    return gemm_qkv2(A0, lda0, W0, nullptr, bias0, C0, ldc0, M0, e0, A1, lda1, W1, nullptr, bias1, C1, ldc1, M1, e1, N, K,
                     workspace, workspace_bytes, stream);
}

// ---- fp8 (OCP e4m3fn) weights with per-output-channel fp32 scales: same calls, W one byte per element -------------------
int rgn_gemm_w8(const void* A, int lda, const void* W8, int ldw, const float* wscale, const void* bias, void* C, int ldc, int M,
                int N, int K, int epilogue, int gelu_from_col, const void* gate, const void* resid, const int64_t* out_rows,
                void* workspace, size_t workspace_bytes, void* stream) {
    if (!wscale) return fail(RGN_E_BADARG, "gemm_w8: scale vector missing");
    return gemm1(A, lda, W8, ldw, wscale, bias, C, ldc, M, N, K, epilogue, gelu_from_col, gate, resid, out_rows, workspace,
                 workspace_bytes, stream);
}

int rgn_gemm_w8_pair(const void* A0, int lda0, const void* W0, const float* wscale0, const void* bias0, void* C0, int ldc0, int M0,
                     const void* gate0, const void* resid0, const void* A1, int lda1, const void* W1, const float* wscale1,
                     const void* bias1, void* C1, int ldc1, int M1, const void* gate1, const void* resid1, int N, int K,
                     int epilogue, int gelu_from_col, void* workspace, size_t workspace_bytes, void* stream) {
    if (!wscale0 || !wscale1) return fail(RGN_E_BADARG, "gemm_w8: scale vector missing");
    return gemm2(A0, lda0, W0, wscale0, bias0, C0, ldc0, M0, gate0, resid0, A1, lda1, W1, wscale1, bias1, C1, ldc1, M1, gate1,
                 resid1, N, K, epilogue, gelu_from_col, workspace, workspace_bytes, stream);
}

int rgn_gemm_w8_qkv(const void* A, int lda, const void* W8, int ldw, const float* wscale, const void* bias, void* C, int ldc, int M,
                    int N, int K, int gelu_from_col, const rgn_qkv_epilogue* e, void* workspace, size_t workspace_bytes,
                    void* stream) {
    if (!wscale) return fail(RGN_E_BADARG, "gemm_w8: scale vector missing");
    return gemm_qkv1(A, lda, W8, ldw, wscale, bias, C, ldc, M, N, K, gelu_from_col, e, workspace, workspace_bytes, stream);
}

int rgn_gemm_w8_qkv_pair(const void* A0, int lda0, const void* W0, const float* wscale0, const void* bias0, void* C0, int ldc0,
                         int M0, const rgn_qkv_epilogue* e0, const void* A1, int lda1, const void* W1, const float* wscale1,
                         const void* bias1, void* C1, int ldc1, int M1, const rgn_qkv_epilogue* e1, int N, int K,
                         void* workspace, size_t workspace_bytes, void* stream) {
    if (!wscale0 || !wscale1) return fail(RGN_E_BADARG, "gemm_w8: scale vector missing");
    return gemm_qkv2(A0, lda0, W0, wscale0, bias0, C0, ldc0, M0, e0, A1, lda1, W1, wscale1, bias1, C1, ldc1, M1, e1, N, K,
                     workspace, workspace_bytes, stream);
}

size_t rgn_gemm_workspace_bytes(void) { return (size_t)256 << 20; }

int rgn_gemv_bf16(const void* x, int ldx, const void* W, const void* bias, void* y, int ldy, int B, int N, int K,
                  int silu_input, void* stream) {
    if (!x || !W || !y || B < 1 || B > 4 || N <= 0 || K <= 0 || (K % 8) || (ldx % 8))
        return fail(RGN_E_BADARG, "gemv: bad argument");
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((N + 3) / 4), blk(256);
    const uint16_t *xx = (const uint16_t*)x, *ww = (const uint16_t*)W, *bb = (const uint16_t*)bias;
    uint16_t* yy = (uint16_t*)y;
    switch (B) {
        case 1: hipLaunchKernelGGL(gemv_bf16_kernel<1>, grid, blk, 0, st, xx, ldx, ww, bb, yy, ldy, N, K, silu_input); break;
        case 2: hipLaunchKernelGGL(gemv_bf16_kernel<2>, grid, blk, 0, st, xx, ldx, ww, bb, yy, ldy, N, K, silu_input); break;
        case 3: hipLaunchKernelGGL(gemv_bf16_kernel<3>, grid, blk, 0, st, xx, ldx, ww, bb, yy, ldy, N, K, silu_input); break;
        default: hipLaunchKernelGGL(gemv_bf16_kernel<4>, grid, blk, 0, st, xx, ldx, ww, bb, yy, ldy, N, K, silu_input); break;
    }
    return check_launch("gemv_bf16_kernel");
}

}  // extern "C"
