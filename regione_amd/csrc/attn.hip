// Region attention for gfx950: softmax(Q K^T * scale) V over the compacted query set (Sq rows)
// against the full Region-Instruction KV cache (Skv rows); non-causal, head_dim 128, bf16 MFMA with
// fp32 online softmax.  Replaces flash_attn_func / SDPA at RegionE/FluxKontext/inplace.py:796-806.
//
// Design (wave64 / MFMA 32x32x16, cdna_hip_programming.md Appendix B "fused attention"):
//   * workgroup = 4 waves = 128 query rows of one head; each wave owns 32 query rows;
//   * "swapped" products so that every softmax quantity is lane-local:
//        S^T[kv, q] = K[kv, :] . Q[q, :]      (A = K tile from LDS, B = Q fragments in registers)
//        O^T[d,  q] = V^T[d, :] . P^T[:, q]   (A = V^T tile from LDS, B = P packed from S^T registers)
//     lane (q = lane&31, half = lane>>5) holds, for ITS query row, 16 of every 32 scores and 64 of
//     the 128 output dims: row max / row sum / rescale never cross lanes except one xor-32 exchange;
//   * the cache stores V TRANSPOSED ([H*128, Skv], written by rgn_qk_norm_rope_store) with the kv
//     index permuted inside 16-groups so the S^T accumulator registers ARE the P operand - no LDS
//     round trip, no permlane shuffles for P;
//   * K / V^T tiles (64 kv) stream HBM/L2 -> LDS with global_load_lds_dwordx4, double buffered,
//     bank-conflict swizzle applied on the source address + read address;
//   * XCD-aware block map: consecutive (head, q-block) items stay on one XCD so a head's K/V
//     (4.4 MB at Skv = 8704) is fetched into that XCD's L2 once.
#include "common.h"

namespace rgn {

typedef __attribute__((ext_vector_type(8))) __bf16 bf8_t;
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

struct AttnArgs {
    const uint16_t* Q;
    const uint16_t* K;    // [skv_pad, H*128]
    const uint16_t* Vt;   // [H*128, skv_pad]
    uint16_t* O;
    int ldq, ldo, skv_pad, Sq, Skv, H;
    float scale_log2e;
};

constexpr int KV_T = 64;                       // kv rows per tile
constexpr int K_TILE_BYTES = KV_T * 128 * 2;   // 16 KiB  [64 kv][128 d]
constexpr int V_TILE_BYTES = 128 * KV_T * 2;   // 16 KiB  [128 d][64 kv]
constexpr int ATT_STAGE = K_TILE_BYTES + V_TILE_BYTES;

__device__ __forceinline__ uint32_t pack2(float a, float b) { return (uint32_t)f2bf(a) | ((uint32_t)f2bf(b) << 16); }

__global__ __launch_bounds__(256, 2) void attention_kernel(const AttnArgs g) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ql = lane & 31, half = lane >> 5;

    // ---- XCD-aware bijective item map: item = head * nQ + qblock ---------------------------------
    const int nQ = (g.Sq + 127) / 128, nitems = g.H * nQ;
    int item;
    {
        const int bid = blockIdx.x, xcd = bid & 7, loc = bid >> 3;
        const int q = nitems >> 3, r = nitems & 7;
        item = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    const int h = item / nQ, qb = item - h * nQ;
    const int q0 = qb * 128 + wave * 32;
    const size_t HD = (size_t)g.H * 128;

    // ---- Q fragments: B operand, lane (q, half) holds Q[q][ks*16 + half*8 .. +8] -----------------
    bf8_t qf[8];
    {
        const int qr = min(q0 + ql, g.Sq - 1);
        const uint16_t* qp = g.Q + (size_t)qr * g.ldq + h * 128 + half * 8;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) qf[ks] = *(const bf8_t*)(qp + ks * 16);
    }

    // ---- staging: K tile 16 pieces of 4 rows, V^T tile 16 pieces of 8 rows; 4+4 per wave ---------
    const uint8_t* k_src[4];
    const uint8_t* v_src[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int piece = wave * 4 + p;
        const int krow = piece * 4 + (lane >> 4);                 // kv row inside the tile
        const int kchunk = (lane & 15) ^ (krow & 15);             // source chunk for LDS slot lane&15
        k_src[p] = (const uint8_t*)(g.K + (size_t)krow * HD + h * 128) + kchunk * 16;
        const int vrow = piece * 8 + (lane >> 3);                 // d row inside the tile
        const int vchunk = (lane & 7) ^ ((vrow >> 1) & 7);
        v_src[p] = (const uint8_t*)(g.Vt + ((size_t)h * 128 + vrow) * g.skv_pad) + vchunk * 16;
    }
    auto stage = [&](int t, int buf) {
        uint8_t* base = smem + buf * ATT_STAGE + (wave * 4) * 1024;
        const size_t koff = (size_t)t * KV_T * HD * 2, voff = (size_t)t * KV_T * 2;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(k_src[p] + koff), (lds_ptr_t)(base + p * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(v_src[p] + voff),
                                             (lds_ptr_t)(base + K_TILE_BYTES + p * 1024), 16, 0, 0);
        }
    };

    // ---- LDS read offsets -------------------------------------------------------------------------
    // K frag (A of S^T): row = b*32 + ql, 16-B chunk c = ks*2 + half, slot = c ^ (row & 15)
    // V frag (A of O^T): row = db*32 + ql, chunk c = kb*2 + half (kb = 0..3), slot = c ^ ((row>>1)&7)
    const int k_row_off = ql * 256, k_sw = ql & 15;
    const int v_row_off = K_TILE_BYTES + ql * 128, v_sw = (ql >> 1) & 7;

    f32x16 o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
    float m_run = -1e30f, l_run = 0.f;

    const int ntiles = (g.Skv + KV_T - 1) / KV_T;
    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int t = 0; t < ntiles; ++t) {
        const int cur = t & 1;
        if (t + 1 < ntiles) stage(t + 1, cur ^ 1);
        const uint8_t* sb = smem + cur * ATT_STAGE;

        // ---- S^T = K Q^T : 2 kv-blocks x 8 k-steps ---------------------------------------------
        f32x16 s[2];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[b][r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const bf8_t kf = *(const bf8_t*)(sb + b * 32 * 256 + k_row_off + (((ks * 2 + half) ^ k_sw) << 4));
                s[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s[b], 0, 0, 0);
            }
        }
        // ---- tail mask ---------------------------------------------------------------------------
        const int kv0 = t * KV_T;
        if (kv0 + KV_T > g.Skv) {
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int kv = kv0 + b * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    if (kv >= g.Skv) s[b][r] = -INFINITY;
                }
        }
        // ---- online softmax (lane-local + one xor-32 exchange) -----------------------------------
        float mx = s[0][0];
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[b][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * g.scale_log2e);
        const float mneg = m_new * g.scale_log2e;
        float psum = 0.f;
        bf8_t pf[2][2];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            float p[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                p[r] = __builtin_amdgcn_exp2f(s[b][r] * g.scale_log2e - mneg);
                psum += p[r];
            }
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                uint32_t w[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) w[j] = pack2(p[kb * 8 + 2 * j], p[kb * 8 + 2 * j + 1]);
                pf[b][kb] = *(bf8_t*)w;
            }
        }
        l_run = l_run * alpha + psum;
        m_run = m_new;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[i][r] *= alpha;
        // ---- O^T += V^T P^T : 4 d-blocks x 4 k-blocks ---------------------------------------------
#pragma unroll
        for (int db = 0; db < 4; ++db) {
#pragma unroll
            for (int kb4 = 0; kb4 < 4; ++kb4) {
                const bf8_t vf = *(const bf8_t*)(sb + db * 32 * 128 + v_row_off + (((kb4 * 2 + half) ^ v_sw) << 4));
                o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[kb4 >> 1][kb4 & 1], o[db], 0, 0, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    // ---- finalize: O = O^T / l, staged through LDS for 16-byte row-contiguous stores ----------------
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    uint16_t* ot = (uint16_t*)smem;                               // [128 q][128 + 8] bf16
    constexpr int OT_LD = 136;
    {
        uint16_t* orow = ot + (wave * 32 + ql) * OT_LD;
#pragma unroll
        for (int db = 0; db < 4; ++db)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const int d = db * 32 + 8 * r4 + 4 * half;
                uint32_t w0 = pack2(o[db][r4 * 4 + 0] * inv, o[db][r4 * 4 + 1] * inv);
                uint32_t w1 = pack2(o[db][r4 * 4 + 2] * inv, o[db][r4 * 4 + 3] * inv);
                *(uint2*)(orow + d) = make_uint2(w0, w1);
            }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int row = (tid >> 4) + it * 16, c = (tid & 15) * 8;
        const int qr = qb * 128 + row;
        if (qr < g.Sq) *(uint4*)(g.O + (size_t)qr * g.ldo + h * 128 + c) = *(const uint4*)(ot + row * OT_LD + c);
    }
}

}  // namespace rgn

using namespace rgn;

extern "C" {

int rgn_attention(const void* Q, int ldq, const void* k_slab, const void* vt_slab, int skv_pad, void* O, int ldo,
                  int Sq, int Skv, int H, float scale, void* stream) {
    if (Sq == 0) return 0;
    if (!Q || !k_slab || !vt_slab || !O || Sq < 0 || Skv <= 0 || H <= 0 || (skv_pad % 64) || skv_pad < Skv ||
        (ldq % 8) || (ldo % 8))
        return fail(RGN_E_BADARG, "attention: bad argument");
    AttnArgs g;
    g.Q = (const uint16_t*)Q; g.K = (const uint16_t*)k_slab; g.Vt = (const uint16_t*)vt_slab; g.O = (uint16_t*)O;
    g.ldq = ldq; g.ldo = ldo; g.skv_pad = skv_pad; g.Sq = Sq; g.Skv = Skv; g.H = H;
    g.scale_log2e = scale * 1.4426950408889634f;
    const int nitems = H * ((Sq + 127) / 128);
    const size_t lds = 2 * ATT_STAGE;
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)attention_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr = true;
    }
    hipLaunchKernelGGL(attention_kernel, dim3(nitems), dim3(256), lds, (hipStream_t)stream, g);
    return check_launch("attention_kernel");
}

}  // extern "C"
