// Region attention for gfx950: softmax(Q K^T * scale) V over the compacted query set (Sq rows)
// against the full Region-Instruction KV cache (Skv rows); non-causal, head_dim 128, bf16 MFMA with
// fp32 online softmax.  Replaces flash_attn_func / SDPA at RegionE/FluxKontext/inplace.py:796-806.
//
// Design (wave64 / MFMA 32x32x16, cdna_hip_programming.md Appendix B "fused attention"):
//   * workgroup = NW waves (4 or 8) = 32*NW query rows of one head; each wave owns 32 query rows;
//   * "swapped" products so that every softmax quantity is lane-local:
//        S^T[kv, q] = K[kv, :] . Q[q, :]      (A = K tile from LDS, B = Q fragments in registers)
//        O^T[d,  q] = V^T[d, :] . P^T[:, q]   (A = V^T tile from LDS, B = P packed from S^T registers)
//     lane (q = lane&31, half = lane>>5) holds, for ITS query row, 16 of every 32 scores and 64 of
//     the 128 output dims: row max / row sum / rescale never cross lanes except one xor-32 exchange;
//   * the cache stores V TRANSPOSED ([H*128, Skv], written by rgn_qk_norm_rope_store) with the kv
//     index permuted inside 16-groups so the S^T accumulator registers ARE the P operand - no LDS
//     round trip, no permlane shuffles for P;
//   * K / V^T tiles (64 kv) stream L2 -> LDS with global_load_lds_dwordx4 into an NSTAGE ring; loads
//     stay in flight across the (raw) barrier behind a counted s_waitcnt vmcnt(N) - one barrier per
//     tile, never a drain to 0 in steady state; bank-conflict swizzle on source + read address;
//   * VALU diet (the softmax, not the MFMA, is the co-bottleneck at 2 waves/SIMD): hardware
//     v_cvt_pk_bf16_f32 packing, scale folded into the exp2 argument, deferred max (T13): the running
//     max is only raised - and O rescaled - when some row's tile max exceeds it by > 2^8;
//   * XCD-aware block map: consecutive (head, q-block) items stay on one XCD so a head's K/V
//     (4.4 MB at Skv = 8704) is fetched into that XCD's L2 once.
#include "common.h"
#include <stdlib.h>

namespace rgn {

typedef __attribute__((ext_vector_type(8))) __bf16 bf8_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf2_t;
typedef __attribute__((ext_vector_type(2))) float f32x2;
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
constexpr float DEFER_THR = 8.0f;              // log2 units: P <= 2^8 before the running max is raised

__device__ __forceinline__ uint32_t cvt_pk_bf16(float a, float b) {
    bf2_t r = __builtin_convertvector(f32x2{a, b}, bf2_t);     // v_cvt_pk_bf16_f32 (RNE)
    return *(uint32_t*)&r;
}

template <int N>
__device__ __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int NW, int NSTAGE, int ABL = 0>
__global__ __launch_bounds__(64 * NW, (NW >= 8) ? 1 : 2) void attention_kernel(const AttnArgs g) {
    constexpr int QB = 32 * NW;                 // query rows per workgroup
    constexpr int PW = 16 / NW;                 // K pieces (= V pieces) per wave per stage
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ql = lane & 31, half = lane >> 5;

    // ---- XCD-aware bijective item map: item = head * nQ + qblock ---------------------------------
    const int nQ = (g.Sq + QB - 1) / QB, nitems = g.H * nQ;
    int item;
    {
        const int bid = blockIdx.x, xcd = bid & 7, loc = bid >> 3;
        const int q = nitems >> 3, r = nitems & 7;
        item = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    const int h = item / nQ, qb = item - h * nQ;
    const int q0 = qb * QB + wave * 32;
    const size_t HD = (size_t)g.H * 128;

    // ---- Q fragments: B operand, lane (q, half) holds Q[q][ks*16 + half*8 .. +8] -----------------
    bf8_t qf[8];
    {
        const int qr = min(q0 + ql, g.Sq - 1);
        const uint16_t* qp = g.Q + (size_t)qr * g.ldq + h * 128 + half * 8;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) qf[ks] = *(const bf8_t*)(qp + ks * 16);
        // consume the loads HERE so hipcc places its vmcnt wait before the DMA pipeline starts, not
        // at the first MFMA inside the tile loop (where it would drain the in-flight K/V stages)
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) asm volatile("" : "+v"(qf[ks]));
    }

    // ---- staging: K tile = 16 pieces of 4 rows, V^T tile = 16 pieces of 8 rows ---------------------
    const uint8_t* k_src[PW];
    const uint8_t* v_src[PW];
#pragma unroll
    for (int p = 0; p < PW; ++p) {
        const int piece = wave * PW + p;
        const int krow = piece * 4 + (lane >> 4);                 // kv row inside the tile
        const int kchunk = (lane & 15) ^ (krow & 15);             // source chunk for LDS slot lane&15
        k_src[p] = (const uint8_t*)(g.K + (size_t)krow * HD + h * 128) + kchunk * 16;
        const int vrow = piece * 8 + (lane >> 3);                 // d row inside the tile
        const int vchunk = (lane & 7) ^ ((vrow >> 1) & 7);
        v_src[p] = (const uint8_t*)(g.Vt + ((size_t)h * 128 + vrow) * g.skv_pad) + vchunk * 16;
    }
    auto stage = [&](int t, int buf) {
        uint8_t* base = smem + buf * ATT_STAGE + (wave * PW) * 1024;
        const size_t koff = (size_t)t * KV_T * HD * 2, voff = (size_t)t * KV_T * 2;
#pragma unroll
        for (int p = 0; p < PW; ++p) {
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
    float m_run = -1e30f;          // running max in scaled-log2 units
    float l_run = 0.f;
    const float sl2e = g.scale_log2e;

    const int ntiles = (g.Skv + KV_T - 1) / KV_T;
#pragma unroll
    for (int s = 0; s < NSTAGE - 1; ++s)
        if (s < ntiles) stage(s, s);
    int cur = 0, nxt = NSTAGE - 1;
    for (int t = 0; t < ntiles; ++t) {
        // tile t landed for this wave: leave the newer stages (at most NSTAGE-2) in flight
        const int newer = min(NSTAGE - 2, ntiles - 1 - t);
        if (NSTAGE >= 4 && newer == 2) wait_vm<2 * 2 * PW>();
        else if (NSTAGE >= 3 && newer == 1) wait_vm<2 * PW>();
        else wait_vm<0>();
        __builtin_amdgcn_s_barrier();
        if (ABL != 4 && t + NSTAGE - 1 < ntiles) stage(t + NSTAGE - 1, nxt);
        const uint8_t* sb = smem + ((ABL == 4) ? 0 : cur) * ATT_STAGE;
        cur = (cur + 1 == NSTAGE) ? 0 : cur + 1;
        nxt = (nxt + 1 == NSTAGE) ? 0 : nxt + 1;

        // ---- S^T = K Q^T : 2 kv-blocks x 8 k-steps, the two accumulators interleaved --------------
        f32x16 s0, s1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s0[r] = 0.f; s1[r] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const int sl = ((ks * 2 + half) ^ k_sw) << 4;
            const bf8_t kf0 = (ABL == 5) ? qf[(ks + 1) & 7] : *(const bf8_t*)(sb + k_row_off + sl);
            const bf8_t kf1 = (ABL == 5) ? qf[(ks + 2) & 7] : *(const bf8_t*)(sb + 32 * 256 + k_row_off + sl);
            s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf0, qf[ks], s0, 0, 0, 0);
            s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf1, qf[ks], s1, 0, 0, 0);
        }
        // ---- tail mask ---------------------------------------------------------------------------
        const int kv0 = t * KV_T;
        if (kv0 + KV_T > g.Skv) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int kv = kv0 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (kv >= g.Skv) s0[r] = -INFINITY;
                if (kv + 32 >= g.Skv) s1[r] = -INFINITY;
            }
        }
        // ---- online softmax with deferred max -------------------------------------------------------
        if (ABL == 3) {   // ablation: no QK^T result dependence (keep MFMAs alive)
#pragma unroll
            for (int r = 0; r < 16; ++r) { asm volatile("" ::"v"(s0[r]), "v"(s1[r])); }
        }
        float mx = fmaxf(s0[0], s1[0]);
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, fmaxf(s0[r], s1[r]));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64)) * sl2e;
        if (__any(mx > m_run + DEFER_THR)) {
            const float m_new = fmaxf(m_run, mx);
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            m_run = m_new;
            l_run *= alpha;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[i][r] *= alpha;
        }
        float psum = 0.f;
        bf8_t pf[4];
        {
            float p0[16], p1[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (ABL == 1) { p0[r] = s0[r]; p1[r] = s1[r]; }      // ablation: no exp
                else {
                    p0[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s0[r], sl2e, -m_run));
                    p1[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s1[r], sl2e, -m_run));
                }
                psum += p0[r] + p1[r];
            }
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                uint32_t w0[4], w1[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    w0[j] = cvt_pk_bf16(p0[kb * 8 + 2 * j], p0[kb * 8 + 2 * j + 1]);
                    w1[j] = cvt_pk_bf16(p1[kb * 8 + 2 * j], p1[kb * 8 + 2 * j + 1]);
                }
                pf[kb] = *(bf8_t*)w0;
                pf[2 + kb] = *(bf8_t*)w1;
            }
        }
        l_run += psum;
        if (ABL == 2) {    // ablation: no PV MFMAs (keep P alive)
#pragma unroll
            for (int k = 0; k < 4; ++k) asm volatile("" ::"v"(pf[k]));
            continue;
        }
        // ---- O^T += V^T P^T : 4 k-blocks x 4 d-blocks (independent accumulators back to back) -------
#pragma unroll
        for (int kb4 = 0; kb4 < 4; ++kb4) {
            const int sl = ((kb4 * 2 + half) ^ v_sw) << 4;
#pragma unroll
            for (int db = 0; db < 4; ++db) {
                const bf8_t vf = (ABL == 5) ? qf[(db + kb4) & 7] : *(const bf8_t*)(sb + db * 32 * 128 + v_row_off + sl);
                o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[kb4], o[db], 0, 0, 0);
            }
        }
    }

    // ---- finalize: O = O^T / l, staged through LDS for 16-byte row-contiguous stores ----------------
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    constexpr int OT_LD = 136;
    constexpr int OCHUNK = (NSTAGE * ATT_STAGE >= QB * OT_LD * 2) ? 1 : 2;     // O tile may need two passes
    constexpr int WPC = NW / OCHUNK;                                            // waves per pass
    uint16_t* ot = (uint16_t*)smem;
    __syncthreads();                                   // every wave is done reading the last K/V stage
#pragma unroll
    for (int ch = 0; ch < OCHUNK; ++ch) {
        if (wave / WPC == ch) {
            uint16_t* orow = ot + ((wave % WPC) * 32 + ql) * OT_LD;
#pragma unroll
            for (int db = 0; db < 4; ++db)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const int d = db * 32 + 8 * r4 + 4 * half;
                    const uint32_t w0 = cvt_pk_bf16(o[db][r4 * 4 + 0] * inv, o[db][r4 * 4 + 1] * inv);
                    const uint32_t w1 = cvt_pk_bf16(o[db][r4 * 4 + 2] * inv, o[db][r4 * 4 + 3] * inv);
                    *(uint2*)(orow + d) = make_uint2(w0, w1);
                }
        }
        __syncthreads();
        constexpr int ROWS = 32 * WPC, RPP = (64 * NW) / 16;
#pragma unroll
        for (int it = 0; it < ROWS / RPP; ++it) {
            const int row = (tid >> 4) + it * RPP, c = (tid & 15) * 8;
            const int qr = qb * QB + ch * ROWS + row;
            if (qr < g.Sq) *(uint4*)(g.O + (size_t)qr * g.ldo + h * 128 + c) = *(const uint4*)(ot + row * OT_LD + c);
        }
        if (ch + 1 < OCHUNK) __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// Software-pipelined variant (T15 "compute[next] || finish[cur]"): inside ONE wave the 16 QK^T MFMAs
// of tile t+1 are issued in the same basic block as the softmax VALU work of tile t, so the matrix
// pipe runs under the exp/pack stream instead of idling behind it (all waves of a workgroup are
// re-aligned by the per-tile barrier, so cross-wave staggering alone cannot provide that overlap).
// Ring of 3 stages: K(t+1) and V(t) are live while tile t+2 streams in.
// ------------------------------------------------------------------------------------------------
template <int NW>
__global__ __launch_bounds__(64 * NW, (NW >= 8) ? 1 : 2) void attention_pipe_kernel(const AttnArgs g) {
    constexpr int NSTAGE = 3;
    constexpr int QB = 32 * NW;
    constexpr int PW = 16 / NW;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ql = lane & 31, half = lane >> 5;
    const int nQ = (g.Sq + QB - 1) / QB, nitems = g.H * nQ;
    int item;
    {
        const int bid = blockIdx.x, xcd = bid & 7, loc = bid >> 3;
        const int q = nitems >> 3, r = nitems & 7;
        item = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    const int h = item / nQ, qb = item - h * nQ;
    const int q0 = qb * QB + wave * 32;
    const size_t HD = (size_t)g.H * 128;

    bf8_t qf[8];
    {
        const int qr = min(q0 + ql, g.Sq - 1);
        const uint16_t* qp = g.Q + (size_t)qr * g.ldq + h * 128 + half * 8;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) qf[ks] = *(const bf8_t*)(qp + ks * 16);
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) asm volatile("" : "+v"(qf[ks]));
    }
    const uint8_t* k_src[PW];
    const uint8_t* v_src[PW];
#pragma unroll
    for (int p = 0; p < PW; ++p) {
        const int piece = wave * PW + p;
        const int krow = piece * 4 + (lane >> 4);
        const int kchunk = (lane & 15) ^ (krow & 15);
        k_src[p] = (const uint8_t*)(g.K + (size_t)krow * HD + h * 128) + kchunk * 16;
        const int vrow = piece * 8 + (lane >> 3);
        const int vchunk = (lane & 7) ^ ((vrow >> 1) & 7);
        v_src[p] = (const uint8_t*)(g.Vt + ((size_t)h * 128 + vrow) * g.skv_pad) + vchunk * 16;
    }
    auto stage = [&](int t, int buf) {
        uint8_t* base = smem + buf * ATT_STAGE + (wave * PW) * 1024;
        const size_t koff = (size_t)t * KV_T * HD * 2, voff = (size_t)t * KV_T * 2;
#pragma unroll
        for (int p = 0; p < PW; ++p) {
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(k_src[p] + koff), (lds_ptr_t)(base + p * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(v_src[p] + voff),
                                             (lds_ptr_t)(base + K_TILE_BYTES + p * 1024), 16, 0, 0);
        }
    };
    const int k_row_off = ql * 256, k_sw = ql & 15;
    const int v_row_off = K_TILE_BYTES + ql * 128, v_sw = (ql >> 1) & 7;

    f32x16 o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
    float m_run = -1e30f, l_run = 0.f;
    const float sl2e = g.scale_log2e;
    const int ntiles = (g.Skv + KV_T - 1) / KV_T;

    auto qk = [&](const uint8_t* sb, f32x16& s0, f32x16& s1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) { s0[r] = 0.f; s1[r] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const int sl = ((ks * 2 + half) ^ k_sw) << 4;
            const bf8_t kf0 = *(const bf8_t*)(sb + k_row_off + sl);
            const bf8_t kf1 = *(const bf8_t*)(sb + 32 * 256 + k_row_off + sl);
            s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf0, qf[ks], s0, 0, 0, 0);
            s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf1, qf[ks], s1, 0, 0, 0);
        }
    };

    stage(0, 0);
    if (ntiles > 1) stage(1, 1);
    if (ntiles > 1) wait_vm<2 * PW>(); else wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    f32x16 c0, c1;                                   // S^T of the current tile
    qk(smem, c0, c1);
    // tile-0 mask + running-max initialisation (the loop always enters with m_run valid for c0/c1)
    auto tile_max = [&](int t, f32x16& a0, f32x16& a1) -> float {
        const int kv0 = t * KV_T;
        if (kv0 + KV_T > g.Skv) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int kv = kv0 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (kv >= g.Skv) a0[r] = -INFINITY;
                if (kv + 32 >= g.Skv) a1[r] = -INFINITY;
            }
        }
        float mx = fmaxf(a0[0], a1[0]);
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, fmaxf(a0[r], a1[r]));
        return fmaxf(mx, __shfl_xor(mx, 32, 64)) * sl2e;
    };
    auto raise_max = [&](float mx) {
        if (__any(mx > m_run + DEFER_THR)) {
            const float m_new = fmaxf(m_run, mx);
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            m_run = m_new;
            l_run *= alpha;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[i][r] *= alpha;
        }
    };
    raise_max(tile_max(0, c0, c1));
    int cur = 0;
    for (int t = 0; t < ntiles; ++t) {
        const int nb = (cur + 1 == NSTAGE) ? 0 : cur + 1;          // buffer of tile t+1
        const int lb = (nb + 1 == NSTAGE) ? 0 : nb + 1;            // buffer of tile t+2 (= tile t-1's)
        wait_vm<0>();                                              // tile t+1 (issued one iteration ago) landed
        __builtin_amdgcn_s_barrier();                              // ... for every wave; PV(t-1) finished everywhere
        if (t + 2 < ntiles) stage(t + 2, lb);
        const uint8_t* sb = smem + cur * ATT_STAGE;
        // ---- block 1: QK^T of tile t+1 (MFMA + LDS) interleaved with exp/pack of tile t (VALU) -------
        // (on the last tile the QK^T runs on a stale buffer and its result is discarded)
        f32x16 n0, n1;
        qk(smem + nb * ATT_STAGE, n0, n1);
        float psum = 0.f;
        bf8_t pf[4];
        {
            float p0[16], p1[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                p0[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(c0[r], sl2e, -m_run));
                p1[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(c1[r], sl2e, -m_run));
                psum += p0[r] + p1[r];
            }
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                uint32_t w0[4], w1[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    w0[j] = cvt_pk_bf16(p0[kb * 8 + 2 * j], p0[kb * 8 + 2 * j + 1]);
                    w1[j] = cvt_pk_bf16(p1[kb * 8 + 2 * j], p1[kb * 8 + 2 * j + 1]);
                }
                pf[kb] = *(bf8_t*)w0;
                pf[2 + kb] = *(bf8_t*)w1;
            }
        }
        l_run += psum;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     // 1 MFMA
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);     // 1 DS read
            __builtin_amdgcn_sched_group_barrier(0x002, 9, 0);     // 9 VALU (incl. transcendental)
        }
        // ---- block 2: PV of tile t (MFMA + LDS) interleaved with the row max of tile t+1 (VALU) ------
#pragma unroll
        for (int kb4 = 0; kb4 < 4; ++kb4) {
            const int sl = ((kb4 * 2 + half) ^ v_sw) << 4;
#pragma unroll
            for (int db = 0; db < 4; ++db) {
                const bf8_t vf = *(const bf8_t*)(sb + db * 32 * 128 + v_row_off + sl);
                o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[kb4], o[db], 0, 0, 0);
            }
        }
        float mxn = -1e30f;
        if (t + 1 < ntiles) mxn = tile_max(t + 1, n0, n1);
        raise_max(mxn);
        c0 = n0; c1 = n1;
        cur = nb;
    }

    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    constexpr int OT_LD = 136;
    uint16_t* ot = (uint16_t*)smem;                    // 3 stages = 96 KiB >= 256 x 136 x 2
    __syncthreads();
    {
        uint16_t* orow = ot + (wave * 32 + ql) * OT_LD;
#pragma unroll
        for (int db = 0; db < 4; ++db)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const int d = db * 32 + 8 * r4 + 4 * half;
                const uint32_t w0 = cvt_pk_bf16(o[db][r4 * 4 + 0] * inv, o[db][r4 * 4 + 1] * inv);
                const uint32_t w1 = cvt_pk_bf16(o[db][r4 * 4 + 2] * inv, o[db][r4 * 4 + 3] * inv);
                *(uint2*)(orow + d) = make_uint2(w0, w1);
            }
    }
    __syncthreads();
    constexpr int RPP = (64 * NW) / 16;
#pragma unroll
    for (int it = 0; it < QB / RPP; ++it) {
        const int row = (tid >> 4) + it * RPP, c = (tid & 15) * 8;
        const int qr = qb * QB + row;
        if (qr < g.Sq) *(uint4*)(g.O + (size_t)qr * g.ldo + h * 128 + c) = *(const uint4*)(ot + row * OT_LD + c);
    }
}

template <int NW>
static int launch_attention_pipe(const AttnArgs& g, hipStream_t st) {
    constexpr int QB = 32 * NW;
    const int nitems = g.H * ((g.Sq + QB - 1) / QB);
    constexpr int LDS = 3 * ATT_STAGE;
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)attention_pipe_kernel<NW>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        attr = true;
    }
    hipLaunchKernelGGL((attention_pipe_kernel<NW>), dim3(nitems), dim3(64 * NW), LDS, st, g);
    return check_launch("attention_pipe_kernel");
}

// ------------------------------------------------------------------------------------------------
// Ping-pong variant: the two waves that share a SIMD (wave w and w + NW/2) run the three per-tile
// phases  QK^T (matrix) -> softmax (VALU) -> PV (matrix)  ONE PHASE APART, separated by workgroup
// barriers, so that a SIMD's matrix pipe and its VALU are busy at the same time:
//      interval 1:  A: QK(t)   | B: PV(t-1)        (matrix | matrix)
//      interval 2:  A: SM(t)   | B: QK(t)          (VALU   | matrix)
//      interval 3:  A: PV(t)   | B: SM(t)          (matrix | VALU)
// A lock-step workgroup (every wave in the same phase, re-aligned by the per-tile barrier) measured
// 3780 cycles per tile per SIMD = MFMA (2048) + VALU (~1800) fully serialised (rocprofv3 PMC:
// SQ_WAVE_CYCLES vs SQ_VALU_MFMA_BUSY_CYCLES); the one-phase skew lets them overlap.
// Ring of 3 K/V stages: tile t-1 (B's PV), tile t, tile t+1 landing.
// ------------------------------------------------------------------------------------------------
template <int NW>
__global__ __launch_bounds__(64 * NW, 1) void attention_pp_kernel(const AttnArgs g) {
    constexpr int NSTAGE = 3;
    constexpr int QB = 32 * NW;
    constexpr int PW = 16 / NW;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool grpB = wave >= NW / 2;
    const int ql = lane & 31, half = lane >> 5;
    const int nQ = (g.Sq + QB - 1) / QB, nitems = g.H * nQ;
    int item;
    {
        const int bid = blockIdx.x, xcd = bid & 7, loc = bid >> 3;
        const int q = nitems >> 3, r = nitems & 7;
        item = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    const int h = item / nQ, qb = item - h * nQ;
    const int q0 = qb * QB + wave * 32;
    const size_t HD = (size_t)g.H * 128;

    bf8_t qf[8];
    {
        const int qr = min(q0 + ql, g.Sq - 1);
        const uint16_t* qp = g.Q + (size_t)qr * g.ldq + h * 128 + half * 8;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) qf[ks] = *(const bf8_t*)(qp + ks * 16);
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) asm volatile("" : "+v"(qf[ks]));
    }
    const uint8_t* k_src[PW];
    const uint8_t* v_src[PW];
#pragma unroll
    for (int p = 0; p < PW; ++p) {
        const int piece = wave * PW + p;
        const int krow = piece * 4 + (lane >> 4);
        const int kchunk = (lane & 15) ^ (krow & 15);
        k_src[p] = (const uint8_t*)(g.K + (size_t)krow * HD + h * 128) + kchunk * 16;
        const int vrow = piece * 8 + (lane >> 3);
        const int vchunk = (lane & 7) ^ ((vrow >> 1) & 7);
        v_src[p] = (const uint8_t*)(g.Vt + ((size_t)h * 128 + vrow) * g.skv_pad) + vchunk * 16;
    }
    auto stage = [&](int t, int buf) {
        uint8_t* base = smem + buf * ATT_STAGE + (wave * PW) * 1024;
        const size_t koff = (size_t)t * KV_T * HD * 2, voff = (size_t)t * KV_T * 2;
#pragma unroll
        for (int p = 0; p < PW; ++p) {
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(k_src[p] + koff), (lds_ptr_t)(base + p * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(v_src[p] + voff),
                                             (lds_ptr_t)(base + K_TILE_BYTES + p * 1024), 16, 0, 0);
        }
    };
    const int k_row_off = ql * 256, k_sw = ql & 15;
    const int v_row_off = K_TILE_BYTES + ql * 128, v_sw = (ql >> 1) & 7;

    f32x16 o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
    float m_run = -1e30f, l_run = 0.f;
    const float sl2e = g.scale_log2e;
    const int ntiles = (g.Skv + KV_T - 1) / KV_T;
    f32x16 s0, s1;
    bf8_t pf[4];
#pragma unroll
    for (int r = 0; r < 16; ++r) { s0[r] = 0.f; s1[r] = 0.f; }

    auto qk = [&](const uint8_t* sb) {
#pragma unroll
        for (int r = 0; r < 16; ++r) { s0[r] = 0.f; s1[r] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const int sl = ((ks * 2 + half) ^ k_sw) << 4;
            const bf8_t kf0 = *(const bf8_t*)(sb + k_row_off + sl);
            const bf8_t kf1 = *(const bf8_t*)(sb + 32 * 256 + k_row_off + sl);
            s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf0, qf[ks], s0, 0, 0, 0);
            s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf1, qf[ks], s1, 0, 0, 0);
        }
    };
    auto sm = [&](int t) {
        const int kv0 = t * KV_T;
        if (kv0 + KV_T > g.Skv) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int kv = kv0 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (kv >= g.Skv) s0[r] = -INFINITY;
                if (kv + 32 >= g.Skv) s1[r] = -INFINITY;
            }
        }
        float mx = fmaxf(s0[0], s1[0]);
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, fmaxf(s0[r], s1[r]));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64)) * sl2e;
        if (__any(mx > m_run + DEFER_THR)) {
            const float m_new = fmaxf(m_run, mx);
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            m_run = m_new;
            l_run *= alpha;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[i][r] *= alpha;
        }
        float psum = 0.f;
        float p0[16], p1[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            p0[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s0[r], sl2e, -m_run));
            p1[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s1[r], sl2e, -m_run));
            psum += p0[r] + p1[r];
        }
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            uint32_t w0[4], w1[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                w0[j] = cvt_pk_bf16(p0[kb * 8 + 2 * j], p0[kb * 8 + 2 * j + 1]);
                w1[j] = cvt_pk_bf16(p1[kb * 8 + 2 * j], p1[kb * 8 + 2 * j + 1]);
            }
            pf[kb] = *(bf8_t*)w0;
            pf[2 + kb] = *(bf8_t*)w1;
        }
        l_run += psum;
    };
    auto pv = [&](const uint8_t* sb) {
#pragma unroll
        for (int kb4 = 0; kb4 < 4; ++kb4) {
            const int sl = ((kb4 * 2 + half) ^ v_sw) << 4;
#pragma unroll
            for (int db = 0; db < 4; ++db) {
                const bf8_t vf = *(const bf8_t*)(sb + db * 32 * 128 + v_row_off + sl);
                o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[kb4], o[db], 0, 0, 0);
            }
        }
    };

    stage(0, 0);
    if (ntiles > 1) stage(1, 1);
    // Two copies of the tile loop (one per wave group) with IDENTICAL barrier counts: separate loops
    // keep each group's live ranges simple (one merged loop with per-interval branches spilled).
    if (!grpB) {
        int cur = 0, prev = NSTAGE - 1;             // buffers of tile t and tile t-1
        for (int t = 0; t < ntiles; ++t) {
            if (t + 1 < ntiles) wait_vm<2 * PW>(); else wait_vm<0>();  // tile t landed (t+1 may be in flight)
            __builtin_amdgcn_s_barrier();
            const uint8_t* sb = smem + cur * ATT_STAGE;
            qk(sb);                                                    // interval 1
            __builtin_amdgcn_s_barrier();
            if (t + 2 < ntiles) stage(t + 2, prev);                    // tile t-1's buffer is free now
            sm(t);                                                     // interval 2
            __builtin_amdgcn_s_barrier();
            pv(sb);                                                    // interval 3
            prev = cur;
            cur = (cur + 1 == NSTAGE) ? 0 : cur + 1;
        }
    } else {
        int cur = 0, prev = NSTAGE - 1;
        for (int t = 0; t < ntiles; ++t) {
            if (t + 1 < ntiles) wait_vm<2 * PW>(); else wait_vm<0>();
            __builtin_amdgcn_s_barrier();
            if (t > 0) pv(smem + prev * ATT_STAGE);                    // interval 1: PV(t-1)
            __builtin_amdgcn_s_barrier();
            if (t + 2 < ntiles) stage(t + 2, prev);
            qk(smem + cur * ATT_STAGE);                                // interval 2
            __builtin_amdgcn_s_barrier();
            sm(t);                                                     // interval 3
            prev = cur;
            cur = (cur + 1 == NSTAGE) ? 0 : cur + 1;
        }
        pv(smem + prev * ATT_STAGE);                                   // B's last PV
    }

    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    constexpr int OT_LD = 136;
    uint16_t* ot = (uint16_t*)smem;                    // 3 stages = 96 KiB >= 256 x 136 x 2
    __syncthreads();
    {
        uint16_t* orow = ot + (wave * 32 + ql) * OT_LD;
#pragma unroll
        for (int db = 0; db < 4; ++db)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const int d = db * 32 + 8 * r4 + 4 * half;
                const uint32_t w0 = cvt_pk_bf16(o[db][r4 * 4 + 0] * inv, o[db][r4 * 4 + 1] * inv);
                const uint32_t w1 = cvt_pk_bf16(o[db][r4 * 4 + 2] * inv, o[db][r4 * 4 + 3] * inv);
                *(uint2*)(orow + d) = make_uint2(w0, w1);
            }
    }
    __syncthreads();
    constexpr int RPP = (64 * NW) / 16;
#pragma unroll
    for (int it = 0; it < QB / RPP; ++it) {
        const int row = (tid >> 4) + it * RPP, c = (tid & 15) * 8;
        const int qr = qb * QB + row;
        if (qr < g.Sq) *(uint4*)(g.O + (size_t)qr * g.ldo + h * 128 + c) = *(const uint4*)(ot + row * OT_LD + c);
    }
}

template <int NW>
static int launch_attention_pp(const AttnArgs& g, hipStream_t st) {
    constexpr int QB = 32 * NW;
    const int nitems = g.H * ((g.Sq + QB - 1) / QB);
    constexpr int LDS = 3 * ATT_STAGE;
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)attention_pp_kernel<NW>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        attr = true;
    }
    hipLaunchKernelGGL((attention_pp_kernel<NW>), dim3(nitems), dim3(64 * NW), LDS, st, g);
    return check_launch("attention_pp_kernel");
}

template <int NW, int NSTAGE, int ABL = 0>
static int launch_attention(const AttnArgs& g, hipStream_t st) {
    constexpr int QB = 32 * NW;
    const int nitems = g.H * ((g.Sq + QB - 1) / QB);
    constexpr int LDS = NSTAGE * ATT_STAGE;
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)attention_kernel<NW, NSTAGE, ABL>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        attr = true;
    }
    hipLaunchKernelGGL((attention_kernel<NW, NSTAGE, ABL>), dim3(nitems), dim3(64 * NW), LDS, st, g);
    return check_launch("attention_kernel");
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
    hipStream_t st = (hipStream_t)stream;
    // 8-wave workgroups (256 query rows share each K/V tile) once they fill the chip; 4-wave otherwise
    const int items8 = H * ((Sq + 255) / 256);
    int variant = (items8 >= 256) ? 83 : 42;
    const char* v = getenv("RGN_ATTN_VARIANT");
    if (v && v[0] >= '0' && v[0] <= '9') variant = atoi(v);
    switch (variant) {
        case 42: return launch_attention<4, 2>(g, st);
        case 43: return launch_attention<4, 3>(g, st);
        case 82: return launch_attention<8, 2>(g, st);
        case 83: return launch_attention<8, 3>(g, st);
        case 84: return launch_attention<8, 4>(g, st);
        case 8: return launch_attention_pipe<8>(g, st);
        case 9: return launch_attention_pp<8>(g, st);
        case 821: return launch_attention<8, 2, 1>(g, st);
        case 822: return launch_attention<8, 2, 2>(g, st);
        case 823: return launch_attention<8, 2, 3>(g, st);
        case 824: return launch_attention<8, 2, 4>(g, st);
        case 825: return launch_attention<8, 2, 5>(g, st);
        case 4: return launch_attention_pipe<4>(g, st);
        default: return fail(RGN_E_BADARG, "attention: unknown RGN_ATTN_VARIANT");
    }
}

}  // extern "C"
