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
// Two kernels share this design.  `attention_asm_kernel` (the default whenever Skv is a multiple of 64): 8 waves, the KV loop is
// ONE hand-scheduled asm statement (attn_loop_asm.inc, tools/gen_attn_loop.py) in which each wave overlaps its own phases -
// MFMA S(t+1) = K(t+1) Q^T under the exp2 / bf16 packing of P(t), MFMA O += V^T(t) P(t)^T under the row sums and the max of
// S(t+1) - over a five-stage LDS-DMA ring; <1> = equal KV split pieces, <2> = stream-K runs of the flattened (item, KV tile)
// steps for remainders (attention_schedule picks per launch; partials merged by attention_combine[_sk]_kernel).
// `attention_kernel` (compiler-scheduled, 4 or 8 waves): ragged KV lengths, tiny query sets.
#include "common.h"
#include <stdlib.h>
#include <utility>
#include "attn_loop_asm.inc"

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
    // round-aware launch: this launch covers items [item_offset, item_offset + nitems_launch); in
    // SPLIT mode every item is cut into nsplit KV ranges whose partial (O, m, l) go to `ws`.
    int item_offset, nitems_launch, nsplit;
    // static softmax shift (rgn_attention_bounded): the caller guarantees |q . k| * scale <= bound for every pair, so
    // P = exp2(S * c - static_m) cannot overflow or vanish and the loop keeps no running max (hand-scheduled kernel only)
    int static_on;
    float static_m;
    float* ws;
};

constexpr int KV_T = 64;                       // kv rows per tile
constexpr int K_TILE_BYTES = KV_T * 128 * 2;   // 16 KiB  [64 kv][128 d]
constexpr int V_TILE_BYTES = 128 * KV_T * 2;   // 16 KiB  [128 d][64 kv]
constexpr int ATT_STAGE = K_TILE_BYTES + V_TILE_BYTES;
constexpr float DEFER_THR = 8.0f;              // log2 units: P <= 2^8 before the running max is raised

__device__ __forceinline__ uint32_t cvt_pk_bf16(float a, float b) {
    bf2_t r = __builtin_convertvector(f32x2{a, b}, bf2_t);     // v_cvt_pk_bf16_f32 (RNE)
    return __builtin_bit_cast(uint32_t, r);
}

template <int N>
__device__ __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int NW, int NSTAGE, bool SPLIT>
__global__ __launch_bounds__(64 * NW, (NW >= 8) ? 1 : 2) void attention_kernel(const AttnArgs g) {
    constexpr int QB = 32 * NW;                 // query rows per workgroup
    constexpr int PW = 16 / NW;                 // K pieces (= V pieces) per wave per stage
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ql = lane & 31, half = lane >> 5;

    // ---- XCD-aware bijective item map: item = head * nQ + qblock ---------------------------------
    const int nQ = (g.Sq + QB - 1) / QB;
    int item, split = 0;
    {
        const int nb = SPLIT ? g.nitems_launch * g.nsplit : g.nitems_launch;
        const int bid = blockIdx.x, xcd = bid & 7, loc = bid >> 3;
        const int q = nb >> 3, r = nb & 7;
        int u = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
        if (SPLIT) { split = u % g.nsplit; u /= g.nsplit; }
        item = g.item_offset + u;
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

    const int ntiles_all = (g.Skv + KV_T - 1) / KV_T;
    int t_begin = 0, ntiles = ntiles_all;
    if (SPLIT) {
        const int per = (ntiles_all + g.nsplit - 1) / g.nsplit;
        t_begin = split * per;
        ntiles = max(0, min(per, ntiles_all - t_begin));
    }
#pragma unroll
    for (int s = 0; s < NSTAGE - 1; ++s)
        if (s < ntiles) stage(t_begin + s, s);
    int cur = 0, nxt = NSTAGE - 1;
    for (int t = 0; t < ntiles; ++t) {
        // tile t landed for this wave: leave the newer stages (at most NSTAGE-2) in flight
        const int newer = min(NSTAGE - 2, ntiles - 1 - t);
        if (NSTAGE >= 4 && newer == 2) wait_vm<2 * 2 * PW>();
        else if (NSTAGE >= 3 && newer == 1) wait_vm<2 * PW>();
        else wait_vm<0>();
        __builtin_amdgcn_s_barrier();
        if (t + NSTAGE - 1 < ntiles) stage(t_begin + t + NSTAGE - 1, nxt);
        const uint8_t* sb = smem + cur * ATT_STAGE;
        cur = (cur + 1 == NSTAGE) ? 0 : cur + 1;
        nxt = (nxt + 1 == NSTAGE) ? 0 : nxt + 1;

        // ---- S^T = K Q^T : 2 kv-blocks x 8 k-steps, the two accumulators interleaved --------------
        // All 16 K fragments are requested from LDS up front (64 VGPRs): left to itself hipcc issues each
        // ds_read one MFMA ahead and waits lgkmcnt(0) in front of it, exposing the LDS latency 16 times.
        f32x16 s0, s1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s0[r] = 0.f; s1[r] = 0.f; }
        {
            bf8_t kf[16];
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const int sl = ((ks * 2 + half) ^ k_sw) << 4;
                kf[2 * ks] = *(const bf8_t*)(sb + k_row_off + sl);
                kf[2 * ks + 1] = *(const bf8_t*)(sb + 32 * 256 + k_row_off + sl);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[2 * ks], qf[ks], s0, 0, 0, 0);
                s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[2 * ks + 1], qf[ks], s1, 0, 0, 0);
            }
        }
        // first half of the V^T fragments: requested BEFORE the softmax so their latency hides under its VALU
        bf8_t vf[16];
#pragma unroll
        for (int kb4 = 0; kb4 < 2; ++kb4)
#pragma unroll
            for (int db = 0; db < 4; ++db)
                vf[kb4 * 4 + db] = *(const bf8_t*)(sb + db * 32 * 128 + v_row_off + (((kb4 * 2 + half) ^ v_sw) << 4));
        __builtin_amdgcn_sched_barrier(0);
        // ---- tail mask ---------------------------------------------------------------------------
        const int kv0 = (t_begin + t) * KV_T;
        if (kv0 + KV_T > g.Skv) {                  // last tile only: keep it a real (scalar) branch -
            asm volatile("; kv tail");             // if-converted it costs ~140 VALU ops on EVERY tile
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int kv = kv0 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (kv >= g.Skv) s0[r] = -INFINITY;
                if (kv + 32 >= g.Skv) s1[r] = -INFINITY;
            }
        }
        // ---- online softmax with deferred max -------------------------------------------------------
        float mx = __builtin_fmaxf(s0[0], s1[0]);
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = __builtin_fmaxf(__builtin_fmaxf(mx, s0[r]), s1[r]);   // v_max3_f32
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
        bf8_t pf[4];
        {                                          // P overwrites S in place; packed fp32 math (v_pk_fma/add)
            const f32x2 sc2 = {sl2e, sl2e}, nm2 = {-m_run, -m_run};
            f32x2 ps2 = {0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                f32x2 a = {s0[2 * j], s0[2 * j + 1]}, b = {s1[2 * j], s1[2 * j + 1]};
                a = __builtin_elementwise_fma(a, sc2, nm2);
                b = __builtin_elementwise_fma(b, sc2, nm2);
                a[0] = __builtin_amdgcn_exp2f(a[0]); a[1] = __builtin_amdgcn_exp2f(a[1]);
                b[0] = __builtin_amdgcn_exp2f(b[0]); b[1] = __builtin_amdgcn_exp2f(b[1]);
                ps2 += a + b;
                s0[2 * j] = a[0]; s0[2 * j + 1] = a[1];
                s1[2 * j] = b[0]; s1[2 * j + 1] = b[1];
            }
            l_run += ps2[0] + ps2[1];
        }
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            uint32_t w0[4], w1[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                w0[j] = cvt_pk_bf16(s0[kb * 8 + 2 * j], s0[kb * 8 + 2 * j + 1]);
                w1[j] = cvt_pk_bf16(s1[kb * 8 + 2 * j], s1[kb * 8 + 2 * j + 1]);
            }
            pf[kb] = *(bf8_t*)w0;
            pf[2 + kb] = *(bf8_t*)w1;
        }
        // second half of the V^T fragments, then O^T += V^T P^T : 4 k-blocks x 4 d-blocks
#pragma unroll
        for (int kb4 = 2; kb4 < 4; ++kb4)
#pragma unroll
            for (int db = 0; db < 4; ++db)
                vf[kb4 * 4 + db] = *(const bf8_t*)(sb + db * 32 * 128 + v_row_off + (((kb4 * 2 + half) ^ v_sw) << 4));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kb4 = 0; kb4 < 4; ++kb4)
#pragma unroll
            for (int db = 0; db < 4; ++db)
                o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[kb4 * 4 + db], pf[kb4], o[db], 0, 0, 0);
    }

    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    if (SPLIT) {
        // partial result of this KV range: unnormalised O (fp32), running max (scaled-log2 units), row sum
        float* base = g.ws + ((size_t)(item - g.item_offset) * g.nsplit + split) * (size_t)(QB * 130);
        float* orow = base + (size_t)(wave * 32 + ql) * 128;
#pragma unroll
        for (int db = 0; db < 4; ++db)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4)
                *(float4*)(orow + db * 32 + 8 * r4 + 4 * half) =
                    make_float4(o[db][r4 * 4 + 0], o[db][r4 * 4 + 1], o[db][r4 * 4 + 2], o[db][r4 * 4 + 3]);
        if (half == 0) {
            base[QB * 128 + wave * 32 + ql] = m_run;
            base[QB * 129 + wave * 32 + ql] = l_tot;
        }
        return;
    }
    // ---- finalize: O = O^T / l, staged through LDS for 16-byte row-contiguous stores ----------------
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

// ------------------------------------------------------------------------------------------------------------------
// Hand-scheduled variant (tools/gen_attn_loop.py -> attn_loop_asm.inc): same data layout, same LDS images, same results
// up to fp32 summation order; the KV loop is ONE asm statement that overlaps, inside every wave, the MFMAs of one tile with
// the softmax VALU of its neighbours.  8 waves x 32 query rows, five 32 KiB stages (all 160 KiB of LDS), Skv % 64 == 0.
// ------------------------------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_s;

template <int B>
__device__ __forceinline__ float agpr_read1() {
    float x;
    asm volatile("v_accvgpr_read_b32 %0, a[%1]" : "=v"(x) : "n"(B));
    return x;
}
template <class F, int... Is>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Is...>) {
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// SPLIT: 0 = one workgroup per item (256 query rows of one head against every KV tile), result written in place;
//        1 = every item cut into g.nsplit equal KV ranges (partials to g.ws, attention_combine_kernel merges);
//        2 = stream-K: the (item, KV tile) steps of the launch, flattened item-major, are dealt out in equal contiguous
//            runs to the gridDim.x workgroups; a run crosses at most one item boundary, so a workgroup works through one
//            or two SEGMENTS (partials to slot 2 * unit + seg, attention_combine_sk_kernel merges).
template <int SPLIT, bool SM = false>
__global__ __launch_bounds__(512, 1) void attention_asm_kernel(const AttnArgs g) {
    constexpr int NW = 8, QB = 256;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nQ = (g.Sq + QB - 1) / QB;
    const int ntiles_all = g.Skv / KV_T;
    int unit, split = 0;
    {
        const int nb = SPLIT == 2 ? (int)gridDim.x : SPLIT == 1 ? g.nitems_launch * g.nsplit : g.nitems_launch;
        const int bid = blockIdx.x, xcd = bid & 7, loc = bid >> 3;
        const int q = nb >> 3, r = nb & 7;
        unit = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    int sk_lo = 0, sk_hi = 0, seg = 0;
    if (SPLIT == 2) {
        const long long total = (long long)g.nitems_launch * ntiles_all;
        sk_lo = (int)((long long)unit * total / (int)gridDim.x);
        sk_hi = (int)((long long)(unit + 1) * total / (int)gridDim.x);
        if (sk_lo >= sk_hi) return;
    }
    const uint32_t HD2 = (uint32_t)g.H * 256u;                 // bytes of one K row (H * 128 bf16)
#pragma nounroll
  for (;;) {
    // the lane id is made opaque per segment: otherwise the compiler hoists the ~20 lane-dependent address registers out of the
    // segment loop and, with v32-v159 / a0-a95 pinned by the asm block, parks them in a96+ - past the 256 registers a wave may own
    int lane = tid & 63;
    if (SPLIT == 2) asm volatile("" : "+v"(lane));
    const int ql = lane & 31, half = lane >> 5;
    int item, t_begin = 0, ntiles = ntiles_all;
    if (SPLIT == 2) {
        const int il = sk_lo / ntiles_all;
        item = g.item_offset + il;
        t_begin = sk_lo - il * ntiles_all;
        ntiles = min(ntiles_all - t_begin, sk_hi - sk_lo);
    } else if (SPLIT == 1) {
        split = unit % g.nsplit;
        item = g.item_offset + unit / g.nsplit;
        const int per = (ntiles_all + g.nsplit - 1) / g.nsplit;
        t_begin = split * per;
        ntiles = max(0, min(per, ntiles_all - t_begin));
    } else {
        item = g.item_offset + unit;
    }
    const int h = item / nQ, qb = item - h * nQ;
    const int q0 = qb * QB + wave * 32;
    float m_run = SM ? g.static_m : -1e30f, l_run = 0.f;
    if (ntiles > 0) {
        // per-lane relative LDS addresses of the fragments (same swizzles as attention_kernel)
        uint32_t krel[8], vrel[4];
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) krel[ks] = ql * 256 + (((ks * 2 + half) ^ (ql & 15)) << 4);
#pragma unroll
        for (int kb4 = 0; kb4 < 4; ++kb4) vrel[kb4] = ql * 128 + (((kb4 * 2 + half) ^ ((ql >> 1) & 7)) << 4);
        // per-lane source offsets of this wave's 2 + 2 DMA pieces (bytes from the slab bases)
        uint32_t dk[2], dv[2];
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int piece = wave * 2 + p;
            const int krow = piece * 4 + (lane >> 4);
            dk[p] = (uint32_t)krow * HD2 + h * 256 + (((lane & 15) ^ (krow & 15)) << 4);
            const int vrow = piece * 8 + (lane >> 3);
            dv[p] = (uint32_t)(h * 128 + vrow) * (uint32_t)(g.skv_pad * 2) + (((lane & 7) ^ ((vrow >> 1) & 7)) << 4);
        }
        const int qr = min(q0 + ql, g.Sq - 1);
        const uint16_t* qptr = g.Q + (size_t)qr * g.ldq + h * 128 + half * 8;
        auto rsrc = [](const void* p) {
            const uint64_t a = (uint64_t)p;
            u32x4_s r;
            r[0] = __builtin_amdgcn_readfirstlane((uint32_t)a);
            r[1] = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32) & 0xffffu);
            r[2] = 0xffffffffu;
            r[3] = 0x00020000u;
            return r;
        };
        const u32x4_s rk = rsrc(g.K), rv = rsrc(g.Vt);
        const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)smem;    // 0: no static LDS
        const uint32_t kadv = __builtin_amdgcn_readfirstlane((uint32_t)KV_T * HD2);
        uint32_t tk = __builtin_amdgcn_readfirstlane((uint32_t)t_begin * kadv), tv = __builtin_amdgcn_readfirstlane((uint32_t)t_begin * 128u);
        const uint32_t tk_last = __builtin_amdgcn_readfirstlane((uint32_t)(t_begin + ntiles - 1) * kadv);
        const uint32_t tv_last = __builtin_amdgcn_readfirstlane((uint32_t)(t_begin + ntiles - 1) * 128u);
        uint32_t stg_k = __builtin_amdgcn_readfirstlane(lds0 + 32768u), stg_v = __builtin_amdgcn_readfirstlane(lds0), stg_d = stg_v;
        const uint32_t wdst = __builtin_amdgcn_readfirstlane((uint32_t)wave * 2048u);
        uint32_t cnt = __builtin_amdgcn_readfirstlane((uint32_t)(ntiles - 1) >> 1), rem = __builtin_amdgcn_readfirstlane((uint32_t)(ntiles - 1) & 1u);
        // every wave in role A: the two-role (dephased) loop of tools/gen_attn_loop.py (ATTN_DEPHASE=1 builds) measured +-0.5 %
        // and is not shipped (profiles/EXPERIMENTS.md 4.7); the operand stays so that such a build needs no other source change
        const uint32_t role = __builtin_amdgcn_readfirstlane(0u);
        const float sl2e = g.scale_log2e;
        const uint64_t sl2e2 = ((uint64_t)__float_as_uint(sl2e) << 32) | __float_as_uint(sl2e);    // both halves: packed-fp32 operand
        uint32_t stmp, stmp2, sdst;
        if constexpr (SM) {
            asm volatile(RGN_ATTN_LOOP_SM_ASM
                         : [m_run] "+&v"(m_run), [l_run] "+&v"(l_run), [tk] "+&s"(tk), [tv] "+&s"(tv), [stg_k] "+&s"(stg_k),
                           [stg_v] "+&s"(stg_v), [stg_d] "+&s"(stg_d), [cnt] "+&s"(cnt), [stmp] "=&s"(stmp), [stmp2] "=&s"(stmp2),
                           [sdst] "=&s"(sdst)
                         : [krel0] "v"(krel[0]), [krel1] "v"(krel[1]), [krel2] "v"(krel[2]), [krel3] "v"(krel[3]), [krel4] "v"(krel[4]),
                           [krel5] "v"(krel[5]), [krel6] "v"(krel[6]), [krel7] "v"(krel[7]), [vrel0] "v"(vrel[0]), [vrel1] "v"(vrel[1]),
                           [vrel2] "v"(vrel[2]), [vrel3] "v"(vrel[3]), [dk0] "v"(dk[0]), [dk1] "v"(dk[1]), [dv0] "v"(dv[0]), [dv1] "v"(dv[1]),
                           [qptr] "v"(qptr), [rk] "s"(rk), [rv] "s"(rv), [kadv] "s"(kadv), [tk_last] "s"(tk_last), [tv_last] "s"(tv_last),
                           [wdst] "s"(wdst), [rem] "s"(rem), [sl2e] "s"(sl2e), [sl2e2] "s"(sl2e2), [ones2] "s"(0x3f803f80u), [role] "s"(role)
                         : RGN_ATTN_LOOP_CLOBBERS);
        } else {
            asm volatile(RGN_ATTN_LOOP_ASM
                         : [m_run] "+&v"(m_run), [l_run] "+&v"(l_run), [tk] "+&s"(tk), [tv] "+&s"(tv), [stg_k] "+&s"(stg_k),
                           [stg_v] "+&s"(stg_v), [stg_d] "+&s"(stg_d), [cnt] "+&s"(cnt), [stmp] "=&s"(stmp), [stmp2] "=&s"(stmp2),
                           [sdst] "=&s"(sdst)
                         : [krel0] "v"(krel[0]), [krel1] "v"(krel[1]), [krel2] "v"(krel[2]), [krel3] "v"(krel[3]), [krel4] "v"(krel[4]),
                           [krel5] "v"(krel[5]), [krel6] "v"(krel[6]), [krel7] "v"(krel[7]), [vrel0] "v"(vrel[0]), [vrel1] "v"(vrel[1]),
                           [vrel2] "v"(vrel[2]), [vrel3] "v"(vrel[3]), [dk0] "v"(dk[0]), [dk1] "v"(dk[1]), [dv0] "v"(dv[0]), [dv1] "v"(dv[1]),
                           [qptr] "v"(qptr), [rk] "s"(rk), [rv] "s"(rv), [kadv] "s"(kadv), [tk_last] "s"(tk_last), [tv_last] "s"(tv_last),
                           [wdst] "s"(wdst), [rem] "s"(rem), [sl2e] "s"(sl2e), [sl2e2] "s"(sl2e2), [ones2] "s"(0x3f803f80u), [role] "s"(role)
                         : RGN_ATTN_LOOP_CLOBBERS);
        }
    }
    // O^T accumulator: a[db * 16 + r] (zero when this split piece had no tiles - but then the launch has none either)
    const float l_other = __shfl_xor(l_run, 32, 64);
    const float l_tot = l_run + l_other;
    if (SPLIT) {
        const size_t slot = SPLIT == 2 ? (size_t)unit * 2 + seg : (size_t)(item - g.item_offset) * g.nsplit + split;
        float* base = g.ws + slot * (size_t)(QB * 130);
        float* orow = base + (size_t)(wave * 32 + ql) * 128;
        static_for<16>([&](auto Ic) {
            constexpr int I = decltype(Ic)::value;           // I = db * 4 + r4
            constexpr int db = I / 4, r4 = I % 4;
            *(float4*)(orow + db * 32 + 8 * r4 + 4 * half) =
                make_float4(agpr_read1<db * 16 + r4 * 4 + 0>(), agpr_read1<db * 16 + r4 * 4 + 1>(),
                            agpr_read1<db * 16 + r4 * 4 + 2>(), agpr_read1<db * 16 + r4 * 4 + 3>());
        });
        if (half == 0) {
            base[QB * 128 + wave * 32 + ql] = m_run;
            base[QB * 129 + wave * 32 + ql] = l_tot;
        }
        if (SPLIT != 2) return;
        sk_lo += ntiles;
        ++seg;
        if (sk_lo >= sk_hi) return;
        __syncthreads();                    // every wave is out of the K/V ring before the next segment's prologue refills it
        continue;
    }
    const float inv = 1.0f / l_tot;
    constexpr int OT_LD = 136;
    uint16_t* ot = (uint16_t*)smem;                      // 256 rows x 136 bf16 = 68 KiB: fits, one pass
    {
        uint16_t* orow = ot + (wave * 32 + ql) * OT_LD;
        static_for<16>([&](auto Ic) {
            constexpr int I = decltype(Ic)::value;
            constexpr int db = I / 4, r4 = I % 4;
            const int d = db * 32 + 8 * r4 + 4 * half;
            const uint32_t w0 = cvt_pk_bf16(agpr_read1<db * 16 + r4 * 4 + 0>() * inv, agpr_read1<db * 16 + r4 * 4 + 1>() * inv);
            const uint32_t w1 = cvt_pk_bf16(agpr_read1<db * 16 + r4 * 4 + 2>() * inv, agpr_read1<db * 16 + r4 * 4 + 3>() * inv);
            *(uint2*)(orow + d) = make_uint2(w0, w1);
        });
    }
    __syncthreads();
    constexpr int RPP = 512 / 16;
#pragma unroll
    for (int it = 0; it < QB / RPP; ++it) {
        const int row = (tid >> 4) + it * RPP, c = (tid & 15) * 8;
        const int qr2 = qb * QB + row;
        if (qr2 < g.Sq) *(uint4*)(g.O + (size_t)qr2 * g.ldo + h * 128 + c) = *(const uint4*)(ot + row * OT_LD + c);
    }
    return;
  }
}

// Merge the nsplit partial results of each split item: O = sum_s O_s 2^(m_s - m*) / sum_s l_s 2^(m_s - m*).
template <int QB>
__global__ __launch_bounds__(256) void attention_combine_kernel(const AttnArgs g) {
    const int nQ = (g.Sq + QB - 1) / QB;
    const int u = blockIdx.x / (QB / 16);                       // split item index
    const int row = (blockIdx.x % (QB / 16)) * 16 + (threadIdx.x >> 4);
    const int c = (threadIdx.x & 15) * 8;
    const int item = g.item_offset + u;
    const int h = item / nQ, qb = item - h * nQ;
    const int qr = qb * QB + row;
    if (qr >= g.Sq) return;
    const float* base = g.ws + (size_t)u * g.nsplit * (size_t)(QB * 130);
    float mstar = -1e30f;
    for (int s = 0; s < g.nsplit; ++s) mstar = fmaxf(mstar, base[(size_t)s * QB * 130 + QB * 128 + row]);
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float l = 0.f;
    for (int s = 0; s < g.nsplit; ++s) {
        const float* b = base + (size_t)s * QB * 130;
        const float w = __builtin_amdgcn_exp2f(b[QB * 128 + row] - mstar);
        l += b[QB * 129 + row] * w;
        const float4 x0 = *(const float4*)(b + (size_t)row * 128 + c), x1 = *(const float4*)(b + (size_t)row * 128 + c + 4);
        acc[0] += x0.x * w; acc[1] += x0.y * w; acc[2] += x0.z * w; acc[3] += x0.w * w;
        acc[4] += x1.x * w; acc[5] += x1.y * w; acc[6] += x1.z * w; acc[7] += x1.w * w;
    }
    const float inv = 1.0f / l;
    uint4 out;
    out.x = cvt_pk_bf16(acc[0] * inv, acc[1] * inv); out.y = cvt_pk_bf16(acc[2] * inv, acc[3] * inv);
    out.z = cvt_pk_bf16(acc[4] * inv, acc[5] * inv); out.w = cvt_pk_bf16(acc[6] * inv, acc[7] * inv);
    *(uint4*)(g.O + (size_t)qr * g.ldo + h * 128 + c) = out;
}

// Stream-K merge: item u of the launch owns the flat steps [u * nt, (u + 1) * nt); run w of the G runs covers
// [w * total / G, (w + 1) * total / G) and left its (first, second) segment in slots (2w, 2w + 1).
template <int QB>
__global__ __launch_bounds__(256) void attention_combine_sk_kernel(const AttnArgs g, int G) {
    const int nQ = (g.Sq + QB - 1) / QB;
    const int u = blockIdx.x / (QB / 16);
    const int row = (blockIdx.x % (QB / 16)) * 16 + (threadIdx.x >> 4);
    const int c = (threadIdx.x & 15) * 8;
    const int item = g.item_offset + u;
    const int h = item / nQ, qb = item - h * nQ;
    const int qr = qb * QB + row;
    if (qr >= g.Sq) return;
    const int nt = g.Skv / KV_T;
    const long long total = (long long)g.nitems_launch * nt;
    const long long lo = (long long)u * nt, hi = lo + nt;
    auto bound = [&](int w) { return (long long)w * total / G; };
    int w0 = (int)(lo * G / total);
    while (w0 + 1 < G && bound(w0 + 1) <= lo) ++w0;
    while (w0 > 0 && bound(w0) > lo) --w0;
    float mstar = -1e30f;
    for (int w = w0; w < G && bound(w) < hi; ++w) {
        if (bound(w + 1) <= bound(w)) continue;                                  // empty run
        const int sg = ((int)(bound(w) / nt) == u) ? 0 : 1;
        mstar = fmaxf(mstar, g.ws[((size_t)w * 2 + sg) * (size_t)(QB * 130) + QB * 128 + row]);
    }
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float l = 0.f;
    for (int w = w0; w < G && bound(w) < hi; ++w) {
        if (bound(w + 1) <= bound(w)) continue;
        const int sg = ((int)(bound(w) / nt) == u) ? 0 : 1;
        const float* b = g.ws + ((size_t)w * 2 + sg) * (size_t)(QB * 130);
        const float wt = __builtin_amdgcn_exp2f(b[QB * 128 + row] - mstar);
        l += b[QB * 129 + row] * wt;
        const float4 x0 = *(const float4*)(b + (size_t)row * 128 + c), x1 = *(const float4*)(b + (size_t)row * 128 + c + 4);
        acc[0] += x0.x * wt; acc[1] += x0.y * wt; acc[2] += x0.z * wt; acc[3] += x0.w * wt;
        acc[4] += x1.x * wt; acc[5] += x1.y * wt; acc[6] += x1.z * wt; acc[7] += x1.w * wt;
    }
    const float inv = 1.0f / l;
    uint4 out;
    out.x = cvt_pk_bf16(acc[0] * inv, acc[1] * inv); out.y = cvt_pk_bf16(acc[2] * inv, acc[3] * inv);
    out.z = cvt_pk_bf16(acc[4] * inv, acc[5] * inv); out.w = cvt_pk_bf16(acc[6] * inv, acc[7] * inv);
    *(uint4*)(g.O + (size_t)qr * g.ldo + h * 128 + c) = out;
}

// nblocks: SPLIT 0 -> items, 1 -> items x nsplit, 2 -> the number of stream-K runs
template <int SPLIT>
static int launch_attention_asm(const AttnArgs& g, int nblocks, hipStream_t st) {
    constexpr int LDS = 5 * ATT_STAGE;                   // 160 KiB
    static bool attr[64] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64 || !attr[dev]) {
        (void)hipFuncSetAttribute((const void*)attention_asm_kernel<SPLIT, false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        (void)hipFuncSetAttribute((const void*)attention_asm_kernel<SPLIT, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (dev >= 0 && dev < 64) attr[dev] = true;
    }
    if (nblocks == 0) return 0;
    if (g.static_on) hipLaunchKernelGGL((attention_asm_kernel<SPLIT, true>), dim3(nblocks), dim3(512), LDS, st, g);
    else hipLaunchKernelGGL((attention_asm_kernel<SPLIT, false>), dim3(nblocks), dim3(512), LDS, st, g);
    return check_launch("attention_asm_kernel");
}

// The asm loop needs whole KV tiles and 32-bit slab offsets (plan_override().attn_asm = 0: the compiler-scheduled loop everywhere - tests).
static bool attention_use_asm(const AttnArgs& g) {
    return plan_override().attn_asm != 0 && (g.Skv % KV_T) == 0 && (size_t)g.skv_pad * g.H * 256 < ((size_t)1 << 32);
}

template <int NW, int NSTAGE, bool SPLIT>
static int launch_attention(const AttnArgs& g, hipStream_t st) {
    if (NW == 8 && attention_use_asm(g))
        return launch_attention_asm<SPLIT ? 1 : 0>(g, SPLIT ? g.nitems_launch * g.nsplit : g.nitems_launch, st);
    constexpr int LDS = NSTAGE * ATT_STAGE;
    static bool attr[64] = {};          // per device (one process may drive several GPUs)
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64 || !attr[dev]) {
        (void)hipFuncSetAttribute((const void*)attention_kernel<NW, NSTAGE, SPLIT>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (dev >= 0 && dev < 64) attr[dev] = true;
    }
    const int nblocks = SPLIT ? g.nitems_launch * g.nsplit : g.nitems_launch;
    if (nblocks == 0) return 0;
    hipLaunchKernelGGL((attention_kernel<NW, NSTAGE, SPLIT>), dim3(nblocks), dim3(64 * NW), LDS, st, g);
    return check_launch("attention_kernel");
}

// Round-aware schedule: items that fill whole rounds of the chip's workgroup slots run as they are;
// the REMAINDER (which would otherwise occupy a full round at partial occupancy - 816 items on 256
// CUs = 3.19 rounds -> 4) is cut along KV into `nsplit` ranges so that it spreads over all CUs, and
// merged by a tiny combine kernel.
static thread_local int g_last_attn_plan = 0;     // what attention_schedule chose last on this thread: bits 0-3 = equal KV pieces of the
                                                   // remainder (1 = none), bit 4 = stream-K remainder, bit 5 = 8-wave workgroups

template <int NW, int NSTAGE>
static int attention_schedule(AttnArgs g, int slots, void* ws, size_t ws_bytes, hipStream_t st, bool dry = false) {
    constexpr int QB = 32 * NW;
    const int nitems = g.H * ((g.Sq + QB - 1) / QB);
    const int ntiles = (g.Skv + KV_T - 1) / KV_T;
    int full = (nitems / slots) * slots, left = nitems - full, best = 1;
    // Cost of the remainder launch in microseconds (8-wave kernels, head_dim 128, MI355X; tools/bench_kernels.py attn):
    // a workgroup spends c_t per 256 x 64 KV tile and c_p per PIECE it starts (Q load, ring fill, fp32 partial dump, workgroup
    // turn-over on a CU whose LDS one workgroup fills), the merge pass c_m.  Fits the measured region-step shapes within 5 %.
    const float c_t = NW == 8 ? 1.69f : 0.95f, c_p = 10.7f, c_m = 8.0f;
    float best_cost = (float)((left + slots - 1) / slots) * ((float)ntiles * c_t + c_p);
    // a launch that is ONE partially filled round of 8-wave workgroups runs its KV loop faster than a full chip does (round 4,
    // tools/probes/attn_plan_sweep.py, no-split launches at Skv 8704 / 2560: 1.10 us per tile with 72 items, 1.19-1.25 with 144, 1.35 with
    // 192, 1.58 with 240 - fewer CUs share the power budget): the model priced 144 items x 40 tiles at 78 us, split them (70 us) and lost
    // to the plain launch (56 us); 192 items x 136 tiles: stream-K 204 us against 194.5 us plain
    if (NW == 8 && full == 0 && left > 0) {
        const float x = fmaxf(0.0f, (float)(left - 72) / 184.0f);
        best_cost = (float)ntiles * (1.10f + 0.59f * x * x) + c_p;
    }
    bool stream_k = false;
    if (left > 0 && ws != nullptr) {
        for (int S = 2; S <= 8; ++S) {
            if (ntiles / S < 4) break;
            if ((size_t)left * S * QB * 130 * sizeof(float) > ws_bytes) break;
            const float cost = (float)((left * S + slots - 1) / slots) * ((float)((ntiles + S - 1) / S) * c_t + c_p) + c_m;
            if (cost < best_cost - 1e-3f) { best_cost = cost; best = S; }
        }
        // stream-K (asm kernel): `slots` equal runs of the flattened (item, KV tile) steps instead of S equal pieces per item -
        // every CU gets the same number of steps whatever left / slots is, in ONE round, and the launch leaves at most
        // left + slots partials instead of left x S; a run that crosses an item boundary pays the piece cost twice.
        // plan_override().attn_streamk: 0 = equal split only, 1 = stream-K wherever it is possible (tests, sweeps).
        const int sk_o = plan_override().attn_streamk;
        const long long steps = (long long)left * ntiles;
        if (NW == 8 && sk_o != 0 && attention_use_asm(g) && left < slots && steps >= 8LL * slots &&
            (size_t)slots * 2 * QB * 130 * sizeof(float) <= ws_bytes) {
            const float cost = (float)((steps + slots - 1) / slots) * c_t + 2.0f * c_p + c_m;
            if (cost < best_cost - 1e-3f || sk_o == 1) { best_cost = cost; stream_k = true; }   // 1: forced (tests)
        }
    }
    g_last_attn_plan = (stream_k ? 1 : best) | (stream_k ? 0x10 : 0) | (NW == 8 ? 0x20 : 0);
    if (dry) return 0;                                  // rgn_attention_plan_query: the plan only
    g.ws = (float*)ws;
    g.nsplit = 1;
    int rc = 0;
    if (stream_k) {
        if (full > 0) {
            g.item_offset = 0; g.nitems_launch = full;
            if ((rc = launch_attention<NW, NSTAGE, false>(g, st))) return rc;
        }
        g.item_offset = full; g.nitems_launch = left;
        if ((rc = launch_attention_asm<2>(g, slots, st))) return rc;
        hipLaunchKernelGGL((attention_combine_sk_kernel<QB>), dim3(left * (QB / 16)), dim3(256), 0, st, g, slots);
        return check_launch("attention_combine_sk_kernel");
    }
    if (best == 1) {
        g.item_offset = 0; g.nitems_launch = nitems;
        return launch_attention<NW, NSTAGE, false>(g, st);
    }
    if (full > 0) {
        g.item_offset = 0; g.nitems_launch = full;
        if ((rc = launch_attention<NW, NSTAGE, false>(g, st))) return rc;
    }
    g.item_offset = full; g.nitems_launch = left; g.nsplit = best;
    if ((rc = launch_attention<NW, NSTAGE, true>(g, st))) return rc;
    hipLaunchKernelGGL((attention_combine_kernel<QB>), dim3(left * (QB / 16)), dim3(256), 0, st, g);
    return check_launch("attention_combine_kernel");
}

}  // namespace rgn

using namespace rgn;

extern "C" {

size_t rgn_attention_workspace_bytes(int Sq, int H) {
    (void)Sq; (void)H;
    return (size_t)128 << 20;        // 128 MiB covers every (remainder x split) choice attention_schedule makes
}

int rgn_attention(const void* Q, int ldq, const void* k_slab, const void* vt_slab, int skv_pad, void* O, int ldo,
                  int Sq, int Skv, int H, float scale, void* workspace, size_t workspace_bytes, void* stream) {
    return rgn_attention_bounded(Q, ldq, k_slab, vt_slab, skv_pad, O, ldo, Sq, Skv, H, scale, 0.0f, workspace, workspace_bytes, stream);
}

int rgn_attention_bounded(const void* Q, int ldq, const void* k_slab, const void* vt_slab, int skv_pad, void* O, int ldo,
                          int Sq, int Skv, int H, float scale, float score_bound, void* workspace, size_t workspace_bytes,
                          void* stream) {
    if (Sq == 0) return 0;
    if (!Q || !k_slab || !vt_slab || !O || Sq < 0 || Skv <= 0 || H <= 0 || (skv_pad % 64) || skv_pad < Skv ||
        (ldq % 8) || (ldo % 8))
        return fail(RGN_E_BADARG, "attention: bad argument");
    AttnArgs g;
    g.Q = (const uint16_t*)Q; g.K = (const uint16_t*)k_slab; g.Vt = (const uint16_t*)vt_slab; g.O = (uint16_t*)O;
    g.ldq = ldq; g.ldo = ldo; g.skv_pad = skv_pad; g.Sq = Sq; g.Skv = Skv; g.H = H;
    g.scale_log2e = scale * 1.4426950408889634f;
    g.item_offset = 0; g.nitems_launch = 0; g.nsplit = 1; g.ws = nullptr;
    // static softmax shift: with |scores| <= score_bound, exp2(s * log2e) stays within 2^+-96 for score_bound * log2e <= 96 - row
    // sums over any Skv < 2^31 and the PV accumulators stay far inside fp32 (and bf16 for P).  score_bound = 0: the running-max loop.
    {
        const float b2 = score_bound * 1.4426950408889634f;
        g.static_on = (score_bound > 0.0f && b2 <= 96.0f) ? 1 : 0;
        g.static_m = 0.0f;
    }
    hipStream_t st = (hipStream_t)stream;
    // 8-wave workgroups (256 query rows share each K/V tile, 1 per CU) unless the query set is tiny
    int variant = (H * ((Sq + 255) / 256) >= 96) ? 8 : 4;
    const PlanOverride& ov = plan_override();
    if (ov.attn_waves == 4 || ov.attn_waves == 8) variant = ov.attn_waves;
    if (ov.attn_split == 0) { workspace = nullptr; workspace_bytes = 0; }        // no KV split
    if (variant == 8) return attention_schedule<8, 3>(g, 256, workspace, workspace_bytes, st);
    return attention_schedule<4, 2>(g, 512, workspace, workspace_bytes, st);
}

int rgn_attention_last_plan(void) { return g_last_attn_plan; }

int rgn_attention_plan_query(int Sq, int Skv, int H, size_t workspace_bytes) {
    if (Sq <= 0 || Skv <= 0 || H <= 0) return fail(RGN_E_BADARG, "attention_plan_query: bad argument");
    AttnArgs g{};
    g.Sq = Sq; g.Skv = Skv; g.H = H; g.skv_pad = (Skv + 63) / 64 * 64;
    int variant = (H * ((Sq + 255) / 256) >= 96) ? 8 : 4;
    const PlanOverride& ov = plan_override();
    if (ov.attn_waves == 4 || ov.attn_waves == 8) variant = ov.attn_waves;
    void* ws = (workspace_bytes > 0 && ov.attn_split != 0) ? (void*)(uintptr_t)64 : nullptr;      // never dereferenced in a dry run
    if (variant == 8) (void)attention_schedule<8, 3>(g, 256, ws, ws ? workspace_bytes : 0, nullptr, true);
    else (void)attention_schedule<4, 2>(g, 512, ws, ws ? workspace_bytes : 0, nullptr, true);
    return g_last_attn_plan;
}

}  // extern "C"
