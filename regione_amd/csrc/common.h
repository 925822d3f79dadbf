// Shared helpers for libregione_hip.so (gfx950 only; wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/regione_hip.h"

namespace rgn {

extern thread_local char g_err[256];

inline int fail(int code, const char* msg) {
    snprintf(g_err, sizeof(g_err), "%s", msg);
    return code;
}

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
        return (int)e;
    }
    return 0;
}

// fp32 -> bf16, round-to-nearest-even like torch: the gfx950 hardware convert (v_cvt_pk_bf16_f32, two values per
// instruction).  Identical to the add-0x7fff bit trick for every finite value and infinity; a NaN stays a quiet
// NaN (torch canonicalises it to 0x7fc0 - no finite computation on this path produces one).
typedef __attribute__((ext_vector_type(2))) __bf16 hw_bf16x2;
typedef __attribute__((ext_vector_type(2))) float hw_f32x2;
__device__ __forceinline__ uint32_t f2bf_pk(float lo, float hi) {
    hw_bf16x2 r = __builtin_convertvector(hw_f32x2{lo, hi}, hw_bf16x2);
    return __builtin_bit_cast(uint32_t, r);
}
__device__ __forceinline__ uint16_t f2bf(float f) { return (uint16_t)f2bf_pk(f, 0.f); }
__device__ __forceinline__ float bf2f(uint16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
__device__ __forceinline__ float rbf(float f) { return __uint_as_float(f2bf_pk(f, 0.f) << 16); }   // round through bf16

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Launch-plan overrides: the library's ONE measurement / test hook (rgn_plan_override in include/regione_hip.h).  Every field is
// -1 in a shipped run (= the cost models decide); tests and the sweep tools force a schedule to reach paths a small problem would
// not take by itself, or to time a schedule the planner rejected.  Set through rgn_plan_override(key, value) or, once per process,
// RGN_PLAN_OVERRIDE="key=value,key=value" (parsed at the first launch: no getenv on the launch path).
struct PlanOverride {
    int gemm_pieces;      // K pieces of a round's remainder: 1 = one plain launch (no cut, no quarter tiles), n >= 2 = n pieces
    int gemm_geometry;    // 128 | 256: tile geometry
    int gemm_asm;         // 0 = compiler-scheduled kernels only (the fallback of operands >= 4 GiB / K < 128 / short fp8 K)
    int gemm_quarter;     // 0 = never, 1 = the remainder always as quarter tiles on the 128 geometry
    int attn_waves;       // 4 | 8 waves per workgroup
    int attn_split;       // 0 = never cut the KV range of remainder items
    int attn_streamk;     // 0 = equal KV pieces only, 1 = stream-K wherever it is possible
    int attn_asm;         // 0 = the compiler-scheduled KV loop (the ragged-KV fallback)
};
PlanOverride& plan_override();          // region.hip

typedef __attribute__((ext_vector_type(8))) short bf16x8;    // 8 bf16 = 4 VGPRs (MFMA A/B fragment)
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

}  // namespace rgn
