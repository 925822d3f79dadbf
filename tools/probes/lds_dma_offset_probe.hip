// Does the immediate `offset:` of `buffer_load_dword ... lds` move the LDS destination as well as the global source?
//   hipcc --offload-arch=gfx950 -O2 tools/probes/lds_dma_offset_probe.hip -o /tmp/lds_probe && /tmp/lds_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
__global__ void k(const uint32_t* src, uint32_t* out) {
    extern __shared__ uint32_t lds[];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = 0xdeadbeefu;
    __syncthreads();
    const uint64_t a = (uint64_t)src;
    u32x4 r;
    r[0] = __builtin_amdgcn_readfirstlane((uint32_t)a);
    r[1] = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32) & 0xffffu);
    r[2] = 0xffffffffu; r[3] = 0x00020000u;
    uint32_t voff = threadIdx.x * 4;
    uint32_t ldsbase = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)lds + 2048;   // bytes
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dword %0, %1, 0 offen offset:1024 lds\n\ts_waitcnt vmcnt(0)"
                 :: "v"(voff), "s"(r), "s"(ldsbase) : "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 4096; i += 64) out[i] = lds[i];
}
int main() {
    uint32_t *src, *out, h[4096], hs[4096];
    for (int i = 0; i < 4096; ++i) hs[i] = i;
    hipMalloc(&src, sizeof hs); hipMalloc(&out, sizeof h);
    hipMemcpy(src, hs, sizeof hs, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 16384, 0, src, out);
    hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost);
    for (int i = 0; i < 4096; ++i)
        if (h[i] != 0xdeadbeefu) { printf("first written LDS dword index %d (byte %d) holds src[%u]; M0 base was byte 2048, offset:1024, lane 0 voffset 0\n", i, i * 4, h[i]); break; }
    return 0;
}
