// Probe: how many bits of M0 does an LDS-DMA (global_load_lds / buffer_load ... lds) honour on gfx950?
// Writes 1 KiB of a known pattern to LDS byte offset `dst` and reports where it landed.
//   hipcc --offload-arch=gfx950 -O2 -o lds_dma_range tools/probes/lds_dma_range.hip && ./lds_dma_range
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

template <int MODE>
__global__ void probe(const uint32_t* src, uint32_t dst, uint32_t* found) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint32_t* l = (uint32_t*)smem;
    for (int i = threadIdx.x; i < 160 * 1024 / 4; i += 64) l[i] = 0;
    __syncthreads();
    if (MODE == 0) {
        __builtin_amdgcn_global_load_lds((glb_ptr_t)(src + threadIdx.x * 4), (lds_ptr_t)(smem + dst), 16, 0, 0);
    } else {
        const uint64_t a = (uint64_t)src;
        u32x4 r;
        r[0] = __builtin_amdgcn_readfirstlane((uint32_t)a);
        r[1] = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32) & 0xffffu);
        r[2] = 0xffffffffu; r[3] = 0x00020000u;
        uint32_t off = threadIdx.x * 16, d = __builtin_amdgcn_readfirstlane(dst), z = 0;
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %3 offen lds\n\t" ::"v"(off), "s"(r), "s"(d), "s"(z) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        int n = 0;
        for (uint32_t i = 0; i < 160 * 1024 / 4 && n < 4; i += 256)      // every 1 KiB boundary
            if (l[i] == 0xabc00000u) found[n++] = i * 4;
        found[4] = n;
    }
}

int main() {
    uint32_t *src, *found, h[256], hf[8];
    for (int i = 0; i < 256; ++i) h[i] = 0xabc00000u + i;
    hipMalloc(&src, 1024); hipMalloc(&found, 32);
    hipMemcpy(src, h, 1024, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void*)probe<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)probe<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const uint32_t dsts[] = {1024, 60 * 1024, 70 * 1024, 127 * 1024, 129 * 1024, 140 * 1024, 159 * 1024};
    for (int mode = 0; mode < 2; ++mode)
        for (uint32_t d : dsts) {
            hipMemset(found, 0xff, 32);
            if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(1), dim3(64), 160 * 1024, 0, src, d, found);
            else hipLaunchKernelGGL(probe<1>, dim3(1), dim3(64), 160 * 1024, 0, src, d, found);
            hipError_t e = hipDeviceSynchronize();
            hipMemcpy(hf, found, 32, hipMemcpyDeviceToHost);
            printf("%s dst=%6u (%3u KiB): err=%d hits=%u first landed at %d\n", mode ? "buffer_load_lds" : "global_load_lds", d, d / 1024,
                   (int)e, hf[4], hf[4] ? (int)hf[0] : -1);
        }
    return 0;
}
