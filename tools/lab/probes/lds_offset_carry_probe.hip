// Does a DS instruction's 16-bit immediate offset carry into address bit 16 on gfx950 (160 KB of LDS)?
// Fills LDS with its own dword index, reads ds_read_b128 at vaddr = lane * 16 with offset:65104 (lanes >= 27 cross 64 KB) and at the
// same addresses computed in the VGPR, and prints the lanes where the two differ.  hipcc --offload-arch=gfx950 -O2 ... && ./a.out
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;
__global__ void probe(uint32_t* out) {
    extern __shared__ uint32_t lds[];
    for (int i = threadIdx.x; i < 160 * 1024 / 4; i += blockDim.x) lds[i] = i;
    __syncthreads();
    if (threadIdx.x < 64) {
        const uint32_t a = threadIdx.x * 16;
        u32x4 imm, full, imm2, full2;
        asm volatile("ds_read_b128 %0, %1 offset:65104\n\ts_waitcnt lgkmcnt(0)" : "=v"(imm) : "v"(a) : "memory");
        const uint32_t b = a + 65104;
        asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(full) : "v"(b) : "memory");
        const uint32_t c = a + 130000;        // crossing 128 KB with the immediate
        asm volatile("ds_read_b128 %0, %1 offset:2000\n\ts_waitcnt lgkmcnt(0)" : "=v"(imm2) : "v"(c) : "memory");
        const uint32_t d = c + 2000;
        asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(full2) : "v"(d) : "memory");
        out[threadIdx.x * 4 + 0] = imm.x; out[threadIdx.x * 4 + 1] = full.x; out[threadIdx.x * 4 + 2] = imm2.x; out[threadIdx.x * 4 + 3] = full2.x;
    }
}
int main() {
    uint32_t* d; hipMalloc(&d, 64 * 16);
    hipFuncSetAttribute(reinterpret_cast<const void*>(probe), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(probe, dim3(1), dim3(256), 160 * 1024, 0, d);
    uint32_t h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        if (h[l * 4] != h[l * 4 + 1] || h[l * 4 + 2] != h[l * 4 + 3]) { ++bad; printf("lane %2d: imm-offset read dword %u vs VGPR-address read %u | 128K: %u vs %u\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]); }
    }
    printf("lanes differing: %d (expected dword index of lane 27 with offset 65104: %u)\n", bad, (27 * 16 + 65104) / 4);
    return 0;
}
