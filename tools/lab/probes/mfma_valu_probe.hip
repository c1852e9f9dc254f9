// Micro-probe: what does one extra VALU instruction of a given kind cost when threaded through a stream of
// v_mfma_f32_16x16x4_f32 on gfx950?  One wave per SIMD (256 threads, 1 block per CU), 60 MFMAs per iteration
// with N filler ops of one kind interleaved.  Build: hipcc --offload-arch=gfx950 -O3 -o probe mfma_valu_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
using f32x4 = __attribute__((ext_vector_type(4))) float;

template <int KIND, int NF>
__global__ __launch_bounds__(256, 1) void probe(float* out, int iters, float seed) {
    f32x4 acc[30];
    for (int i = 0; i < 30; ++i) acc[i] = f32x4{seed, seed, seed, seed};
    float a = seed + threadIdx.x, b = seed * 0.5f;
    float f[8];
    for (int i = 0; i < 8; ++i) f[i] = seed + i;
    float g[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    __shared__ float lds[1024];
    lds[threadIdx.x] = seed;
    __syncthreads();
    unsigned sreg = (unsigned)iters;
    unsigned ldsaddr = (threadIdx.x & 63) * 8;
    const float* gptr = out + (threadIdx.x & 63) * 2;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 60; ++m) {
            acc[m % 30] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[m % 30], 0, 0, 0);
            if (m < NF) {
                float& x = f[m % 8];
                if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x) : "v"(b));
                if (KIND == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(x));
                if (KIND == 2) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(b));
                if (KIND == 3) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x) : "v"(b));
                if (KIND == 4) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(b));
                if (KIND == 5) asm volatile("v_mov_b32 %0, %1" : "+v"(x) : "v"(b));
                if (KIND == 6) asm volatile("v_max_f32 %0, %0, %1" : "+v"(x) : "v"(b));
                if (KIND == 7) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(*(double*)&f[(m % 4) * 2]) : "v"(*(double*)&f[((m + 1) % 4) * 2]));
                if (KIND == 8) asm volatile("v_and_b32 %0, %0, %1" : "+v"(x) : "v"(b));
                if (KIND == 9) asm volatile("v_cmp_gt_f32 vcc, %0, %1" :: "v"(x), "v"(b) : "vcc");
                if (KIND == 10) asm volatile("s_add_u32 %0, %0, 4" : "+s"(sreg));
                if (KIND == 11) asm volatile("ds_read_b64 %0, %1" : "=v"(*(double*)&g[(m % 4) * 2]) : "v"(ldsaddr));
                if (KIND == 12) asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(*(double*)&g[(m % 4) * 2]) : "v"(gptr));
                if (KIND == 13) asm volatile("s_nop 0");
                if (KIND == 14) asm volatile("v_accvgpr_read_b32 %0, a0" : "=v"(x));
                if (KIND == 15) asm volatile("v_min_f32 %0, %0, %1" : "+v"(x) : "v"(b));
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 30; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    for (int i = 0; i < 8; ++i) s += f[i] + g[i];
    s += (float)sreg;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int KIND, int NF>
float run(float* d, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    probe<KIND, NF><<<256, 256>>>(d, 10, 0.0f);
    hipEventRecord(e0);
    probe<KIND, NF><<<256, 256>>>(d, iters, 0.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    float* d; hipMalloc(&d, 256 * 256 * 4);
    const int iters = 20000;
    const char* names[] = {"v_fma_f32", "v_exp_f32", "v_add_f32", "v_mul_f32", "v_cndmask", "v_mov_b32", "v_max_f32", "v_pk_fma_f32", "v_and_b32", "v_cmp_gt_f32", "s_add_u32", "ds_read_b64", "global_load_x2", "s_nop", "v_accvgpr_read", "v_min_f32"};
    float base = run<0, 0>(d, iters);
    double cyc = base * 1e-3 * 2.4e9 / iters / 60;   // cycles per MFMA at nominal 2.4 GHz
    printf("base: %.3f ms, %.2f cyc/MFMA (nominal 2.4 GHz)\n", base, cyc);
#define P(K) { float t30 = run<K, 30>(d, iters), t60 = run<K, 60>(d, iters); \
    printf("%-14s  +30 fillers: %.3f ms (%.2f cyc/filler)   +60 fillers: %.3f ms (%.2f cyc/filler)\n", names[K], t30, \
           (t30 - base) * 1e-3 * 2.4e9 / iters / 30, t60, (t60 - base) * 1e-3 * 2.4e9 / iters / 60); }
    P(0) P(1) P(2) P(3) P(4) P(5) P(6) P(7) P(8) P(9) P(10) P(11) P(12) P(13) P(14) P(15)
    return 0;
}
