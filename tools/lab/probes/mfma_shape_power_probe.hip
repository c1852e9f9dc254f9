// Sustained (power-limited) rate of v_mfma_f32_16x16x32_f16 vs v_mfma_f32_32x32x16_f16 on NON-ZERO data, whole chip.
// Question: would re-tiling the f16x2 decoder from 16x16 to 32x32 tiles buy clock (fewer operand reads per MAC)?
// Every workgroup = 8 waves (2 per SIMD), each wave streams MFMAs over 16 independent accumulators; the A / B operands
// are refreshed from a small LDS table every iteration (so operands toggle like real data; zero data clocks higher).
// Runs ~0.5 s per variant so DVFS settles; reports TFLOP/s and the effective MFMA issue interval.
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_shape_power_probe mfma_shape_power_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using h8 = __attribute__((ext_vector_type(8))) _Float16;

constexpr int kIters = 4096;

// 16x16x32: 16 accumulator tiles (64 regs), 4 A x 4 B operand sets
__global__ __launch_bounds__(512, 2) void k16(const h8* __restrict__ tab, float* out, int zero) {
    __shared__ h8 lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 512) lds[i] = zero ? h8{0, 0, 0, 0, 0, 0, 0, 0} : tab[i];
    __syncthreads();
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int lane = threadIdx.x & 63;
    for (int it = 0; it < kIters; ++it) {
        h8 a[4], b[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { a[i] = lds[(it * 8 + i) * 64 % 960 + lane]; b[i] = lds[(it * 8 + 4 + i) * 64 % 960 + lane]; }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[j], acc[i * 4 + j], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

// 32x32x16: 4 accumulator tiles (64 regs), 2 A x 2 B operand sets, two K = 16 instructions per 32 k -> same MACs per iteration
// as k16 (16 tiles x 16*16*32 = 4 tiles x 2 x 32*32*16) and the same LDS operand bytes per MAC would be HALF; here the
// operand refresh is kept at 8 fragments per iteration like k16 (worst case for the comparison)
__global__ __launch_bounds__(512, 2) void k32(const h8* __restrict__ tab, float* out, int zero) {
    __shared__ h8 lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 512) lds[i] = zero ? h8{0, 0, 0, 0, 0, 0, 0, 0} : tab[i];
    __syncthreads();
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    const int lane = threadIdx.x & 63;
    for (int it = 0; it < kIters; ++it) {
        h8 a[4], b[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { a[i] = lds[(it * 8 + i) * 64 % 960 + lane]; b[i] = lds[(it * 8 + 4 + i) * 64 % 960 + lane]; }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i * 2 + kk], b[j * 2 + kk], acc[i * 2 + j], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 16; ++j) s += acc[i][j];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <class K>
static void run(const char* name, K kern, const h8* tab, float* out, int zero, int grid) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int dyn = 96 * 1024;       // dynamic LDS nobody touches: caps residency at ONE workgroup (2 waves per SIMD) per CU, like the decoder
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, dyn);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(kern, dim3(grid), dim3(512), dyn, 0, tab, out, zero);
    hipDeviceSynchronize();
    const int reps = 12;
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(kern, dim3(grid), dim3(512), dyn, 0, tab, out, zero);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double macs = (double)reps * grid * 8.0 * kIters * 16.0 * 16 * 16 * 32;     // per wave and iteration: 16 tiles of 16x16x32
    const double tf = 2.0 * macs / (ms * 1e-3) / 1e12;
    printf("%-28s %s data: %8.2f ms  %7.0f TFLOP/s  (%.1f %% of 2500)\n", name, zero ? "zero  " : "random", ms, tf, tf / 25.0);
}

int main() {
    std::mt19937 rng(1);
    std::normal_distribution<float> nd(0.f, 1.f);
    std::vector<_Float16> h(1024 * 8);
    for (auto& v : h) v = (_Float16)nd(rng);
    h8* tab;
    float* out;
    const int grid = 256 * 8;           // 8 rounds of one workgroup per CU (2 would fit by registers; LDS is small)
    hipMalloc(&tab, h.size() * 2);
    hipMalloc(&out, (size_t)grid * 512 * 4);
    hipMemcpy(tab, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    for (int pass = 0; pass < 2; ++pass) {
        run("v_mfma_f32_16x16x32_f16", k16, tab, out, 0, grid);
        run("v_mfma_f32_32x32x16_f16", k32, tab, out, 0, grid);
    }
    run("v_mfma_f32_16x16x32_f16", k16, tab, out, 1, grid);
    run("v_mfma_f32_32x32x16_f16", k32, tab, out, 1, grid);
    return 0;
}
