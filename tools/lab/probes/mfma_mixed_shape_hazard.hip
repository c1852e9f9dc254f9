// Does a v_mfma_f32_16x16x16_f16 that takes the result of a just-issued v_mfma_f32_16x16x32_f16 as srcC see the finished accumulator
// on gfx950 (ROCm 7.2 hipcc)?  r03 found the GRU recurrence's results changing from run to run once two waves shared a SIMD
// (DESIGN.md 3.5); this probe isolates the pair: the SAME dependent chain acc = mfma32(a, b, acc); acc = mfma16(c, d, acc) is run
// (1) as the compiler schedules it and (2) with s_nop 15 x 2 after every MFMA (nothing can be in flight when the next one issues).
// Same arithmetic, same order: any difference between (1) and (2), or between two runs of (1), is a read of a stale accumulator.
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_mixed_shape_hazard mfma_mixed_shape_hazard.hip ; run with 1 and with 8 waves per workgroup.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
using f32x4 = __attribute__((ext_vector_type(4))) float;
using h8 = __attribute__((ext_vector_type(8))) _Float16;
using h4 = __attribute__((ext_vector_type(4))) _Float16;

// FIRST / SECOND: K of the two MFMAs of the dependent pair (32 = v_mfma_f32_16x16x32_f16, 16 = v_mfma_f32_16x16x16_f16)
// GAP: independent MFMAs (of the FIRST shape, on another accumulator) issued between the two dependent ones
template <bool SAFE, int FIRST, int SECOND, int GAP = 0>
__global__ __launch_bounds__(512) void chain(const _Float16* __restrict__ src, float* __restrict__ out, int iters) {
    const int lane = threadIdx.x & 63, gw = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    h8 a, b;
    h4 c, d;
    for (int i = 0; i < 8; ++i) { a[i] = src[(gw * 64 + lane) * 8 % 4096 + i]; b[i] = src[(lane * 8 + 1024 + i) % 4096]; }
    for (int i = 0; i < 4; ++i) { c[i] = src[(lane * 4 + 2048 + i) % 4096]; d[i] = src[(lane * 4 + 3072 + gw + i) % 4096]; }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f}, side = {0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
        if (FIRST == 32) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
        else acc = __builtin_amdgcn_mfma_f32_16x16x16f16(c, d, acc, 0, 0, 0);
        if (SAFE) asm volatile("s_nop 15\n\ts_nop 15" : "+v"(acc));
#pragma unroll
        for (int g = 0; g < GAP; ++g) {
            if (FIRST == 32) side = __builtin_amdgcn_mfma_f32_16x16x32_f16(b, a, side, 0, 0, 0);
            else side = __builtin_amdgcn_mfma_f32_16x16x16f16(d, c, side, 0, 0, 0);
        }
        if (SECOND == 32) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(b, a, acc, 0, 0, 0);
        else acc = __builtin_amdgcn_mfma_f32_16x16x16f16(d, c, acc, 0, 0, 0);
        if (SAFE) asm volatile("s_nop 15\n\ts_nop 15" : "+v"(acc));
        acc *= 0.5f;                                   // keeps the values bounded; a VALU read of the chain every iteration
        side *= 0.5f;
    }
    if (GAP > 0 && side[0] == 12345.678f) acc += side;          // keeps the side chain alive
    *reinterpret_cast<f32x4*>(out + ((size_t)(blockIdx.x * blockDim.x + threadIdx.x)) * 4) = acc;
}

int main() {
    std::vector<_Float16> h(4096);
    unsigned s = 12345;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (_Float16)(((int)(s >> 16) % 2001 - 1000) / 1000.0f); }
    _Float16* d_src;
    float *d_a, *d_b, *d_c;
    const int grid = 2048;
    hipMalloc(&d_src, 4096 * 2);
    hipMemcpy(d_src, h.data(), 4096 * 2, hipMemcpyHostToDevice);
    auto run = [&](const char* name, auto kfast, auto ksafe) {
        for (int threads : {64, 512}) {
            const size_t n = (size_t)grid * threads * 4;
            hipMalloc(&d_a, n * 4); hipMalloc(&d_b, n * 4); hipMalloc(&d_c, n * 4);
            hipLaunchKernelGGL(kfast, dim3(grid), dim3(threads), 0, 0, d_src, d_a, 4096);
            hipLaunchKernelGGL(kfast, dim3(grid), dim3(threads), 0, 0, d_src, d_b, 4096);
            hipLaunchKernelGGL(ksafe, dim3(grid), dim3(threads), 0, 0, d_src, d_c, 4096);
            hipDeviceSynchronize();
            std::vector<float> a(n), b(n), c(n);
            hipMemcpy(a.data(), d_a, n * 4, hipMemcpyDeviceToHost);
            hipMemcpy(b.data(), d_b, n * 4, hipMemcpyDeviceToHost);
            hipMemcpy(c.data(), d_c, n * 4, hipMemcpyDeviceToHost);
            size_t ab = 0, ac = 0;
            for (size_t i = 0; i < n; ++i) { ab += memcmp(&a[i], &b[i], 4) != 0; ac += memcmp(&a[i], &c[i], 4) != 0; }
            printf("%-12s %d waves per workgroup: run-to-run differences %8zu of %zu values; as-scheduled vs nop-separated chain: %8zu differ\n", name, threads / 64, ab, n, ac);
            (void)hipFree(d_a); (void)hipFree(d_b); (void)hipFree(d_c);
        }
    };
    run("K32 -> K16", chain<false, 32, 16>, chain<true, 32, 16>);
    run("K16 -> K32", chain<false, 16, 32>, chain<true, 16, 32>);
    run("K32 -> K32", chain<false, 32, 32>, chain<true, 32, 32>);
    run("K16 -> K16", chain<false, 16, 16>, chain<true, 16, 16>);
    // how many independent MFMAs between the pair make the hand-over safe?
    run("K32 1 K16", chain<false, 32, 16, 1>, chain<true, 32, 16, 1>);
    run("K32 2 K16", chain<false, 32, 16, 2>, chain<true, 32, 16, 2>);
    run("K32 3 K16", chain<false, 32, 16, 3>, chain<true, 32, 16, 3>);
    run("K16 1 K32", chain<false, 16, 32, 1>, chain<true, 16, 32, 1>);
    run("K16 2 K32", chain<false, 16, 32, 2>, chain<true, 16, 32, 2>);
    run("K16 3 K32", chain<false, 16, 32, 3>, chain<true, 16, 32, 3>);
    return 0;
}
