// tae_probe_mfma_f16: what the f16 matrix pipes of THIS device sustain on non-zero operands, measured where the benchmark runs.
//
// The decoder's roofline is priced against the dense fp16 MFMA SPEC peak (2.5 PFLOP/s / 3 products per fp32-equivalent MAC).  On
// real data the chip clocks to its power budget well below that (/opt/skills/guides/MI355X_MICROARCH.md, DVFS
// give-back), so bench.py reports next to the spec-peak fraction how far the decoder is from what a PURE stream of
// v_mfma_f32_16x16x32_f16 reaches on the same device a moment earlier.  Same residency as the decoder: one 8-wave workgroup per CU
// (two waves per SIMD, pinned by dynamic LDS nobody touches), 16 independent accumulator tiles per wave, A / B fragments refreshed
// from an LDS table every iteration so the operands toggle like data.  No reference counterpart (measurement support).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <vector>

#include "../../include/turboae_hip.h"
#include "philox.hpp"
#include "turboae_internal.hpp"

namespace tae {

using f32x4p = __attribute__((ext_vector_type(4))) float;
using h8p = __attribute__((ext_vector_type(8))) _Float16;

constexpr int kProbeIters = 4096;
constexpr int kProbeTable = 1024;                 // h8 entries = 16 KB

__global__ __launch_bounds__(512, 2) void mfma_f16_stream_kernel(const h8p* __restrict__ tab, float* __restrict__ out) {
    __shared__ h8p lds[kProbeTable];
    for (int i = threadIdx.x; i < kProbeTable; i += 512) lds[i] = tab[i];
    __syncthreads();
    f32x4p acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = f32x4p{0.f, 0.f, 0.f, 0.f};
    const int lane = threadIdx.x & 63;
    for (int it = 0; it < kProbeIters; ++it) {
        h8p a[4], b[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            a[i] = lds[(it * 8 + i) * 64 % 960 + lane];
            b[i] = lds[(it * 8 + 4 + i) * 64 % 960 + lane];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[j], acc[i * 4 + j], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[(size_t)blockIdx.x * 512 + threadIdx.x] = s;
}

}  // namespace tae

#define PROBE_HIP(x)                                                        \
    do {                                                                    \
        hipError_t e_ = (x);                                                \
        if (e_ != hipSuccess) { rc = tae::fail_msg(TAE_EHIP, hipGetErrorString(e_)); goto done; } \
    } while (0)

extern "C" int tae_probe_mfma_f16(int32_t zero_data, int32_t min_ms, double* tflops, double* ms_measured) {
    using namespace tae;
    if (!tflops) return fail_msg(TAE_EINVAL, "NULL argument");
    if (min_ms < 1 || min_ms > 5000) return fail_msg(TAE_EINVAL, "min_ms must be in 1..5000");
    int rc = TAE_OK;
    h8p* tab = nullptr;
    float* out = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    const int grid = 256 * 8;                     // 8 rounds of one workgroup per CU
    const int dyn = 96 * 1024;                    // residency: ONE workgroup (2 waves per SIMD) per CU, like the decoder
    const double flops_per_launch = 2.0 * grid * 8.0 * kProbeIters * 16.0 * 16 * 16 * 32;
    {
        std::vector<uint16_t> h((size_t)kProbeTable * 8, 0);
        if (!zero_data) {
            // N(0,1) as fp16 from the library's own Philox stream (Box-Muller), fixed seed: the same table every run
            for (size_t c = 0; c < h.size() / 4; ++c) {
                const u32x4 w = philox_call(0x7AE0F16ull, STREAM_WEIGHTS, c);
                const double r0 = sqrt(-2.0 * log(u32_to_unit_open(w.x))), t0 = 6.283185307179586 * u32_to_unit_open(w.y);
                const double r1 = sqrt(-2.0 * log(u32_to_unit_open(w.z))), t1 = 6.283185307179586 * u32_to_unit_open(w.w);
                const float z[4] = {(float)(r0 * cos(t0)), (float)(r0 * sin(t0)), (float)(r1 * cos(t1)), (float)(r1 * sin(t1))};
                for (int k = 0; k < 4; ++k) {
                    const _Float16 v = (_Float16)z[k];
                    uint16_t bits;
                    __builtin_memcpy(&bits, &v, 2);
                    h[4 * c + k] = bits;
                }
            }
        }
        PROBE_HIP(hipMalloc(&tab, h.size() * 2));
        PROBE_HIP(hipMalloc(&out, (size_t)grid * 512 * sizeof(float)));
        PROBE_HIP(hipMemcpy(tab, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    }
    PROBE_HIP(hipEventCreate(&e0));
    PROBE_HIP(hipEventCreate(&e1));
    PROBE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(mfma_f16_stream_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, dyn));
    {
        // one launch is ~12 ms at the sustained rate: two warm-ups (clock ramp), then enough launches for >= min_ms
        float ms1 = 0.f;
        hipLaunchKernelGGL(mfma_f16_stream_kernel, dim3(grid), dim3(512), dyn, 0, tab, out);
        PROBE_HIP(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(mfma_f16_stream_kernel, dim3(grid), dim3(512), dyn, 0, tab, out);
        PROBE_HIP(hipEventRecord(e1, 0));
        PROBE_HIP(hipEventSynchronize(e1));
        PROBE_HIP(hipEventElapsedTime(&ms1, e0, e1));
        int reps = (int)((double)min_ms / (ms1 > 0.1f ? ms1 : 0.1f)) + 1;
        if (reps > 2000) reps = 2000;
        float ms = 0.f;
        PROBE_HIP(hipEventRecord(e0, 0));
        for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(mfma_f16_stream_kernel, dim3(grid), dim3(512), dyn, 0, tab, out);
        PROBE_HIP(hipEventRecord(e1, 0));
        PROBE_HIP(hipEventSynchronize(e1));
        PROBE_HIP(hipGetLastError());
        PROBE_HIP(hipEventElapsedTime(&ms, e0, e1));
        *tflops = flops_per_launch * reps / (ms * 1e-3) / 1e12;
        if (ms_measured) *ms_measured = ms;
    }
done:
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (tab) (void)hipFree(tab);
    if (out) (void)hipFree(out);
    return rc;
}
