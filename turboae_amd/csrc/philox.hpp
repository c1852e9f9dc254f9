// Philox4x32-10 counter-based generator, host+device, bit-identical to turboae_amd/philox.py.
// Stream layout: counter = (idx_lo, idx_hi, stream, 0), key = (seed_lo, seed_hi); one call yields
// 4 x u32, element e of a stream is word (e & 3) of call (e >> 2).
//
// Replaces the reference's unseeded host RNG draws for test inputs (trainer.py:167
// torch.randint, channels.py:35 torch.randn) so any rank can generate any shard on device.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define TAE_HD __host__ __device__ __forceinline__
#else
#define TAE_HD inline
#endif

namespace tae {

enum : uint32_t { STREAM_BITS = 1, STREAM_NOISE = 2, STREAM_WEIGHTS = 3,
                  // channel generators (tae_generate_noise): keep / radar-position masks, Gilbert-Elliott state walk, second and third
                  // normal streams (radar bursts, fading_h), chi-square draws of the t distribution (attempt index in counter word 3)
                  STREAM_MASK = 4, STREAM_CHAIN = 5, STREAM_AUX_A = 6, STREAM_AUX_B = 7, STREAM_GAMMA = 8 };

struct u32x4 { uint32_t x, y, z, w; };

TAE_HD u32x4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)M0 * c0;
        const uint64_t p1 = (uint64_t)M1 * c2;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        const uint32_t n0 = hi1 ^ c1 ^ k0;
        const uint32_t n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += W0; k1 += W1;
    }
    return u32x4{c0, c1, c2, c3};
}

// words of call `call` of stream `stream` under `seed`
TAE_HD u32x4 philox_call(uint64_t seed, uint32_t stream, uint64_t call) {
    return philox4x32_10((uint32_t)call, (uint32_t)(call >> 32), stream, 0u, (uint32_t)seed, (uint32_t)(seed >> 32));
}

// word e of a stream as a uniform in (0,1); normal e of a stream (Box-Muller in fp64: call e >> 2, pair (e & 3) >> 1, even e takes
// the cosine branch, odd e the sine branch - the layout of turboae_amd/philox.py::random_normal)
// u32 -> double in (0,1): ((x >> 8) + 0.5) * 2^-24
TAE_HD double u32_to_unit_open(uint32_t x) { return ((double)(x >> 8) + 0.5) * (1.0 / 16777216.0); }

#if defined(__HIPCC__)
__device__ __forceinline__ double philox_uniform(uint64_t seed, uint32_t stream, uint64_t e) {
    const u32x4 w = philox_call(seed, stream, e >> 2);
    const uint32_t k = (uint32_t)e & 3u;
    return u32_to_unit_open(k == 0 ? w.x : k == 1 ? w.y : k == 2 ? w.z : w.w);
}
__device__ __forceinline__ double philox_normal(uint64_t seed, uint32_t stream, uint64_t e) {
    const u32x4 w = philox_call(seed, stream, e >> 2);
    const bool second = ((uint32_t)e & 2u) != 0;
    const double r = sqrt(-2.0 * log(u32_to_unit_open(second ? w.z : w.x)));
    const double th = 6.283185307179586476925 * u32_to_unit_open(second ? w.w : w.y);
    return ((uint32_t)e & 1u) ? r * sin(th) : r * cos(th);
}
#endif

}  // namespace tae
