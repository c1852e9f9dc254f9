// LSTM / vanilla-RNN cells of DEC_LargeRNN (decoders.py:27-32: `-dec_rnn lstm | rnn`, 2 layers, bidirectional, batch_first) in the
// f16x2 representation, in the unit-split layout gru_l1f_kernel proved for the GRU (turboae_gru_l1f.hip).  Through r04 these cells ran
// on the generic fp32 kernels only (turboae_generic.hip: 9.9 M bits/s for the LSTM decoder, 3.8x behind the GRU's).
//
// PyTorch cells (the arithmetic lives in ATen, third-party to the reference), gate order i, f, g, o:
//   LSTM  i = s(W_ii x + b_ii + W_hi h + b_hi), f likewise, g = tanh(...), o = s(...);  c' = f c + i g;  h' = o tanh(c');  h_0 = c_0 = 0
//   RNN   h' = tanh(W_ih x + b_ih + W_hh h + b_hh)
// A GRU needs W_ih1 next to W_hh on the chip because its n gate keeps the two products apart; here all gate pre-activations are
// plain sums, and 4 gates x (W_hh + W_ih1) = 480 KB of fp16 pairs per direction do not fit one CU's registers + LDS.  So:
//   rnn_proj_u   GI = W_ih1 * Y0 + b_ih + b_hh for all positions and both directions as one f16x2 GEMM on conv_accumulate_h
//                (gru_proj_h with the row-tile count as a template parameter: 6 G + 1 tiles per direction), written in the
//                recurrence's accumulator layout, pre-multiplied by the recurrence's own power-of-two scale;
//   rnn_rec_u    one workgroup = 32 blocks (two N tiles) of one direction, 8 waves: six unit waves own 16 hidden units each = G gate
//                row tiles, W_hh hi + lo in registers for the whole launch (G x 28 VGPRs); the remainder wave owns units 96..99 as one
//                mixed tile (row 4 qq + g) and, in layer 1, the Linear head tile; h_t is exchanged through LDS as B fragments (one
//                barrier per step).  Layer 0 contracts its K = 2 + F inputs as one more K = 16 slab (staged by the eighth wave);
//                layer 1 starts its accumulators from GI (fetched one step ahead).  Layer 0 writes Y0 as halves in the GRU path's
//                layout, layer 1 the per-direction head products - so gru_head_part closes the stack unchanged.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <type_traits>
#include "turboae_internal.hpp"
#include "turboae_device.hpp"
#include "turboae_y0.hpp"

namespace tae {

namespace {

using u32x4v = __attribute__((ext_vector_type(4))) uint32_t;
using lds_q4 = const u32x4v __attribute__((address_space(3)));
using lds_f4c = const f32x4 __attribute__((address_space(3)));
using lds_w4 = u32x4v __attribute__((address_space(3)));
using lds_w2 = u32x2v __attribute__((address_space(3)));
using lds_ptr = char __attribute__((address_space(3)))*;

// NT = N tiles (of 16 blocks) per workgroup: 2 amortises the step's latencies over twice the MFMAs (large batches); 1 for batches that
// would leave CUs idle (a shorter step on twice the workgroups).  Columns are independent: results do not depend on NT.
constexpr int kProjSlabs = 7;          // k-slabs of the layer-1 input projection (K = 200)

template <int G, int NT = 2> struct Geo {
    static constexpr int kHBsz = NT * 8192;                       // h exchange of one step: per N tile 3 slabs x (hi | lo) + remainder (b1 | b2)
    static constexpr int kXBsz = NT * 2048;                       // layer 0: x_t as the K = 16 slab's (b1 | b2) per N tile
    static constexpr int kBiasB = 6 * G * 64 + 64 + 16;          // [ut][gate][16] + remainder-tile row + (2^-S, 2^-S_head, 0, 0)
    static constexpr int kHB = (kBiasB + 15) / 16 * 16;
    static constexpr int kXB = kHB + 2 * kHBsz;
    static constexpr int kLds = kXB + 2 * kXBsz;
    static constexpr int kFragU = G * 8;                          // fragments of a unit wave: per gate W_hh {3 x (hi, lo), remainder} + the input slab
    static constexpr int kFragR = 8 + 7;                          // remainder wave: mixed tile {7 + input slab} + head tile 7
    static constexpr int kDirB = (6 * kFragU + kFragR) * 1024 + kHB;
};
static_assert(Geo<4>::kDirB == RnnULayout::dir_bytes(4) && Geo<1>::kDirB == RnnULayout::dir_bytes(1) && Geo<4, 1>::kDirB == Geo<4, 2>::kDirB, "host packing");

__device__ __forceinline__ float sigm_f(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896341f * x));
}
__device__ __forceinline__ float tanh_f(float x) {
    return fmaf(-2.0f, __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.88539008177792681f * x)), 1.0f);
}
__device__ __forceinline__ h8 lds_h8(lds_cptr p) { return __builtin_bit_cast(h8, *reinterpret_cast<lds_q4*>(p)); }
__device__ __forceinline__ h8 glb_h8(const char* p) { return __builtin_bit_cast(h8, *reinterpret_cast<const u32x4v*>(p)); }

// LDS writes of this wave are done and visible, then the workgroup barrier (LDS-only fence: global loads / stores stay in flight)
__device__ __forceinline__ void step_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

template <int N, class F, int I = 0>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<N, F, I + 1>(static_cast<F&&>(f));
    }
}

// one 32-k slab over G gate tiles: hi*lo, lo*hi, hi*hi, each product across the gates (independent chains)
template <int G>
__device__ __forceinline__ void mma_g(f32x4 (&acc)[G], const h8 (&ah)[G], const h8 (&al)[G], h8 bh, h8 bl) {
#pragma unroll
    for (int g = 0; g < G; ++g) acc[g] = mfma16x16x32h(ah[g], bl, acc[g]);
#pragma unroll
    for (int g = 0; g < G; ++g) acc[g] = mfma16x16x32h(al[g], bh, acc[g]);
#pragma unroll
    for (int g = 0; g < G; ++g) acc[g] = mfma16x16x32h(ah[g], bh, acc[g]);
}
// K = 8 slab (the W_hh remainder: units 96..99, one real k per lane group; layer 0's inputs: k = 2 kq, 2 kq + 1), the three products
// in ONE MFMA (late r06, as gru_l1f's mma3r; through r06 two, on [4 hi | 4 lo] x ([lo | hi], [hi | 0])): per lane
// A = [hi k0 k1 | hi k0 k1 | lo k0 k1 | 0 0], B = [hi k0 k1 | lo k0 k1 | hi k0 k1 | 0 0]
template <int G>
__device__ __forceinline__ void mma_gr(f32x4 (&acc)[G], const h8 (&ar)[G], h8 b) {
#pragma unroll
    for (int g = 0; g < G; ++g) acc[g] = mfma16x16x32h(ar[g], b, acc[g]);
}

// B operand of the projection's K = 8 tail slab (y0 halves 192..199, late r06): `p` = the lane's address of halves 192 + 2 kq, + 1 in the
// hi plane, `lo_off` bytes further in the lo plane -> [hi k0 k1 | lo k0 k1 | hi k0 k1 | 0 0] (A: pack_rnn_u_proj, slab 6)
__device__ __forceinline__ h8 lds_tail8(lds_cptr p, int lo_off) {
    using lds_u1 = const uint32_t __attribute__((address_space(3)));
    const uint32_t h = *reinterpret_cast<lds_u1*>(p), l = *reinterpret_cast<lds_u1*>(p + lo_off);
    return __builtin_bit_cast(h8, u32x4v{h, l, h, 0u});
}

// cell arithmetic on the de-scaled pre-activations of one (unit, block): returns h', updates c (LSTM)
template <int G>
__device__ __forceinline__ float cell(const float (&a)[G], float& c) {
    if constexpr (G == 4) {
        const float ig = sigm_f(a[0]), fg = sigm_f(a[1]), gg = tanh_f(a[2]), og = sigm_f(a[3]);
        c = fmaf(fg, c, ig * gg);
        return og * tanh_f(c);
    } else {
        return tanh_f(a[0]);
    }
}

struct Ctx {
    const RnnUParams& P;
    const char* wdir;
    lds_cptr lds;
    int lane, n, q, dir, L;
    float inv, inv_head;
    float mul = 1.0f;       // fused layer 1: what rnn_proj_u multiplies GI by (its 2^-S times the recurrence's 2^S)
};

// ---- unit wave: units 16 ut .. 16 ut + 15, G gate tiles -------------------------------------------------------------------------
template <int G, bool LAYER0, int NT>
__device__ __forceinline__ void unit_wave(const Ctx& c, int ut) {
    using GE = Geo<G, NT>;
    constexpr int kNT = NT, kHBsz = GE::kHBsz, kXBsz = GE::kXBsz;
    constexpr int CTT = 6 * G + 1;
    const int lane = c.lane, L = c.L, n = c.n, q = c.q, dir = c.dir;
    const char* wr = c.wdir + (size_t)ut * GE::kFragU * 1024 + lane * 16;
    h8 hh_hi[3][G], hh_lo[3][G], hh_r[G], xw[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
#pragma unroll
        for (int sl = 0; sl < 3; ++sl) {
            hh_hi[sl][g] = glb_h8(wr + (g * 8 + 2 * sl) * 1024);
            hh_lo[sl][g] = glb_h8(wr + (g * 8 + 2 * sl + 1) * 1024);
        }
        hh_r[g] = glb_h8(wr + (g * 8 + 6) * 1024);
        xw[g] = LAYER0 ? glb_h8(wr + (g * 8 + 7) * 1024) : h8{0, 0, 0, 0, 0, 0, 0, 0};
    }
    const lds_cptr bias = c.lds + ut * (G * 64) + q * 16;
    const lds_cptr hb = c.lds + GE::kHB + lane * 16, xb = c.lds + GE::kXB + lane * 16;
    const lds_ptr hw = (lds_ptr)(c.lds + GE::kHB + (ut >> 1) * 2048 + lane * 16 + (ut & 1) * 8);
    const float inv = c.inv;

    for (int grp = blockIdx.x; grp < c.P.ngroups; grp += gridDim.x) {
        // layer 1: GI tiles of this wave, [(g16 L + t) 2 + dir][CTT][lane][4 floats]; layer 0: Y0 rows of the two block groups
        __amdgpu_buffer_rsrc_t rs[kNT];
#pragma unroll
        for (int nt = 0; nt < kNT; ++nt) {
            const size_t g16 = (size_t)grp * kNT + nt;
            rs[nt] = LAYER0 ? __builtin_amdgcn_make_buffer_rsrc(c.P.y0 + g16 * L * (16 * 800), 0, L * 16 * 800, 0x00020000)
                            : __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(c.P.gi) + g16 * L * 2 * (CTT * 256), 0, L * 2 * CTT * 1024, 0x00020000);
        }
        const uint32_t v_gi = (uint32_t)((G * ut) * 1024 + (n * 4 + q) * 16);        // the projection's D layout: block n major, row quad q minor
        uint32_t v_y, v_ylo;                                   // Y0 (layer 0): turboae_y0.hpp
        y0_unit_tile(dir, ut, n, q, v_y, v_ylo);
        f32x4 gi[kNT][G];
        auto fetch_gi = [&](int s, int nt) {
            const uint32_t so = (uint32_t)__builtin_amdgcn_readfirstlane(((dir ? L - 1 - s : s) * 2 + dir) * (CTT * 1024));
#pragma unroll
            for (int g = 0; g < G; ++g) gi[nt][g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs[nt], v_gi + g * 1024, so, 0));
        };
#pragma unroll
        for (int nt = 0; nt < kNT; ++nt) {
            *reinterpret_cast<lds_w2*>(hw + nt * 8192) = u32x2v{0, 0};          // h_{-1} = 0: this wave's slots of buffer 0
            *reinterpret_cast<lds_w2*>(hw + nt * 8192 + 1024) = u32x2v{0, 0};
            if (!LAYER0) fetch_gi(0, nt);
        }
        f32x4 cs[kNT];
#pragma unroll
        for (int nt = 0; nt < kNT; ++nt) cs[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        step_barrier();                                   // B0: h buffer 0 cleared, x of step 0 staged
#pragma unroll 1
        for (int s = 0; s < L; ++s) {
            const int p0 = s & 1, p1 = p0 ^ 1;
            const int t = dir ? L - 1 - s : s;
            f32x4 acc[kNT][G];
#pragma unroll
            for (int nt = 0; nt < kNT; ++nt) {
                const lds_cptr hc = hb + p0 * kHBsz + nt * 8192;
#pragma unroll
                for (int g = 0; g < G; ++g) acc[nt][g] = LAYER0 ? *reinterpret_cast<lds_f4c*>(bias + g * 64) : gi[nt][g];
                if (!LAYER0) fetch_gi(s + 1 < L ? s + 1 : s, nt);            // next step's tiles into the registers just consumed
#pragma unroll
                for (int sl = 0; sl < 3; ++sl) mma_g<G>(acc[nt], hh_hi[sl], hh_lo[sl], lds_h8(hc + sl * 2048), lds_h8(hc + sl * 2048 + 1024));
                mma_gr<G>(acc[nt], hh_r, lds_h8(hc + 6144));
                if (LAYER0) {
                    const lds_cptr xc = xb + p0 * kXBsz + nt * 2048;
                    mma_gr<G>(acc[nt], xw, lds_h8(xc));
                }
            }
#pragma unroll
            for (int nt = 0; nt < kNT; ++nt) {
                f32x4 hn;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float a[G];
#pragma unroll
                    for (int g = 0; g < G; ++g) a[g] = acc[nt][g][i] * inv;
                    float cc = cs[nt][i];
                    hn[i] = cell<G>(a, cc);
                    cs[nt][i] = cc;
                }
                h4 nhi, nlo;
                split4(hn, nhi, nlo);
                const lds_ptr hn_w = hw + p1 * kHBsz + nt * 8192;
                *reinterpret_cast<lds_w2*>(hn_w) = __builtin_bit_cast(u32x2v, nhi);
                *reinterpret_cast<lds_w2*>(hn_w + 1024) = __builtin_bit_cast(u32x2v, nlo);
                if (LAYER0) {        // Y0 as halves, logically [pos'][hi 200 | lo 200] (the projection kernel's operand; turboae_y0.hpp)
                    const uint32_t so = (uint32_t)__builtin_amdgcn_readfirstlane(t * (16 * 800));
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2v, nhi), rs[nt], v_y, so, 0);
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2v, nlo), rs[nt], v_ylo, so, 0);
                }
            }
            step_barrier();
        }
        step_barrier();                                   // the remainder wave's last head product has read the h buffer
    }
}

// ---- remainder wave: units 96..99 as one mixed tile (row 4 qq + g = gate g of unit 96 + qq) + the head tile (layer 1) ---------------
template <int G, bool LAYER0, int NT>
__device__ __forceinline__ void rem_wave(const Ctx& c) {
    using GE = Geo<G, NT>;
    constexpr int kNT = NT, kHBsz = GE::kHBsz, kXBsz = GE::kXBsz;
    constexpr int CTT = 6 * G + 1;
    const int lane = c.lane, L = c.L, n = c.n, q = c.q, dir = c.dir;
    const char* wr = c.wdir + (size_t)6 * GE::kFragU * 1024 + lane * 16;
    h8 hh_hi[3], hh_lo[3], hh_r, xw, hd_hi[3], hd_lo[3], hd_r;
#pragma unroll
    for (int sl = 0; sl < 3; ++sl) { hh_hi[sl] = glb_h8(wr + (2 * sl) * 1024); hh_lo[sl] = glb_h8(wr + (2 * sl + 1) * 1024); }
    hh_r = glb_h8(wr + 6 * 1024);
    xw = glb_h8(wr + 7 * 1024);
#pragma unroll
    for (int sl = 0; sl < 3; ++sl) { hd_hi[sl] = glb_h8(wr + (8 + 2 * sl) * 1024); hd_lo[sl] = glb_h8(wr + (9 + 2 * sl) * 1024); }
    hd_r = glb_h8(wr + 14 * 1024);
    const lds_cptr bias = c.lds + 6 * (G * 64) + q * 16;
    const lds_cptr hb = c.lds + GE::kHB + lane * 16, xb = c.lds + GE::kXB + lane * 16;
    const lds_ptr hw = (lds_ptr)(c.lds + GE::kHB + 6144 + lane * 16);
    const float inv = c.inv, inv_head = c.inv_head;
    auto mma1 = [](f32x4& a, h8 ah, h8 al, h8 bh, h8 bl) {
        a = mfma16x16x32h(ah, bl, a); a = mfma16x16x32h(al, bh, a); a = mfma16x16x32h(ah, bh, a);
    };
    auto mma1r = [](f32x4& a, h8 ar, h8 b) { a = mfma16x16x32h(ar, b, a); };

    for (int grp = blockIdx.x; grp < c.P.ngroups; grp += gridDim.x) {
        __amdgpu_buffer_rsrc_t rs[kNT], rs_h[kNT];
#pragma unroll
        for (int nt = 0; nt < kNT; ++nt) {
            const size_t g16 = (size_t)grp * kNT + nt;
            rs[nt] = LAYER0 ? __builtin_amdgcn_make_buffer_rsrc(c.P.y0 + g16 * L * (16 * 800), 0, L * 16 * 800, 0x00020000)
                            : __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(c.P.gi) + g16 * L * 2 * (CTT * 256), 0, L * 2 * CTT * 1024, 0x00020000);
            rs_h[nt] = __builtin_amdgcn_make_buffer_rsrc(c.P.hpart + g16 * L * 256, 0, L * 1024, 0x00020000);
        }
        const uint32_t v_gi = (uint32_t)((6 * G) * 1024 + (n * 4 + q) * 16);
        const uint32_t v_y = y0_half(n, dir * 100 + 96 + q), v_ylo = y0_half_lo(n, dir * 100 + 96 + q);
        const uint32_t v_h = q < 2 ? (uint32_t)(n * 64 + dir * 32 + q * 16) : 0x80000000u;
        f32x4 gi[kNT];
        auto fetch_gi = [&](int s, int nt) {
            const uint32_t so = (uint32_t)__builtin_amdgcn_readfirstlane(((dir ? L - 1 - s : s) * 2 + dir) * (CTT * 1024));
            gi[nt] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs[nt], v_gi, so, 0));
        };
        auto head = [&](lds_cptr hc, int nt, int t) {
            f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int sl = 0; sl < 3; ++sl) mma1(a, hd_hi[sl], hd_lo[sl], lds_h8(hc + sl * 2048), lds_h8(hc + sl * 2048 + 1024));
            mma1r(a, hd_r, lds_h8(hc + 6144));
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v, a * inv_head), rs_h[nt], v_h, (uint32_t)t * 1024u, 0);
        };
        float cs[kNT];
#pragma unroll
        for (int nt = 0; nt < kNT; ++nt) {
            *reinterpret_cast<lds_w4*>(hw + nt * 8192) = u32x4v{0, 0, 0, 0};
            *reinterpret_cast<lds_w4*>(hw + nt * 8192 + 1024) = u32x4v{0, 0, 0, 0};
            cs[nt] = 0.0f;
            if (!LAYER0) fetch_gi(0, nt);
        }
        step_barrier();                                   // B0
#pragma unroll 1
        for (int s = 0; s < L; ++s) {
            const int p0 = s & 1, p1 = p0 ^ 1;
            const int t = dir ? L - 1 - s : s;
#pragma unroll
            for (int nt = 0; nt < kNT; ++nt) {
                const lds_cptr hc = hb + p0 * kHBsz + nt * 8192;
                f32x4 acc = LAYER0 ? *reinterpret_cast<lds_f4c*>(bias) : gi[nt];
                if (!LAYER0) fetch_gi(s + 1 < L ? s + 1 : s, nt);
#pragma unroll
                for (int sl = 0; sl < 3; ++sl) mma1(acc, hh_hi[sl], hh_lo[sl], lds_h8(hc + sl * 2048), lds_h8(hc + sl * 2048 + 1024));
                mma1r(acc, hh_r, lds_h8(hc + 6144));
                if (LAYER0) {
                    const lds_cptr xc = xb + p0 * kXBsz + nt * 2048;
                    mma1r(acc, xw, lds_h8(xc));
                } else if (s > 0) {
                    head(hc, nt, dir ? L - s : s - 1);        // Linear head on h_{s-1} (the state this step started from)
                }
                float a[G];
#pragma unroll
                for (int g = 0; g < G; ++g) a[g] = acc[g] * inv;
                const float hr = cell<G>(a, cs[nt]);
                const _Float16 hi = (_Float16)hr;
                const _Float16 lo = (_Float16)(hr - (float)hi);
                const h8 br = {hi, 0, lo, 0, hi, 0, 0, 0};               // [hi k0 k1 | lo k0 k1 | hi k0 k1 | 0 0], k0 = unit 96 + q, no k1
                const lds_ptr hn_w = hw + p1 * kHBsz + nt * 8192;
                *reinterpret_cast<lds_w4*>(hn_w) = __builtin_bit_cast(u32x4v, br);
                if (LAYER0) {
                    const uint32_t so = (uint32_t)__builtin_amdgcn_readfirstlane(t * (16 * 800));
                    __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(uint16_t, hi), rs[nt], v_y, so, 0);
                    __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(uint16_t, lo), rs[nt], v_ylo, so, 0);
                }
            }
            step_barrier();
        }
        if (!LAYER0) {
#pragma unroll
            for (int nt = 0; nt < kNT; ++nt) head(hb + (L & 1) * kHBsz + nt * 8192, nt, dir ? 0 : L - 1);
        }
        step_barrier();
    }
}

// ---- eighth wave: layer 0 stages x_t (B, L, 8) as the input slab's B fragments; layer 1 only keeps the barrier count ------------------
template <int G, bool LAYER0, int NT>
__device__ __forceinline__ void stage_wave(const Ctx& c) {
    using GE = Geo<G, NT>;
    constexpr int kNT = NT, kXBsz = GE::kXBsz;
    const int lane = c.lane, L = c.L, n = c.n, q = c.q, dir = c.dir;
    const lds_ptr xw = (lds_ptr)(c.lds + GE::kXB + lane * 16);
    for (int grp = blockIdx.x; grp < c.P.ngroups; grp += gridDim.x) {
        // lane (n, kq) supplies k = 2 kq, 2 kq + 1 of block n (the panel is 8 wide)
        auto stage = [&](int s, int buf) {
            const int t = dir ? L - 1 - s : s;
#pragma unroll
            for (int nt = 0; nt < kNT; ++nt) {
                const size_t b = (size_t)grp * (16 * kNT) + nt * 16 + n;
                float2 v2 = {0.f, 0.f};
                if (b < (size_t)c.P.B) v2 = *reinterpret_cast<const float2*>(c.P.x + (b * L + t) * 8 + 2 * q);
                h4 hi, lo;
                split4(f32x4{v2.x, v2.y, 0.f, 0.f}, hi, lo);
                const u32x2v h2 = __builtin_bit_cast(u32x2v, hi), l2 = __builtin_bit_cast(u32x2v, lo);
                *reinterpret_cast<lds_w4*>(xw + buf * kXBsz + nt * 2048) = u32x4v{h2.x, l2.x, h2.x, 0};              // [hi k0 k1 | lo k0 k1 | hi k0 k1 | 0 0]
            }
        };
        if (LAYER0) stage(0, 0);
        step_barrier();                                   // B0
#pragma unroll 1
        for (int s = 0; s < L; ++s) {
            if (LAYER0 && s + 1 < L) stage(s + 1, (s + 1) & 1);
            step_barrier();
        }
        step_barrier();
    }
}

template <int G, bool LAYER0, int NT>
__global__ __launch_bounds__(512) void rnn_rec_u_kernel(RnnUParams P) {
    using GE = Geo<G, NT>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int dir = blockIdx.y;
    const char* wdir = P.w + (size_t)dir * P.w_dir_stride;
    {
        const f32x4* src = reinterpret_cast<const f32x4*>(wdir + (size_t)(6 * GE::kFragU + GE::kFragR) * 1024);
        for (int i = tid; i < GE::kHB / 16; i += 512) reinterpret_cast<f32x4*>(smem)[i] = src[i];
    }
    __syncthreads();
    const float inv = *reinterpret_cast<const float*>(smem + 6 * G * 64 + 64);
    const float inv_head = *reinterpret_cast<const float*>(smem + 6 * G * 64 + 68);
    const Ctx c{P, wdir, (lds_cptr)smem, lane, lane & 15, lane >> 4, dir, P.L, inv, inv_head};
    if (wave == 3) rem_wave<G, LAYER0, NT>(c);
    else if (wave == 7) stage_wave<G, LAYER0, NT>(c);
    else unit_wave<G, LAYER0, NT>(c, wave < 3 ? wave : wave - 1);
}

// ---- layer 1 with the projection INSIDE (r06): rnn_l1f_u_kernel -------------------------------------------------------------------------
// rnn_proj_u writes 3.2 KB of GI per position (5.1 GB per launch at 16 384 blocks) and is bound by exactly that (2.5 TB/s of writes,
// pipes 41 % busy); the recurrence reads it back.  Here every wave projects ITS OWN gate tiles, two steps at a time
// (a chunk = kTC steps x NT N tiles = NS N-tile-steps), into the registers that then serve as the recurrence's accumulators:
//   P phase   gi[ns][g] = bias + sum over 7 k-slabs of W_ih1 (A fragments streamed L2 -> registers through a ring of kRingF pairs, straight
//             from rnn_proj_u's own image) x Y0 rows (B fragments in LDS, put there by the staging wave with LDS-DMA while the PREVIOUS
//             chunk's recurrence ran), then x mul - per accumulator the same products in the same order as rnn_proj_u, so the two forms
//             are bit-identical (tests/test_gpu_generic.py::test_rnn_layer1_forms_are_bit_identical);
//   R phase   kTC steps of rnn_rec_u's layer-1 step with acc = gi (no GI fetch).
// One more barrier per chunk (the B fragments are free for the next chunk's DMA).  GI never exists; Y0 is read once per direction.
constexpr int kRingF = 4;
template <int G, int NT> struct GeoF {
    using GE = Geo<G, NT>;
    static constexpr int kTC = NT == 1 ? 4 : 2;                  // steps per chunk: four N-tile-steps either way (64 accumulator registers)
    static constexpr int NS = NT * kTC;                          // N-tile-steps of a chunk, ns = tc * NT + nt
    static constexpr int kTileB = 14 * 1024;                     // B fragments of one N-tile-step: [slab 7][hi | lo][lane][8 halves]
    static constexpr int kYB = GE::kXB;                          // layer 1 stages no x_t: the region begins where layer 0's would
    static constexpr int kHR = kYB + NS * kTileB;                // W_hh fragments of the unit waves kept in LDS instead of registers, [ut][gate][3][1 KB]:
    static constexpr int kLds = kHR + 6 * G * 3072;              // the remainder slab and the lo halves of slabs 1, 2 - 48 registers a wave that the
                                                                 // chunk's accumulators need (LSTM: 256 registers and 20 spilled with all of W_hh in registers)
    static constexpr int GB = G < 2 ? G : 2;                     // a block of the P phase: GB gates x NB N-tile-steps = independent accumulators
    static constexpr int NB = (4 / GB) < NS ? (4 / GB) : NS;
};

template <int G, int NT>
__device__ __forceinline__ void unit_wave_f(const Ctx& c, int ut) {
    using GE = Geo<G, NT>;
    using GF = GeoF<G, NT>;
    constexpr int kNT = NT, kTC = GF::kTC, NS = GF::NS, kHBsz = GE::kHBsz, CTT = 6 * G + 1, GB = GF::GB, NB = GF::NB;
    constexpr int NPAIR = kProjSlabs * G;                       // A-fragment pairs of a chunk, in the order (slab, gate)
    constexpr uint32_t DIRB = kProjSlabs * CTT * 2048u;
    const int lane = c.lane, L = c.L, n = c.n, q = c.q, dir = c.dir;
    const char* wr = c.wdir + (size_t)ut * GE::kFragU * 1024 + lane * 16;
    h8 hh_hi[3][G], hh_lo[1][G];
    const lds_cptr hrl = c.lds + GF::kHR + (ut * G) * 3072 + lane * 16;
#pragma unroll
    for (int g = 0; g < G; ++g) {
#pragma unroll
        for (int sl = 0; sl < 3; ++sl) hh_hi[sl][g] = glb_h8(wr + (g * 8 + 2 * sl) * 1024);
        hh_lo[0][g] = glb_h8(wr + (g * 8 + 1) * 1024);
        *reinterpret_cast<lds_w4*>((lds_ptr)hrl + g * 3072) = *reinterpret_cast<const u32x4v*>(wr + (g * 8 + 6) * 1024);             // read back by this wave only
        *reinterpret_cast<lds_w4*>((lds_ptr)hrl + g * 3072 + 1024) = *reinterpret_cast<const u32x4v*>(wr + (g * 8 + 5) * 1024);      // slab 2, lo
        *reinterpret_cast<lds_w4*>((lds_ptr)hrl + g * 3072 + 2048) = *reinterpret_cast<const u32x4v*>(wr + (g * 8 + 3) * 1024);      // slab 1, lo
    }
    const lds_cptr bias = c.lds + ut * (G * 64) + q * 16;          // the PROJECTION's bias rows (the kernel prologue put them there)
    const lds_cptr hb = c.lds + GE::kHB + lane * 16, yb = c.lds + GF::kYB + lane * 16;
    const lds_ptr hw = (lds_ptr)(c.lds + GE::kHB + (ut >> 1) * 2048 + lane * 16 + (ut & 1) * 8);
    const float inv = c.inv, mul = c.mul;
    const __amdgpu_buffer_rsrc_t rsp = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(c.P.wproj), 0, (int)(2 * DIRB + 2 * CTT * 64 + 16), 0x00020000);
    const uint32_t va = (uint32_t)lane * 16u + (uint32_t)dir * DIRB + (uint32_t)(G * ut) * 2048u;
    h8 ah[kRingF], al[kRingF];
    auto a_issue = [&](auto I) {                                // pair i = (slab i / G, gate i % G) into ring slot i % kRingF
        constexpr int i = decltype(I)::value, r = i % kRingF;
        constexpr uint32_t off = (uint32_t)(i / G) * (CTT * 2048u) + (uint32_t)(i % G) * 2048u;
        ah[r] = __builtin_bit_cast(h8, __builtin_amdgcn_raw_buffer_load_b128(rsp, va + off, 0, 0));
        if constexpr (i / G < kProjSlabs - 1) al[r] = __builtin_bit_cast(h8, __builtin_amdgcn_raw_buffer_load_b128(rsp, va + off + 1024, 0, 0));     // the tail slab has one fragment
    };
    static_for<GB>(a_issue);                                     // the first block's fragments (every chunk starts with the same ones)

    for (int grp = blockIdx.x; grp < c.P.ngroups; grp += gridDim.x) {
#pragma unroll
        for (int nt = 0; nt < kNT; ++nt) {
            *reinterpret_cast<lds_w2*>(hw + nt * 8192) = u32x2v{0, 0};          // h_{-1} = 0: this wave's slots of buffer 0
            *reinterpret_cast<lds_w2*>(hw + nt * 8192 + 1024) = u32x2v{0, 0};
        }
        f32x4 cs[kNT];
#pragma unroll
        for (int nt = 0; nt < kNT; ++nt) cs[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        step_barrier();                                   // B0: h buffer 0 cleared, the first chunk's Y0 fragments landed
#pragma unroll 1
        for (int s0 = 0; s0 < L; s0 += kTC) {
            // ---- P phase -------------------------------------------------------------------------------------------------------
            f32x4 gi[NS][G];
#pragma unroll
            for (int ns = 0; ns < NS; ++ns)
#pragma unroll
                for (int g = 0; g < G; ++g) gi[ns][g] = *reinterpret_cast<lds_f4c*>(bias + g * 64);
            static_for<NPAIR / GB>([&](auto BI) {
                constexpr int b = decltype(BI)::value, i0 = b * GB, sl = i0 / G, g0 = i0 % G;
                if constexpr (i0 + GB < NPAIR) static_for<GB>([&](auto J) { a_issue(std::integral_constant<int, i0 + GB + decltype(J)::value>{}); });
                if constexpr (sl == kProjSlabs - 1) {
                    // the K = 8 tail slab: ONE MFMA per accumulator; piece 24 of a row sits in lane (n, 0)'s slot of the slab's tiles
#pragma unroll
                    for (int n0 = 0; n0 < NS; n0 += NB) {
                        h8 bt[NB];
#pragma unroll
                        for (int j = 0; j < NB; ++j) bt[j] = lds_tail8(yb - 252 * q + (n0 + j) * GF::kTileB + sl * 2048, 1024);
#pragma unroll
                        for (int k = 0; k < GB; ++k)
#pragma unroll
                            for (int j = 0; j < NB; ++j) gi[n0 + j][g0 + k] = mfma16x16x32h(ah[(i0 + k) % kRingF], bt[j], gi[n0 + j][g0 + k]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                } else
#pragma unroll
                for (int n0 = 0; n0 < NS; n0 += NB) {
                    h8 bh[NB], bl[NB];
#pragma unroll
                    for (int j = 0; j < NB; ++j) {
                        bh[j] = lds_h8(yb + (n0 + j) * GF::kTileB + sl * 2048);
                        bl[j] = lds_h8(yb + (n0 + j) * GF::kTileB + sl * 2048 + 1024);
                    }
                    // per accumulator hi*lo, hi*hi, lo*hi of this slab (mma_tile_h's order, TAE_MMA_ORDER 1, which rnn_proj_u runs on), each product across the block's accumulators
#pragma unroll
                    for (int k = 0; k < GB; ++k)
#pragma unroll
                        for (int j = 0; j < NB; ++j) gi[n0 + j][g0 + k] = mfma16x16x32h(ah[(i0 + k) % kRingF], bl[j], gi[n0 + j][g0 + k]);
#pragma unroll
                    for (int k = 0; k < GB; ++k)
#pragma unroll
                        for (int j = 0; j < NB; ++j) gi[n0 + j][g0 + k] = mfma16x16x32h(ah[(i0 + k) % kRingF], bh[j], gi[n0 + j][g0 + k]);
#pragma unroll
                    for (int k = 0; k < GB; ++k)
#pragma unroll
                        for (int j = 0; j < NB; ++j) gi[n0 + j][g0 + k] = mfma16x16x32h(al[(i0 + k) % kRingF], bh[j], gi[n0 + j][g0 + k]);
                    __builtin_amdgcn_sched_barrier(0);      // one block at a time: hoisted operand reads of later blocks cost registers the chunk's accumulators need
                }
            });
            static_for<GB>(a_issue);                             // the next chunk's first block: in flight through the whole R phase
#pragma unroll
            for (int ns = 0; ns < NS; ++ns)
#pragma unroll
                for (int g = 0; g < G; ++g) gi[ns][g] *= mul;
#ifdef TAE_L1F_DBG_GI
            if (c.P.gi) {      // debug build: the chunk's accumulators against rnn_proj_u's GI (the caller ran it first)
#pragma unroll
                for (int ns = 0; ns < NS; ++ns) {
                    const int tcd = ns / kNT, ntd = ns % kNT, sd = s0 + tcd < L ? s0 + tcd : L - 1;
                    const size_t g16 = (size_t)grp * kNT + ntd;
                    const float* gp = c.P.gi + g16 * L * 2 * (CTT * 256) + (size_t)(((dir ? L - 1 - sd : sd) * 2 + dir)) * (CTT * 256) + (G * ut) * 256 + (n * 4 + q) * 4;
#pragma unroll
                    for (int g = 0; g < G; ++g) {
                        const f32x4 r = *reinterpret_cast<const f32x4*>(gp + g * 256);
                        const bool bad = r.x != gi[ns][g].x || r.y != gi[ns][g].y || r.z != gi[ns][g].z || r.w != gi[ns][g].w;
                        if (bad && grp == 0 && ntd == 0) printf("gi mismatch dir %d s0 %d ns %d ut %d g %d lane %d (n %d q %d): %g vs %g\n", dir, s0, ns, ut, g, lane, n, q, gi[ns][g].x, r.x);
                    }
                }
            }
#endif
            step_barrier();                                   // BA: every wave has read the chunk's Y0 fragments (the staging wave refills them)
            // ---- R phase: kTC steps of rnn_rec_u's layer 1 ---------------------------------------------------------------------
            static_for<kTC>([&](auto TI) {
                constexpr int tc = decltype(TI)::value;
                const int s = s0 + tc;
                if (s < L) {
                    const int p0 = s & 1, p1 = p0 ^ 1;
#pragma unroll
                    for (int nt = 0; nt < kNT; ++nt) {
                        const lds_cptr hc = hb + p0 * kHBsz + nt * 8192;
                        mma_g<G>(gi[tc * kNT + nt], hh_hi[0], hh_lo[0], lds_h8(hc), lds_h8(hc + 1024));
                        h8 hh_l2[G], hh_r[G];
#pragma unroll
                        for (int g = 0; g < G; ++g) hh_l2[g] = lds_h8(hrl + g * 3072 + 2048);
                        mma_g<G>(gi[tc * kNT + nt], hh_hi[1], hh_l2, lds_h8(hc + 2048), lds_h8(hc + 2048 + 1024));
#pragma unroll
                        for (int g = 0; g < G; ++g) hh_l2[g] = lds_h8(hrl + g * 3072 + 1024);
                        mma_g<G>(gi[tc * kNT + nt], hh_hi[2], hh_l2, lds_h8(hc + 2 * 2048), lds_h8(hc + 2 * 2048 + 1024));
#pragma unroll
                        for (int g = 0; g < G; ++g) hh_r[g] = lds_h8(hrl + g * 3072);
                        mma_gr<G>(gi[tc * kNT + nt], hh_r, lds_h8(hc + 6144));
                    }
#pragma unroll
                    for (int nt = 0; nt < kNT; ++nt) {
                        f32x4 hn;
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            float a[G];
#pragma unroll
                            for (int g = 0; g < G; ++g) a[g] = gi[tc * kNT + nt][g][i] * inv;
                            float cc = cs[nt][i];
                            hn[i] = cell<G>(a, cc);
                            cs[nt][i] = cc;
                        }
                        h4 nhi, nlo;
                        split4(hn, nhi, nlo);
                        const lds_ptr hn_w = hw + p1 * kHBsz + nt * 8192;
                        *reinterpret_cast<lds_w2*>(hn_w) = __builtin_bit_cast(u32x2v, nhi);
                        *reinterpret_cast<lds_w2*>(hn_w + 1024) = __builtin_bit_cast(u32x2v, nlo);
                    }
                    step_barrier();
                }
            });
        }
        step_barrier();                                   // the remainder wave's last head product has read the h buffer
    }
}

template <int G, int NT>
__device__ __forceinline__ void rem_wave_f(const Ctx& c) {
    using GE = Geo<G, NT>;
    using GF = GeoF<G, NT>;
    constexpr int kNT = NT, kTC = GF::kTC, NS = GF::NS, kHBsz = GE::kHBsz, CTT = 6 * G + 1;
    constexpr uint32_t DIRB = kProjSlabs * CTT * 2048u;
    const int lane = c.lane, L = c.L, n = c.n, q = c.q, dir = c.dir;
    const char* wr = c.wdir + (size_t)6 * GE::kFragU * 1024 + lane * 16;
    h8 hh_hi[3], hh_lo[3], hh_r, hd_hi[3], hd_lo[3], hd_r, pa_hi[kProjSlabs], pa_lo[kProjSlabs];
#pragma unroll
    for (int sl = 0; sl < 3; ++sl) { hh_hi[sl] = glb_h8(wr + (2 * sl) * 1024); hh_lo[sl] = glb_h8(wr + (2 * sl + 1) * 1024); }
    hh_r = glb_h8(wr + 6 * 1024);
#pragma unroll
    for (int sl = 0; sl < 3; ++sl) { hd_hi[sl] = glb_h8(wr + (8 + 2 * sl) * 1024); hd_lo[sl] = glb_h8(wr + (9 + 2 * sl) * 1024); }
    hd_r = glb_h8(wr + 14 * 1024);
    {   // the mixed tile's W_ih1 fragments (tile 6 G of the projection image) stay in registers: this wave has them to spare
        const char* pa = c.P.wproj + (size_t)dir * DIRB + (size_t)(6 * G) * 2048 + lane * 16;
#pragma unroll
        for (int sl = 0; sl < kProjSlabs; ++sl) { pa_hi[sl] = glb_h8(pa + (size_t)sl * (CTT * 2048)); pa_lo[sl] = glb_h8(pa + (size_t)sl * (CTT * 2048) + 1024); }     // (the tail slab's lo fragment is zero and unused)
    }
    const lds_cptr bias = c.lds + 6 * (G * 64) + q * 16;
    const lds_cptr hb = c.lds + GE::kHB + lane * 16, yb = c.lds + GF::kYB + lane * 16;
    const lds_ptr hw = (lds_ptr)(c.lds + GE::kHB + 6144 + lane * 16);
    const float inv = c.inv, inv_head = c.inv_head, mul = c.mul;
    auto mma1 = [](f32x4& a, h8 ah, h8 al, h8 bh, h8 bl) {
        a = mfma16x16x32h(ah, bl, a); a = mfma16x16x32h(al, bh, a); a = mfma16x16x32h(ah, bh, a);
    };
    auto mma1r = [](f32x4& a, h8 ar, h8 b) { a = mfma16x16x32h(ar, b, a); };

    for (int grp = blockIdx.x; grp < c.P.ngroups; grp += gridDim.x) {
        __amdgpu_buffer_rsrc_t rs_h[kNT];
#pragma unroll
        for (int nt = 0; nt < kNT; ++nt) {
            const size_t g16 = (size_t)grp * kNT + nt;
            rs_h[nt] = __builtin_amdgcn_make_buffer_rsrc(c.P.hpart + g16 * L * 256, 0, L * 1024, 0x00020000);
        }
        const uint32_t v_h = q < 2 ? (uint32_t)(n * 64 + dir * 32 + q * 16) : 0x80000000u;
        auto head = [&](lds_cptr hc, int nt, int t) {
            f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int sl = 0; sl < 3; ++sl) mma1(a, hd_hi[sl], hd_lo[sl], lds_h8(hc + sl * 2048), lds_h8(hc + sl * 2048 + 1024));
            mma1r(a, hd_r, lds_h8(hc + 6144));
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v, a * inv_head), rs_h[nt], v_h, (uint32_t)t * 1024u, 0);
        };
        float cs[kNT];
#pragma unroll
        for (int nt = 0; nt < kNT; ++nt) {
            *reinterpret_cast<lds_w4*>(hw + nt * 8192) = u32x4v{0, 0, 0, 0};
            *reinterpret_cast<lds_w4*>(hw + nt * 8192 + 1024) = u32x4v{0, 0, 0, 0};
            cs[nt] = 0.0f;
        }
        step_barrier();                                   // B0
#pragma unroll 1
        for (int s0 = 0; s0 < L; s0 += kTC) {
            f32x4 gi[NS];
#pragma unroll
            for (int ns = 0; ns < NS; ++ns) gi[ns] = *reinterpret_cast<lds_f4c*>(bias);
#pragma unroll
            for (int sl = 0; sl < kProjSlabs - 1; ++sl)
#pragma unroll
                for (int ns = 0; ns < NS; ++ns) {           // mma_tile_h's product order (hi*lo, hi*hi, lo*hi), not the recurrence's
                    const h8 bh = lds_h8(yb + ns * GF::kTileB + sl * 2048), bl = lds_h8(yb + ns * GF::kTileB + sl * 2048 + 1024);
                    gi[ns] = mfma16x16x32h(pa_hi[sl], bl, gi[ns]);
                    gi[ns] = mfma16x16x32h(pa_hi[sl], bh, gi[ns]);
                    gi[ns] = mfma16x16x32h(pa_lo[sl], bh, gi[ns]);
                }
#pragma unroll
            for (int ns = 0; ns < NS; ++ns)                  // the K = 8 tail slab: one MFMA
                gi[ns] = mfma16x16x32h(pa_hi[kProjSlabs - 1], lds_tail8(yb - 252 * q + ns * GF::kTileB + (kProjSlabs - 1) * 2048, 1024), gi[ns]);
#pragma unroll
            for (int ns = 0; ns < NS; ++ns) gi[ns] *= mul;
#ifdef TAE_L1F_DBG_GI
            if (c.P.gi) {
#pragma unroll
                for (int ns = 0; ns < NS; ++ns) {
                    const int tcd = ns / kNT, ntd = ns % kNT, sd = s0 + tcd < L ? s0 + tcd : L - 1;
                    const size_t g16 = (size_t)grp * kNT + ntd;
                    const float* gp = c.P.gi + g16 * L * 2 * (CTT * 256) + (size_t)(((dir ? L - 1 - sd : sd) * 2 + dir)) * (CTT * 256) + (6 * G) * 256 + (n * 4 + q) * 4;
                    const f32x4 r = *reinterpret_cast<const f32x4*>(gp);
                    const bool bad = r.x != gi[ns].x || r.y != gi[ns].y || r.z != gi[ns].z || r.w != gi[ns].w;
                    if (bad && grp == 0 && ntd == 0) printf("REM gi mismatch dir %d s0 %d ns %d lane %d (n %d q %d): %g %g %g %g vs %g %g %g %g\n", dir, s0, ns, lane, n, q, gi[ns].x, gi[ns].y, gi[ns].z, gi[ns].w, r.x, r.y, r.z, r.w);
                }
            }
#endif
            step_barrier();                                   // BA
            static_for<kTC>([&](auto TI) {
                constexpr int tc = decltype(TI)::value;
                const int s = s0 + tc;
                if (s < L) {
                    const int p0 = s & 1, p1 = p0 ^ 1;
#pragma unroll
                    for (int nt = 0; nt < kNT; ++nt) {
                        const lds_cptr hc = hb + p0 * kHBsz + nt * 8192;
                        f32x4 acc = gi[tc * kNT + nt];
#pragma unroll
                        for (int sl = 0; sl < 3; ++sl) mma1(acc, hh_hi[sl], hh_lo[sl], lds_h8(hc + sl * 2048), lds_h8(hc + sl * 2048 + 1024));
                        mma1r(acc, hh_r, lds_h8(hc + 6144));
                        if (s > 0) head(hc, nt, dir ? L - s : s - 1);        // Linear head on h_{s-1} (the state this step started from)
                        float a[G];
#pragma unroll
                        for (int g = 0; g < G; ++g) a[g] = acc[g] * inv;
                        const float hr = cell<G>(a, cs[nt]);
                        const _Float16 hi = (_Float16)hr;
                        const _Float16 lo = (_Float16)(hr - (float)hi);
                        const h8 br = {hi, 0, lo, 0, hi, 0, 0, 0};
                        const lds_ptr hn_w = hw + p1 * kHBsz + nt * 8192;
                        *reinterpret_cast<lds_w4*>(hn_w) = __builtin_bit_cast(u32x4v, br);
                    }
                    step_barrier();
                }
            });
        }
#pragma unroll
        for (int nt = 0; nt < kNT; ++nt) head(hb + (L & 1) * kHBsz + nt * 8192, nt, dir ? 0 : L - 1);
        step_barrier();
    }
}

// the eighth wave: Y0 rows of the next chunk -> B fragments in LDS by LDS-DMA (every lane's 16 bytes land at the tile base + lane * 16:
// the rows never touch a register), issued right after the barrier that frees the buffer, waited for before the chunk's last step barrier
template <int G, int NT>
__device__ __forceinline__ void stage_wave_f(const Ctx& c) {
    using GF = GeoF<G, NT>;
    constexpr int kNT = NT, kTC = GF::kTC;
    const int L = c.L, n = c.n, q = c.q, dir = c.dir;
    const lds_ptr ybu = (lds_ptr)(c.lds + GF::kYB);                  // tile bases (no lane term)
    // lane (n, q) takes piece 4 sl + q of row n for slab sl (turboae_y0.hpp): slabs 0..2 from region A, slab 3 = the shared piece 12 | B,
    // slabs 4.. from region B (the last slab's pieces past 24 are K padding: zero weights, initialised halves)
    const uint32_t vA = y0_piece(n, q), v3 = y0_piece(n, 12 + q), v3lo = v3 + y0_lo_add(12 + q), vB = y0_piece(n, 16 + q);
    for (int grp = blockIdx.x; grp < c.P.ngroups; grp += gridDim.x) {
        __amdgpu_buffer_rsrc_t rs[kNT];
#pragma unroll
        for (int nt = 0; nt < kNT; ++nt)
            rs[nt] = __builtin_amdgcn_make_buffer_rsrc(c.P.y0 + ((size_t)grp * kNT + nt) * L * (16 * 800), 0, L * 16 * 800, 0x00020000);
        auto dma_chunk = [&](int s0) {
#pragma unroll
            for (int tc = 0; tc < kTC; ++tc) {
                const int s = s0 + tc < L ? s0 + tc : L - 1;      // a chunk past the end repeats the last step: fetched, never used
                const uint32_t so = (uint32_t)__builtin_amdgcn_readfirstlane(dir ? L - 1 - s : s) * (16 * 800u);
#pragma unroll
                for (int nt = 0; nt < kNT; ++nt) {
                    const lds_ptr y = ybu + (tc * kNT + nt) * GF::kTileB;
#pragma unroll
                    for (int sl = 0; sl < kProjSlabs; ++sl) {
                        const uint32_t vh = sl < 3 ? vA + sl * 64 : (sl == 3 ? v3 : vB + (sl - 4) * 64);
                        const uint32_t vl = sl < 3 ? vA + kY0PlaneAB + sl * 64 : (sl == 3 ? v3lo : vB + kY0PlaneAB + (sl - 4) * 64);
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs[nt], y + (2 * sl) * 1024, 16, vh, so, 0, 0);
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs[nt], y + (2 * sl + 1) * 1024, 16, vl, so, 0, 0);
                    }
                }
            }
        };
        dma_chunk(0);
        __builtin_amdgcn_s_waitcnt(0x0F70);                   // vmcnt(0): an LDS-DMA is a pending LDS write on the VM counter
        step_barrier();                                   // B0
#pragma unroll 1
        for (int s0 = 0; s0 < L; s0 += kTC) {
            step_barrier();                                   // BA: the chunk's fragments have been read
            if (s0 + kTC < L) dma_chunk(s0 + kTC);
#pragma unroll
            for (int tc = 0; tc < kTC; ++tc) {
                if (s0 + tc < L) {
                    if (tc == kTC - 1 || s0 + tc == L - 1) __builtin_amdgcn_s_waitcnt(0x0F70);
                    step_barrier();
                }
            }
        }
        step_barrier();
    }
}

template <int G, int NT>
__global__ __launch_bounds__(512) void rnn_l1f_u_kernel(RnnUParams P) {
    using GE = Geo<G, NT>;
    constexpr int CTT = 6 * G + 1;
    constexpr uint32_t DIRB = kProjSlabs * CTT * 2048u;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int dir = blockIdx.y;
    const char* wdir = P.w + (size_t)dir * P.w_dir_stride;
    {
        const f32x4* src = reinterpret_cast<const f32x4*>(wdir + (size_t)(6 * GE::kFragU + GE::kFragR) * 1024);
        for (int i = tid; i < GE::kHB / 16; i += 512) reinterpret_cast<f32x4*>(smem)[i] = src[i];
    }
    __syncthreads();
    {   // accumulators start from the PROJECTION's bias rows ([tile][16 floats] = the [unit tile][gate][16] rows of this region)
        const float* pb = reinterpret_cast<const float*>(P.wproj + 2 * (size_t)DIRB) + dir * (CTT * 16);
        for (int i = tid; i < CTT * 16; i += 512) reinterpret_cast<float*>(smem)[i] = pb[i];
    }
    __syncthreads();
    const float inv = *reinterpret_cast<const float*>(smem + 6 * G * 64 + 64);
    const float inv_head = *reinterpret_cast<const float*>(smem + 6 * G * 64 + 68);
    const float mul = reinterpret_cast<const float*>(P.wproj + 2 * (size_t)DIRB)[2 * CTT * 16] * P.gi_mul[dir];
    const Ctx c{P, wdir, (lds_cptr)smem, lane, lane & 15, lane >> 4, dir, P.L, inv, inv_head, mul};
    if (wave == 3) rem_wave_f<G, NT>(c);
    else if (wave == 7) stage_wave_f<G, NT>(c);
    else unit_wave_f<G, NT>(c, wave < 3 ? wave : wave - 1);
}

// ---- layer-1 input projections: GI = (W_ih1 * Y0 + b) * gi_mul as an f16x2 GEMM, CTT row tiles per direction -------------------------
// Workgroup = 4 waves, 80 positions (5 tiles) staged in LDS as hi / lo planes of 200 halves; wave = (direction, half of the row tiles),
// passes of <= 5 tiles.  A fragments [slab 7][tile CTT][hi | lo][lane][8 halves] per direction, then the bias rows and 2^-S.
// Panel rows are kProjRow = 416 bytes apart (400 of halves + 16 of zeros): ds_read_b128 is served in the lane groups
// {0-3, 12-15, 20-27}, ... (MI355X_MICROARCH.md, LDS), i.e. eight positions at one k quarter with eight at the next, and a row stride
// of s 16-byte slots keeps those sixteen reads on sixteen different slots iff s = 2 (mod 4): 25 (r05, unpadded) made seven of every
// eight pairs collide (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 47 %), 26 none.
constexpr int kPT = 5, kProjRow = 416;
constexpr int kProjPos = 16 * kPT, kPlaneB = kProjPos * kProjRow + 512, kProjLds = 2 * kPlaneB;

template <int CTT, int C0, int NC>
__device__ __forceinline__ void proj_pass(const RnnProjParams& P, const char* smem, int dir, int lane, size_t p0) {
    const int n = lane & 15, kq = lane >> 4;
    constexpr uint32_t DIRB = kProjSlabs * CTT * 2048u;
    const char* wb = reinterpret_cast<const char*>(P.w);
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(wb), 0, (int)(2 * DIRB + 2 * CTT * 64 + 16), 0x00020000);
    const uint32_t voff = (uint32_t)lane * 16u, soff = (uint32_t)dir * DIRB;
    const float* bias = reinterpret_cast<const float*>(wb + 2 * DIRB) + dir * (CTT * 16);
    const float mul = reinterpret_cast<const float*>(wb + 2 * DIRB)[2 * CTT * 16] * P.gi_mul[dir];
    uint32_t bh[kPT], bl[kPT];
#pragma unroll
    for (int p = 0; p < kPT; ++p) {
        bh[p] = (uint32_t)((p * 16 + n) * kProjRow + 16 * kq);
        bl[p] = bh[p] + (uint32_t)kPlaneB;
    }
    OpsHA<NC> a0;
    load_wh<CTT, C0, NC>(a0, rsrc, voff, soff);
    f32x4 acc[kPT][NC];
#pragma unroll
    for (int ct = 0; ct < NC; ++ct) {
        const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + (C0 + ct) * 16 + 4 * kq);
#pragma unroll
        for (int p = 0; p < kPT; ++p) acc[p][ct] = bv;
    }
    conv_accumulate_h<CTT, C0, NC, kPT, kProjSlabs - 1, true>(acc, a0, rsrc, voff, soff, smem, bh, bl);
    {   // the K = 8 tail slab as ONE MFMA per accumulator, exactly as rnn_l1f_u's P phase issues it (the two forms of layer 1 are bit-identical);
        // its fragments are fetched here, behind the K loop: this kernel serves the small calls, where the projection is a small part
        constexpr uint32_t SB = CTT * 2048;
        h8 at[NC];
#pragma unroll
        for (int ct = 0; ct < NC; ++ct) at[ct] = __builtin_bit_cast(h8, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff + (C0 + ct) * 2048, soff + (kProjSlabs - 1) * SB, 0));
#pragma unroll
        for (int p = 0; p < kPT; ++p) {
            const h8 bt = lds_tail8((lds_cptr)smem + bh[p] + (kProjSlabs - 1) * 64 - 12 * kq, kPlaneB);
#pragma unroll
            for (int ct = 0; ct < NC; ++ct) acc[p][ct] = mfma16x16x32h(at[ct], bt, acc[p][ct]);
        }
    }
#pragma unroll
    for (int p = 0; p < kPT; ++p) {
        const size_t pos = p0 + p * 16 + n;
        if (pos < P.npos) {
            float* dst = P.gi + ((pos >> 4) * 2 + dir) * (size_t)(CTT * 256) + (size_t)C0 * 256 + (n * 4 + kq) * 4;
#pragma unroll
            for (int ct = 0; ct < NC; ++ct) __builtin_nontemporal_store(acc[p][ct] * mul, reinterpret_cast<f32x4*>(dst + ct * 256));
        }
    }
}

template <int CTT>
__global__ __launch_bounds__(256, 2) void rnn_proj_u_kernel(RnnProjParams P) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const size_t p0 = (size_t)blockIdx.x * kProjPos;
    const int np = (int)min((size_t)kProjPos, P.npos - p0);
    {
        for (int i = tid; i < kProjLds / 16; i += 256) reinterpret_cast<f32x4*>(smem)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        __syncthreads();
        for (int i = tid; i < np * 50; i += 256) {      // Y0 rows are halves, logically [hi 200 | lo 200]: 25 16-byte pieces per plane
            const int pos = i / 50, c = i - pos * 50, plane = c >= 25 ? 1 : 0, cc = c - plane * 25;
            const size_t row = p0 + pos;                     // (group, step, block) -> the step's regions (turboae_y0.hpp)
            const char* piece = reinterpret_cast<const char*>(P.yin) + (row >> 4) * kY0StepB + y0_piece((int)(row & 15), cc) + (plane ? y0_lo_add(cc) : 0u);
            *reinterpret_cast<f32x4*>(smem + plane * kPlaneB + pos * kProjRow + cc * 16) = *reinterpret_cast<const f32x4*>(piece);
        }
    }
    __syncthreads();
    const int dir = wave >> 1;
    constexpr int H0 = (CTT + 1) / 2, H1 = CTT - H0;
    if ((wave & 1) == 0) {
        static_for<(H0 + 4) / 5>([&](auto I) {
            constexpr int c0 = 5 * decltype(I)::value, nc = H0 - c0 < 5 ? H0 - c0 : 5;
            proj_pass<CTT, c0, nc>(P, smem, dir, lane, p0);
        });
    } else {
        static_for<(H1 + 4) / 5>([&](auto I) {
            constexpr int c0 = 5 * decltype(I)::value, nc = H1 - c0 < 5 ? H1 - c0 : 5;
            proj_pass<CTT, H0 + c0, nc>(P, smem, dir, lane, p0);
        });
    }
}

template <int G, bool LAYER0, int NT>
hipError_t launch_rec_nt(RnnUParams P, int ncu, hipStream_t st) {
    constexpr int lds = Geo<G, NT>::kLds;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(rnn_rec_u_kernel<G, LAYER0, NT>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return e;
    P.ngroups = (P.B + 16 * NT - 1) / (16 * NT);
    const dim3 grid((unsigned)std::min(P.ngroups, std::max(1, ncu / 2)), 2);       // one workgroup per CU, half of them per direction
    hipLaunchKernelGGL((rnn_rec_u_kernel<G, LAYER0, NT>), grid, dim3(512), lds, st, P);
    return hipGetLastError();
}
template <int G, bool LAYER0>
hipError_t launch_rec(const RnnUParams& P, hipStream_t st) {
    const int ncu = P.ncu > 0 ? P.ncu : 256;
    // one 16-block tile per workgroup while that still fits one round of workgroups (two per pair of directions and CU)
    bool small = 2 * ((P.B + 15) / 16) <= ncu;
    static const int nt_env = [] { const char* e = tae::debug_knob("TAE_RNN_NT"); return e ? atoi(e) : 0; }();     // experiments: read once
    if (nt_env == 1 || nt_env == 2) small = nt_env == 1;
    return small ? launch_rec_nt<G, LAYER0, 1>(P, ncu, st) : launch_rec_nt<G, LAYER0, 2>(P, ncu, st);
}

template <int G, int NT>
hipError_t launch_l1f_nt(RnnUParams P, int ncu, hipStream_t st) {
    constexpr int lds = GeoF<G, NT>::kLds;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(rnn_l1f_u_kernel<G, NT>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return e;
    P.ngroups = (P.B + 16 * NT - 1) / (16 * NT);
    const dim3 grid((unsigned)std::min(P.ngroups, std::max(1, ncu / 2)), 2);
    hipLaunchKernelGGL((rnn_l1f_u_kernel<G, NT>), grid, dim3(512), lds, st, P);
    return hipGetLastError();
}
template <int G>
hipError_t launch_l1f(const RnnUParams& P, hipStream_t st) {
    // ONE N tile per workgroup, four steps per chunk, at every batch size.  The two-tile form (NT = 2, two steps per chunk: the same 64
    // accumulator registers) was as fast and is not instantiated: its LSTM instantiation produced wrong values in blocks 12..15 of
    // every first tile from the second chunk on, in a way that changed with unrelated code (LABNOTES 12.3) - a defect not understood
    // is not shipped.  The bit-identity test against the split form guards the instantiations that are.
    const int ncu = P.ncu > 0 ? P.ncu : 256;
    return launch_l1f_nt<G, 1>(P, ncu, st);
}

template <int CTT>
hipError_t launch_proj(const RnnProjParams& P, hipStream_t st) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(rnn_proj_u_kernel<CTT>), hipFuncAttributeMaxDynamicSharedMemorySize, kProjLds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((rnn_proj_u_kernel<CTT>), dim3((unsigned)((P.npos + kProjPos - 1) / kProjPos)), dim3(256), kProjLds, st, P);
    return hipGetLastError();
}

}  // namespace

hipError_t launch_rnn_rec_u(int gates, bool layer0, const RnnUParams& P, hipStream_t st) {
    if (gates == 4) return layer0 ? launch_rec<4, true>(P, st) : launch_rec<4, false>(P, st);
    if (gates == 1) return layer0 ? launch_rec<1, true>(P, st) : launch_rec<1, false>(P, st);
    return hipErrorInvalidValue;
}

hipError_t launch_rnn_l1f_u(int gates, const RnnUParams& P, hipStream_t st) {
    if (!P.wproj || !P.y0 || !P.hpart) return hipErrorInvalidValue;
    if (gates == 4) return launch_l1f<4>(P, st);
    if (gates == 1) return launch_l1f<1>(P, st);
    return hipErrorInvalidValue;
}

hipError_t launch_rnn_proj_u(int gates, const RnnProjParams& P, hipStream_t st) {
    if (gates == 4) return launch_proj<25>(P, st);
    if (gates == 1) return launch_proj<7>(P, st);
    return hipErrorInvalidValue;
}

}  // namespace tae
