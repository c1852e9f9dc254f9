// Shared device building blocks of the MFMA kernels (implicit-GEMM K loop, fragment loads, ELU).
// Included by turboae_kernels.hip (CNN encoder / decoder) and turboae_gru.hip (GRU decoder).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#ifndef TAE_X
#define TAE_X 0   // timing-experiment bitmask (results become wrong): 1 no layer barriers, 2 no panel writes, 4 no ELU, 8 no weight loads in loop, 16 no LDS reads in loop
#endif

namespace tae {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using lds_cptr = const char __attribute__((address_space(3)))*;   // explicit LDS pointer (ds_* instructions)

// ELU(alpha=1) = x > 0 ? x : expm1(x)  (F.elu, cnn_utils.py:26,43) as med3(x, exp(x) - 1, 0):
// exp(x) - 1 >= x everywhere, so the median of {x, exp(x)-1, 0} is x for x > 0 and exp(x)-1 for x < 0
// (large x: exp overflows to +inf, the median is still x).  4 VALU ops, branch- and compare-free -
// on gfx950 every VALU op issued by a wave that is streaming fp32 MFMAs costs ~5 matrix-pipe cycles,
// v_cmp / v_exp ~9 (tools/probes/mfma_valu_probe.hip).  Absolute error <= 1.2e-7 for every x (one
// rounding of exp near 1, one of the subtraction); the relative error for tiny negative x is not
// preserved (expm1 would return ~x), which is below the rounding noise of the 500-term fp32 dot
// products that consume these activations.
__device__ __forceinline__ float elu1(float x) {
    const float e = __builtin_amdgcn_exp2f(x * 1.44269504088896341f) - 1.0f;
    return __builtin_amdgcn_fmed3f(x, e, 0.0f);
}

__device__ __forceinline__ f32x4 mfma16x16x4(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// Workgroup shape: 8 waves = 2 per SIMD.  Wave w = g + 4h: g selects the position group (PT
// position tiles), h the channel half (tiles [0, CTA) or [CTA, CT)), so the two waves that share a
// SIMD (w and w + 4 land on the same SIMD) split one position group's channel tiles 4 + 3 and
// every SIMD carries the same MFMA load.  Two waves per SIMD let one wave's MFMAs cover the
// other's vector-memory issue, s_waitcnt parking, barrier skew and (vector-ALU-only) epilogues.
constexpr int kWaves = 8;
constexpr int kGroups = 4;                 // position groups per workgroup
constexpr int kThreads = 64 * kWaves;
constexpr int kXW = 8;                     // floats per row of the XA / XB input panels
constexpr int kHeadSlots = kGroups * 5 * 16;   // positions per workgroup (head-combine scratch rows)

// Operands of one K-chunk (8 k values = 2 MFMA k-steps) for a wave's PT x NC grid of 16x16 tiles.
template <int NC, int PT>
struct Ops {
    float2 a[NC];
    float2 b[PT];
};

// number of load instructions load_w issues for channel tiles [C0, C0 + NC) of CTT
constexpr int n_wloads(int CTT, int C0, int NC) {
    int n = 0;
    for (int i = 0; i < NC; ++i) {
        const int ct = C0 + i;
        const bool paired = ct < 2 * (CTT / 2);
        if (paired && (ct % 2 == 1) && i >= 1) continue;
        ++n;
    }
    return n;
}

// Weight (A) fragments of channel tiles [C0, C0 + NC) of one chunk through a buffer resource:
// wave-uniform SGPR byte offset `soff`, per-lane VGPR offset `voff` (lane * 16), everything else
// immediates - no vector-ALU address arithmetic.  Packed chunk layout: channel tiles in pairs
// [pair][lane][ct even: k0 k1 | ct odd: k0 k1] (one 16-byte load fetches both) and, for odd CTT, a
// trailing [lane][k0 k1] tile at (CTT/2) * 1024 bytes.
template <int CTT, int C0, int NC, int PT>
__device__ __forceinline__ void load_w(Ops<NC, PT>& o, __amdgpu_buffer_rsrc_t rsrc, uint32_t voff, uint32_t soff) {
    if (TAE_X & 8) { asm volatile("" :: "s"(soff)); return; }
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        const int ct = C0 + i;
        const bool paired = ct < 2 * (CTT / 2);
        if (paired && (ct % 2 == 1) && i >= 1) continue;      // already fetched with its even partner
        if (paired && (ct % 2 == 0) && (i + 1 < NC)) {
            // (bit_cast the whole vector: element-wise bit_cast of v[i] is mis-folded by this clang)
            const f32x4 f = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff + (ct / 2) * 1024, soff, 0));
            o.a[i] = float2{f.x, f.y};
            o.a[i + 1 < NC ? i + 1 : i] = float2{f.z, f.w};
        } else if (paired) {
            const f32x2 f = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rsrc, voff + (ct / 2) * 1024 + (ct % 2) * 8, soff, 0));
            o.a[i] = float2{f.x, f.y};
        } else {
            const f32x2 f = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rsrc, (voff >> 1) + (CTT / 2) * 1024, soff, 0));
            o.a[i] = float2{f.x, f.y};
        }
    }
}

// Activation (B) fragments: one ds_read_b64 per position tile at running base + immediate.
template <int NC, int PT, int OFF>
__device__ __forceinline__ void load_x(Ops<NC, PT>& o, const lds_cptr (&cur)[PT]) {
    if (TAE_X & 16) return;
#pragma unroll
    for (int p = 0; p < PT; ++p) {
        const f32x2 v = *reinterpret_cast<const f32x2 __attribute__((address_space(3)))*>(cur[p] + OFF);
        o.b[p] = float2{v.x, v.y};
    }
}

#ifndef TAE_SPREAD
#define TAE_SPREAD 1
#endif
// Issue-order hint for one chunk region {weight loads + LDS reads of a later chunk, MFMAs of this one}:
// spread the vector-memory loads evenly through the MFMA stream.  The waves of a workgroup fetch the
// same fragments at nearly the same time; issued back to back the 1 KB loads saturate the CU's
// 64 B/clk texture-address path and the MFMAs queued behind them in program order wait.
template <int NV, int ND, int NM>
__device__ __forceinline__ void spread_loads() {
#if TAE_SPREAD
    constexpr int GV = NM / (2 * NV);          // MFMAs between vector-memory loads (first half of the chunk)
    constexpr int GD = (NM - NV * GV) / ND;    // MFMAs between LDS reads (second half)
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, GV, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
    }
#pragma unroll
    for (int i = 0; i < ND; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, GD, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x008, NM - NV * GV - ND * GD, 0);
#endif
}

template <int NC, int PT>
__device__ __forceinline__ void mma_chunk(f32x4 (&acc)[PT][NC], const Ops<NC, PT>& o) {
#pragma unroll
    for (int p = 0; p < PT; ++p)
#pragma unroll
        for (int ct = 0; ct < NC; ++ct) acc[p][ct] = mfma16x16x4(o.a[ct].x, o.b[p].x, acc[p][ct]);
#pragma unroll
    for (int p = 0; p < PT; ++p)
#pragma unroll
        for (int ct = 0; ct < NC; ++ct) acc[p][ct] = mfma16x16x4(o.a[ct].y, o.b[p].y, acc[p][ct]);
}

// ---- "super-tile" for the remainder channels --------------------------------------------------------
// U = 100 leaves 4 channels (96..99) in a 7th 16-row tile that is 75 % padding.  Instead of M = 16
// channels x N = 16 positions, the remainder uses M = 4 position shifts x 4 channels and N = 16 position
// QUADS: row r = 4s + c of A is channel 96 + c of the (s)-th position of a quad, column n is quad n, and the
// contraction runs over K' = 8 shifts x C_in, K' = u * C_in + ci, with A[(s,c)][(u,ci)] = W[96+c][ci][u - s]
// for 0 <= u - s <= 4 and 0 otherwise, B[(u,ci)][n] = X[row(quad n) - 2 + u][ci] - again one contiguous run of
// panel floats per column, so B fragments are single ds_read_b64s.  One tile thus produces 64 positions x 4
// channels in 2 * (K'/8) MFMAs (200 for a U->U layer) where the padded tile needs 4 * 126 = 504.  The two
// waves of a SIMD take one super-tile each (quads 0..15 / 16..19 of their position group's 20) and three
// regular channel tiles each, which balances them exactly.  Every
// position is computed by the same chain whatever its place in the batch (shift s = index-in-block mod 4),
// so results stay bitwise independent of batching.  Lane (n, q) receives D rows 4q..4q+3 = the 4 channels
// of position 4n + q of its quads.  Requires quads not to straddle blocks (block_len % 4 == 0).
struct SuperCtx {      // this wave's super-tile (lower channel half: quads 0..15, upper half: quads 16..19)
    int row0;         // panel row of the first position of this lane's quad
    int slot;         // workgroup-wide position slot (head scratch row) of this lane's position
    bool valid;       // this lane's position is a real in-block position (its activations are written back)
    bool center;      // ... whose stack output this workgroup owns
};

struct SOps {
    f32x4 a;          // A fragments of a chunk pair: even chunk (k-step x, y), odd chunk (x, y)
    float2 b[2];      // B fragments of the two chunks
};

// chunk pair `PAIR` relative to the loop-carried bases (ssoff: SGPR byte offset of the super A fragments,
// scur: LDS pointer of this lane's quad column); everything else is an immediate
template <int PAIR>
__device__ __forceinline__ void sload(SOps& o, __amdgpu_buffer_rsrc_t rsrc, uint32_t voff, uint32_t ssoff, lds_cptr scur) {
    o.a = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, ssoff + PAIR * 1024u, 0));
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const f32x2 v = *reinterpret_cast<const f32x2 __attribute__((address_space(3)))*>(scur + PAIR * 64 + c * 32);
        o.b[c] = float2{v.x, v.y};
    }
}

// two independent accumulation chains (k-step parity) so consecutive MFMAs never wait on each other
__device__ __forceinline__ void smma(f32x4 (&accS)[2], const SOps& o) {
    accS[0] = mfma16x16x4(o.a.x, o.b[0].x, accS[0]);
    accS[1] = mfma16x16x4(o.a.y, o.b[0].y, accS[1]);
    accS[0] = mfma16x16x4(o.a.z, o.b[1].x, accS[0]);
    accS[1] = mfma16x16x4(o.a.w, o.b[1].y, accS[1]);
}

// acc += W (16*NC x 8*NCH) * im2col (8*NCH x PT*16), software-pipelined one chunk ahead, four chunks
// per loop iteration so that every address is a loop-carried base plus an immediate.
// `o0` arrives with the chunk-0 WEIGHT fragments already loaded (prefetched across the previous
// layer's epilogue and barriers).  The prefetch of chunk NCH (one past the end) is a harmless
// over-read: weights continue into the bias block, LDS rows into the panel's slack row.
// With NPAIR > 0 the wave also accumulates its super-tile (accS): three
// chunk pairs of the K' = 8-shift contraction ride along with every four main chunks, on a 3-stage
// register ring (each pair is fetched a whole loop iteration before it is used).
// soff : wave-uniform byte offset of this layer's A fragments inside the packed weight buffer
// baddr: per position tile, LDS byte address of (row-2)*stride + 8*kq for this lane
template <int CTT, int C0, int NC, int PT, int NCH, int NPAIR>
__device__ __forceinline__ void conv_accumulate(f32x4 (&acc)[PT][NC], f32x4 (&accS)[2], Ops<NC, PT>& o0,
                                                __amdgpu_buffer_rsrc_t rsrc, uint32_t voff, uint32_t soff, uint32_t ssoff,
                                                const char* lds, const uint32_t (&baddr)[PT], uint32_t sbase) {
    constexpr uint32_t CSB = CTT * 512;   // bytes of A fragments per chunk
    constexpr bool SUP = NPAIR > 0;
    constexpr int NV = n_wloads(CTT, C0, NC), NM = 2 * PT * NC;
    Ops<NC, PT> o1;
    const lds_cptr lds3 = (lds_cptr)lds;
    lds_cptr cur[PT];        // loop-carried LDS pointers: one add per tile per 4 chunks, immediates otherwise
#pragma unroll
    for (int p = 0; p < PT; ++p) cur[p] = lds3 + baddr[p];
    lds_cptr scur = lds3 + sbase;
    SOps s0, s1, s2;
    if constexpr (SUP) {
        sload<0>(s0, rsrc, voff, ssoff, scur);
        sload<1>(s1, rsrc, voff, ssoff, scur);
        sload<2>(s2, rsrc, voff, ssoff, scur);
    }
    load_x<NC, PT, 0>(o0, cur);
    for (int it = 0; it < NCH / 4; ++it) {
        load_w<CTT, C0, NC, PT>(o1, rsrc, voff, soff + 1 * CSB);
        load_x<NC, PT, 32>(o1, cur);
        mma_chunk<NC, PT>(acc, o0);
        if constexpr (SUP) { smma(accS, s0); sload<3>(s0, rsrc, voff, ssoff, scur); }
        spread_loads<NV + (SUP ? 1 : 0), PT + (SUP ? 2 : 0), NM + (SUP ? 4 : 0)>();
        load_w<CTT, C0, NC, PT>(o0, rsrc, voff, soff + 2 * CSB);
        load_x<NC, PT, 64>(o0, cur);
        mma_chunk<NC, PT>(acc, o1);
        if constexpr (SUP) { smma(accS, s1); sload<4>(s1, rsrc, voff, ssoff, scur); }
        spread_loads<NV + (SUP ? 1 : 0), PT + (SUP ? 2 : 0), NM + (SUP ? 4 : 0)>();
        load_w<CTT, C0, NC, PT>(o1, rsrc, voff, soff + 3 * CSB);
        load_x<NC, PT, 96>(o1, cur);
        mma_chunk<NC, PT>(acc, o0);
        if constexpr (SUP) { smma(accS, s2); sload<5>(s2, rsrc, voff, ssoff, scur); }
        spread_loads<NV + (SUP ? 1 : 0), PT + (SUP ? 2 : 0), NM + (SUP ? 4 : 0)>();
        load_w<CTT, C0, NC, PT>(o0, rsrc, voff, soff + 4 * CSB);
        load_x<NC, PT, 128>(o0, cur);
        mma_chunk<NC, PT>(acc, o1);
        spread_loads<NV, PT, NM>();
        soff += 4 * CSB;
#pragma unroll
        for (int p = 0; p < PT; ++p) cur[p] += 128;
        if constexpr (SUP) {
            ssoff += 3 * 1024u;
            scur += 3 * 64;
        }
    }
    constexpr int TAIL = NCH % 4;     // o0 holds chunk NCH - TAIL; the ring holds pairs 3*(NCH/4) + {0,1,2}
    constexpr int DONE = 3 * (NCH / 4);
    if constexpr (TAIL >= 2) {
        load_w<CTT, C0, NC, PT>(o1, rsrc, voff, soff + 1 * CSB);
        load_x<NC, PT, 32>(o1, cur);
    }
    if constexpr (TAIL >= 1) mma_chunk<NC, PT>(acc, o0);
    if constexpr (SUP && DONE + 0 < NPAIR) { smma(accS, s0); if constexpr (DONE + 3 < NPAIR) sload<3>(s0, rsrc, voff, ssoff, scur); }
    if constexpr (TAIL >= 3) {
        load_w<CTT, C0, NC, PT>(o0, rsrc, voff, soff + 2 * CSB);
        load_x<NC, PT, 64>(o0, cur);
    }
    if constexpr (TAIL >= 2) mma_chunk<NC, PT>(acc, o1);
    if constexpr (SUP && DONE + 1 < NPAIR) { smma(accS, s1); if constexpr (DONE + 4 < NPAIR) sload<4>(s1, rsrc, voff, ssoff, scur); }
    if constexpr (TAIL >= 3) mma_chunk<NC, PT>(acc, o0);
    if constexpr (SUP && DONE + 2 < NPAIR) { smma(accS, s2); if constexpr (DONE + 5 < NPAIR) sload<5>(s2, rsrc, voff, ssoff, scur); }
    // drain the pairs the main chunks did not cover (2 for a U->U layer of U = 100)
    if constexpr (SUP && DONE + 3 < NPAIR) smma(accS, s0);
    if constexpr (SUP && DONE + 4 < NPAIR) smma(accS, s1);
    if constexpr (SUP && DONE + 5 < NPAIR) smma(accS, s2);
    static_assert(!SUP || NPAIR <= DONE + 6, "super-tile pairs must fit the main loop plus one drain round");
}


}  // namespace tae
