// Shared device building blocks of the MFMA kernels (implicit-GEMM K loop, fragment loads, ELU).
// Included by turboae_kernels.hip (CNN encoder / decoder) and turboae_gru.hip (GRU decoder).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// Timing-experiment bitmasks (the results become WRONG by construction; tools/build_variant.sh).  They exist only in builds made
// with -DTAE_EXPERIMENT; in every other build they are the constant 0 and the code they guard is compiled out.
//   TAE_X      (CNN kernels)           1 no layer barriers, 2 no panel writes, 4 no ELU, 8 no weight loads in loop, 16 no LDS reads in loop
//                                      1024 cycle stamps per layer phase of workgroup 0 (turboae_h2_impl.hpp stamp_h; results stay correct)
//   TAE_REC_X  (gru_rec_h)             1 no exp / rcp in the gates, 2 no LDS fragment reads, 4 no GI loads / Y0 stores, 8 no MFMAs
//   TAE_PROJ_X (gru_proj_h)            1 no GI stores, 2 no K loop, 4 no Y0 staging loads
#ifdef TAE_EXPERIMENT
#ifndef TAE_X
#define TAE_X 0
#endif
#ifndef TAE_REC_X
#define TAE_REC_X 0
#endif
#ifndef TAE_PROJ_X
#define TAE_PROJ_X 0
#endif
#else
#if defined(TAE_X) || defined(TAE_REC_X) || defined(TAE_PROJ_X)
#error "TAE_X / TAE_REC_X / TAE_PROJ_X are timing experiments that break the results: build with -DTAE_EXPERIMENT as well"
#endif
#define TAE_X 0
#define TAE_REC_X 0
#define TAE_PROJ_X 0
#endif

namespace tae {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using lds_cptr = const char __attribute__((address_space(3)))*;   // explicit LDS pointer (ds_* instructions)

// ELU(alpha=1) = x > 0 ? x : expm1(x)  (F.elu, cnn_utils.py:26,43) as med3(x, expm1(x), 0): expm1(x) >= x everywhere, so
// the median of {x, expm1(x), 0} is x for x > 0 and expm1(x) for x < 0 - no branch on the sign.  expm1 has to be RELATIVELY
// accurate, as torch's is: the reference's activations keep 24 bits at any scale, and exp2(x log2 e) - 1 alone (r01 - r03) has
// an absolute error of 6e-8 from the rounding of exp near 1 - invisible at O(1), but 1e-4 relative for activations of 1e-3 and a
// ReLU below 6e-8 (tests/test_gpu_range.py walks such networks).  So: |x| < 1/4 -> x (1 + x/2 + ... + x^5/720) (truncation
// x^6/5040 < 5e-8 relative), else exp2 - 1 (6e-8 absolute on a result > 0.22).  For x > 0 the polynomial side exceeds x and the
// median is x; x = +-inf and NaN pass through as in the reference.
__device__ __forceinline__ float expm1_poly(float x) {       // expm1(x) / x on |x| < 1/4
    float p = __builtin_fmaf(x, 1.0f / 720.0f, 1.0f / 120.0f);
    p = __builtin_fmaf(x, p, 1.0f / 24.0f);
    p = __builtin_fmaf(x, p, 1.0f / 6.0f);
    p = __builtin_fmaf(x, p, 0.5f);
    return __builtin_fmaf(x, p, 1.0f);
}
constexpr float kExpm1Switch = -0.25f;
__device__ __forceinline__ float elu1(float x) {
    const float small = x * expm1_poly(x);
    const float big = __builtin_amdgcn_exp2f(x * 1.44269504088896341f) - 1.0f;
    return __builtin_amdgcn_fmed3f(x, x > kExpm1Switch ? small : big, 0.0f);
}

// -enc_act / -dec_act (encoders.py:86-100, decoders.py:59-73): 0 elu, 1 linear, 2 tanh, 3 relu, 4 selu, 5 sigmoid.  Applied to a
// handful of scalars per position (the Linear heads' outputs), so the accurate library functions are used.
__device__ __forceinline__ float act_apply(float v, int act) {
    switch (act) {
        case 0: return elu1(v);
        case 2: return tanhf(v);
        case 3: return fmaxf(v, 0.0f);
        case 4: return 1.0507009873554804934f * (v > 0.0f ? v : 1.6732632423543772848f * expm1f(v));
        case 5: return 1.0f / (1.0f + expf(-v));
        default: return v;
    }
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


// ---- workgroup geometry and helpers shared by the fp32 and the f16x2 kernels ---------------------------
template <int U>
struct Geo {
    static constexpr int CT = (U + 15) / 16;         // 16-wide output-channel tiles
    static constexpr int CP = CT * 16;               // padded channel count
    static constexpr int NCH_MID = (5 * U + 7) / 8;  // K chunks (8 k each) of a U->U layer
    static constexpr int NCH_L0 = 5;                 // K chunks of the first layer (5 taps x 8 padded inputs)
    static constexpr int CS = CT * 128;              // floats of A fragments per chunk
    static constexpr int MIDF = NCH_MID * CS;        // floats of A fragments per U->U layer
    static constexpr int L0F = NCH_L0 * CS;
    // "super-tile" for the 4 remainder channels when U % 16 == 4 (see super_accumulate): A fragments with
    // rows = 4 position shifts x 4 channels over K' = 8 shifts x C_in, stored after the bias of every layer
    static constexpr bool SUP = (U % 16) == 4;
    static constexpr int SCH_MID = SUP ? U : 0;      // K' / 8 chunks of a U->U layer
    static constexpr int SCH_L0 = SUP ? 8 : 0;       // first layer: 8 shifts x 8 padded inputs
    static constexpr int SFM = SCH_MID * 128;        // floats of super A fragments per U->U layer
    static constexpr int SF0 = SCH_L0 * 128;
};

// Per-lane view of the position tiles a wave owns.
template <int PT>
struct TileCtx {
    int row[PT];      // panel row of this lane's position in tile p
    int rowbase[PT];  // panel row of position 0 of the same block
    int t[PT];        // index inside the block
    int blk[PT];      // block index inside the workgroup
    bool valid[PT];   // in-block row of the panel: its activations are written back
    bool center[PT];  // position whose stack output this workgroup owns (== valid for whole blocks)
};

// Reduce 8 per-lane partial outputs over the 4 lane groups (q = lane >> 4) of a position with a
// reduce-scatter butterfly: afterwards lane group q holds outputs f = 2q (k2[0]) and f = 2q + 1 (k2[1]).
__device__ __forceinline__ void butterfly8(const float (&part)[8], bool hi32, bool hi16, float (&k2)[2]) {
    float k4[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float keep = hi32 ? part[4 + j] : part[j];
        const float send = hi32 ? part[j] : part[4 + j];
        k4[j] = keep + __shfl_xor(send, 32);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const float keep = hi16 ? k4[2 + j] : k4[j];
        const float send = hi16 ? k4[j] : k4[2 + j];
        k2[j] = keep + __shfl_xor(send, 16);
    }
}

template <int PT>
__device__ __forceinline__ void make_tiles(TileCtx<PT>& tc, int g, int lane, int L, int npos) {
    const int n = lane & 15;
#pragma unroll
    for (int p = 0; p < PT; ++p) {
        const int m = (g * PT + p) * 16 + n;
        const bool v = m < npos;
        const int mm = v ? m : 0;
        const int b = mm / L;
        const int t = mm - b * L;
        tc.valid[p] = v;
        tc.center[p] = v;
        tc.blk[p] = b;
        tc.t[p] = t;
        tc.rowbase[p] = b * (L + 2) + 2;
        tc.row[p] = tc.rowbase[p] + t;
    }
}

__device__ __forceinline__ void zero_lds(char* smem, int bytes, int tid) {
    f32x4 z = {0.f, 0.f, 0.f, 0.f};
    for (int i = tid; i < bytes / 16; i += kThreads) reinterpret_cast<f32x4*>(smem)[i] = z;
}

// channel-tile split between the two waves of a SIMD: lower half [0, CTA), upper half [CTA, CT)
template <int U>
struct Split {
    static constexpr int CT = Geo<U>::CT;
    static constexpr int CTA = CT >= 3 ? 2 * ((CT + 2) / 4) : 1;   // even when possible, so 16-byte pair loads stay whole
    static constexpr int CTB = CT - CTA;
    static_assert(CTB >= 1, "both channel halves need at least one tile");
    // with super-tiles the padded last tile disappears and each half takes one super-tile: split CT - 1 evenly
    static constexpr int SA = (CT - 1 + 1) / 2;
    static constexpr int SB = CT - 1 - SA;
};

__device__ __forceinline__ void block_reduce_stats(char* smem, int tid, double sum, double sumsq, double* partials) {
    // deterministic fixed-order tree: xor butterfly inside each wave (registers), then the 8 wave totals in wave order; the
    // panels are dead after the last stack's closing barrier, so the scratch aliases them (no static LDS: guide G17)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        sum += __shfl_xor(sum, off);
        sumsq += __shfl_xor(sumsq, off);
    }
    double* red = reinterpret_cast<double*>(smem);
    if ((tid & 63) == 0) { red[tid >> 6] = sum; red[kWaves + (tid >> 6)] = sumsq; }
    __syncthreads();
    if (tid == 0) {
        double a = red[0], b = red[kWaves];
#pragma unroll
        for (int w = 1; w < kWaves; ++w) { a += red[w]; b += red[kWaves + w]; }
        partials[2 * blockIdx.x] = a;
        partials[2 * blockIdx.x + 1] = b;
    }
}


// =====================================================================================================
// fp16-split contraction ("f16x2"): every fp32 operand x is carried as hi = f16(x), lo = f16(x - hi)
// (both round-to-nearest; fp16 denormals are kept by the matrix pipe, measured in
// tools/probes/f16_split_probe.hip), and a product is the sum of three v_mfma_f32_16x16x32_f16:
// hi*lo + lo*hi + hi*hi, accumulated in fp32.  hi + lo represents x to 2^-22 relative (absolute floor
// 2^-25); the dropped lo*lo term is 2^-22 of the product.  Against an fp64 reference the 500-term dot
// products of this network come out with rms error 2.0e-7 vs 3.2e-7 for the v_mfma_f32_16x16x4_f32
// chain (fewer fp32 roundings: 48 accumulations instead of 125), i.e. the result is fp32-grade while the
// matrix pipe runs 16x32-deep f16 instructions in 16 cycles instead of 16x4-deep f32 ones in 32:
// 3 MFMAs / 16 cycles per 32 k  vs  8 MFMAs / 32 cycles  ->  5.3x fewer pipe cycles.
// Weights are pre-scaled per layer by a power of two (max |w| -> [2^13, 2^14)) so that their lo halves
// stay normal; the accumulators carry that scale (bias pre-scaled) and the epilogue multiplies it out.
// Order of the three products of a tile (A/B on one box, r03): 1 = per channel tile (hi,lo) (hi,hi) (lo,hi) back to back, so consecutive
// MFMAs share an operand register set - 1.2-1.9 % faster than 0 = all (hi,lo), then all (lo,hi), then all (hi,hi) on this power-limited
// kernel (the same MFMAs, less operand switching); 2 = as 1 with odd tiles reversed (every consecutive pair shares an operand): no
// further gain.
#ifndef TAE_MMA_ORDER
#define TAE_MMA_ORDER 1
#endif
using h8 = __attribute__((ext_vector_type(8))) _Float16;
using h4 = __attribute__((ext_vector_type(4))) _Float16;
using u32x2v = __attribute__((ext_vector_type(2))) uint32_t;
using u32x4w = __attribute__((ext_vector_type(4))) uint32_t;

__device__ __forceinline__ f32x4 mfma16x16x32h(h8 a, h8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

constexpr float kH2Limit = 65504.0f;     // activations are clamped to the fp16 range before the split

// 4 fp32 values -> 4 hi halves + 4 lo halves.  Three instructions per value pair: v_cvt_pk_f16_f32 (both hi halves,
// round to nearest), then v_fma_mixlo_f16 / v_fma_mixhi_f16, which read the f16 hi half as fp32, compute v - hi exactly
// in fp32 and round once to f16.  (Left to the compiler the same arithmetic takes eight: it converts hi twice and back.)
__device__ __forceinline__ void split4(f32x4 v, h4& hi, h4& lo) {
    uint32_t h01, h23, l01, l23;
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h01) : "v"(v.x), "v"(v.y));
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h23) : "v"(v.z), "v"(v.w));
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l01) : "v"(h01), "v"(v.x));
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l23) : "v"(h23), "v"(v.z));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l01) : "v"(h01), "v"(v.y));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l23) : "v"(h23), "v"(v.w));
    hi = __builtin_bit_cast(h4, u32x2v{h01, h23});
    lo = __builtin_bit_cast(h4, u32x2v{l01, l23});
}

// A fragments of one 32-k slab for NC channel tiles: hi and lo halves, 16 bytes per lane each
template <int NC>
struct OpsHA {
    h8 hi[NC], lo[NC];
};

// packed slab layout: [channel tile][hi | lo][lane][8 halves] = 2 KB per tile
template <int CTT, int C0, int NC>
__device__ __forceinline__ void load_wh(OpsHA<NC>& o, __amdgpu_buffer_rsrc_t rsrc, uint32_t voff, uint32_t soff) {
    if (TAE_X & 8) { asm volatile("" :: "s"(soff)); return; }
    // hi halves first: the first product of a tile (hi * lo) needs them, the lo halves one product later
#pragma unroll
    for (int i = 0; i < NC; ++i) o.hi[i] = __builtin_bit_cast(h8, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff + (C0 + i) * 2048, soff, 0));
#pragma unroll
    for (int i = 0; i < NC; ++i) o.lo[i] = __builtin_bit_cast(h8, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff + (C0 + i) * 2048 + 1024, soff, 0));
    if (TAE_X & 256) {      // timing experiment: ~+25 % weight-load traffic (two extra 1 KB loads per slab, results discarded)
        const auto d0 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff + ((C0 + 1) % CTT) * 2048 + 512, soff, 0);
        const auto d1 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff + ((C0 + 2) % CTT) * 2048 + 1536, soff, 0);
        asm volatile("" :: "v"(d0), "v"(d1));
    }
}

struct OpsHB {
    h8 hi, lo;
};

// B fragments of one position tile: 8 consecutive halves of the im2col row from each plane (8-byte aligned)
// B128: the rows are 16-byte aligned (the projection GEMMs' panels), one ds_read_b128 per plane - 4 LDS cycles per wave-instruction
// against 8 for the ds_read2_b64 the 8-byte form compiles to, and banked mod 64 instead of mod 32
template <int OFF, bool B128 = false>
__device__ __forceinline__ void load_xh(OpsHB& o, lds_cptr ph, lds_cptr pl) {
    using lds_u2 = const u32x2v __attribute__((address_space(3)));
    if (TAE_X & 16) return;
    if constexpr (B128) {
        using lds_u4 = const u32x4w __attribute__((address_space(3)));
        o.hi = __builtin_bit_cast(h8, *reinterpret_cast<lds_u4*>(ph + OFF));
        o.lo = __builtin_bit_cast(h8, *reinterpret_cast<lds_u4*>(pl + OFF));
        return;
    }
    const u32x2v a0 = *reinterpret_cast<lds_u2*>(ph + OFF), a1 = *reinterpret_cast<lds_u2*>(ph + OFF + 8);
    const u32x2v b0 = *reinterpret_cast<lds_u2*>(pl + OFF), b1 = *reinterpret_cast<lds_u2*>(pl + OFF + 8);
    o.hi = __builtin_bit_cast(h8, u32x4w{a0.x, a0.y, a1.x, a1.y});
    o.lo = __builtin_bit_cast(h8, u32x4w{b0.x, b0.y, b1.x, b1.y});
    if (TAE_X & 512) {      // timing experiment: +100 % LDS operand reads (results discarded)
        const u32x2v c0 = *reinterpret_cast<lds_u2*>(ph + OFF + 16), c1 = *reinterpret_cast<lds_u2*>(ph + OFF + 24);
        const u32x2v d0 = *reinterpret_cast<lds_u2*>(pl + OFF + 16), d1 = *reinterpret_cast<lds_u2*>(pl + OFF + 24);
        asm volatile("" :: "v"(c0), "v"(c1), "v"(d0), "v"(d1));
    }
}

// PROD = 3: the fp32-grade scheme (hi*lo + lo*hi + hi*hi).  PROD = 1: the hi halves only - a plain fp16 contraction with fp32
// accumulation, NOT fp32-grade; instantiated for the separately labelled `precision = f16x1` decoder line only.
template <int NC, int PROD = 3>
__device__ __forceinline__ void mma_tile_h(f32x4 (&acc)[NC], const OpsHA<NC>& a, const OpsHB& b) {
    if constexpr (PROD == 1) {
#pragma unroll
        for (int ct = 0; ct < NC; ++ct) acc[ct] = mfma16x16x32h(a.hi[ct], b.hi, acc[ct]);
        return;
    }
#if TAE_MMA_ORDER == 2
    // experiment: as order 1, odd tiles reversed, so EVERY consecutive pair of MFMAs shares one operand register set
#pragma unroll
    for (int ct = 0; ct < NC; ++ct) {
        if (ct & 1) {
            acc[ct] = mfma16x16x32h(a.lo[ct], b.hi, acc[ct]);
            acc[ct] = mfma16x16x32h(a.hi[ct], b.hi, acc[ct]);
            acc[ct] = mfma16x16x32h(a.hi[ct], b.lo, acc[ct]);
        } else {
            acc[ct] = mfma16x16x32h(a.hi[ct], b.lo, acc[ct]);
            acc[ct] = mfma16x16x32h(a.hi[ct], b.hi, acc[ct]);
            acc[ct] = mfma16x16x32h(a.lo[ct], b.hi, acc[ct]);
        }
    }
#elif TAE_MMA_ORDER == 1
    // experiment: per channel tile the three products back to back (consecutive MFMAs share an operand register set)
#pragma unroll
    for (int ct = 0; ct < NC; ++ct) {
        acc[ct] = mfma16x16x32h(a.hi[ct], b.lo, acc[ct]);
        acc[ct] = mfma16x16x32h(a.hi[ct], b.hi, acc[ct]);
        acc[ct] = mfma16x16x32h(a.lo[ct], b.hi, acc[ct]);
    }
#else
#pragma unroll
    for (int ct = 0; ct < NC; ++ct) acc[ct] = mfma16x16x32h(a.hi[ct], b.lo, acc[ct]);
#pragma unroll
    for (int ct = 0; ct < NC; ++ct) acc[ct] = mfma16x16x32h(a.lo[ct], b.hi, acc[ct]);
#pragma unroll
    for (int ct = 0; ct < NC; ++ct) acc[ct] = mfma16x16x32h(a.hi[ct], b.hi, acc[ct]);
#endif
}

// The 20-k tail slab of a 100-channel, 5-tap layer (K = 500 = 15 x 32 + 20; late r06): its three products are 60 k slots - TWO
// MFMAs instead of a padded slab's three.  In quarters of a lane's 8 k slots, lane group kq:
//     MFMA 1   A1 = [hi k 4kq.. | hi k 4kq..]        B1 = [hi k 4kq.. | lo k 4kq..]                      hi*hi and hi*lo of k 0..15
//     MFMA 2   A2 = [lo k 4kq.. | X(kq)]              B2 = [hi k 4kq.. | Y(kq)]                           lo*hi of k 0..15, and k 16..19:
//              X = hi, hi, lo, 0 (kq = 0..3) of k 16..19;  Y = hi, lo, hi, (hi) of k 16..19
// The host stores A1 where the slab's hi fragments go and A2 in place of its lo fragments (pack_conv_h, tail20), so the weight stream
// does not change; the B operands are three 8-byte reads per tile (H = hi plane k 4kq.., Lq = lo plane k 4kq.., Y).
template <int NC>
__device__ __forceinline__ void mma_tile_h_tail20(f32x4 (&acc)[NC], const OpsHA<NC>& a, const OpsHB& b) {
#pragma unroll
    for (int ct = 0; ct < NC; ++ct) acc[ct] = mfma16x16x32h(a.hi[ct], b.hi, acc[ct]);
#pragma unroll
    for (int ct = 0; ct < NC; ++ct) acc[ct] = mfma16x16x32h(a.lo[ct], b.lo, acc[ct]);
}
// ph / pl: the lane's fragment addresses of the slab BEFORE the tail slab in the hi / lo plane (row + 32 slab + 16 kq bytes) MINUS 8 kq
// (conv_accumulate_h moves its pointers there once they have served the last full slab): k = 4 kq of the tail slab is 64 bytes on,
// k = 16 another 32 - 8 kq, in the lo plane for lane group 1
__device__ __forceinline__ void load_xh_tail20(OpsHB& o, lds_cptr ph, lds_cptr pl, uint32_t voff) {
    using lds_u2 = const u32x2v __attribute__((address_space(3)));
    const uint32_t k8 = (voff >> 5) & ~7u;                            // voff = lane * 16: 8 kq
    const lds_cptr y = ((voff >> 8) == 1u ? pl : ph) - k8;
    const u32x2v H = *reinterpret_cast<lds_u2*>(ph + 64), Lq = *reinterpret_cast<lds_u2*>(pl + 64), Y = *reinterpret_cast<lds_u2*>(y + 96);
    o.hi = __builtin_bit_cast(h8, u32x4w{H.x, H.y, Lq.x, Lq.y});
    o.lo = __builtin_bit_cast(h8, u32x4w{H.x, H.y, Y.x, Y.y});
}

// acc += W (16*NC x 32*NSLAB) * im2col (32*NSLAB x PT*16) in the f16x2 representation.
// `a0` arrives with slab 0's A fragments loaded.  Slabs are processed in pairs (A fragments ping-pong
// between two register sets, fetched one slab = ~1000 cycles ahead); within a slab the position tiles are
// walked one at a time, their B fragments fetched two tiles ahead from LDS.
// soff  : wave-uniform byte offset of this layer's A fragments
// bh/bl : per position tile, LDS byte address of (row-2)*stride + 16*kq in the hi / lo plane
// NSLAB = 0: the slab count is the (wave-uniform) run-time argument `nslab_rt` (conv stacks: it follows the kernel size).
// TAIL20: `nslab_rt` is ODD and counts the full slabs; one more slab follows them in the packed layer, in the two-MFMA tail form above
// (a compile-time property of the instantiation: with the choice at run time the allocator spilled 57 vector registers of the decoder).
template <int CTT, int C0, int NC, int PT, int NSLAB, bool B128 = false, int PROD = 3, bool TAIL20 = false>
__device__ __forceinline__ void conv_accumulate_h(f32x4 (&acc)[PT][NC], OpsHA<NC>& a0, __amdgpu_buffer_rsrc_t rsrc, uint32_t voff,
                                                  uint32_t soff, const char* lds, const uint32_t (&bh)[PT], const uint32_t (&bl)[PT],
                                                  int nslab_rt = 0) {
    constexpr uint32_t SB = CTT * 2048;       // bytes of A fragments per slab
    const int nslab = NSLAB > 0 ? NSLAB : nslab_rt;
    const lds_cptr lds3 = (lds_cptr)lds;
    lds_cptr ch[PT], cl[PT];
#pragma unroll
    for (int p = 0; p < PT; ++p) { ch[p] = lds3 + bh[p]; cl[p] = lds3 + bl[p]; }
    OpsHA<NC> a1;
    OpsHB b[2];                                // ring: tile j of the running (slab, tile) sequence sits in b[j % 2]
    load_xh<0, B128>(b[0], ch[0], cl[0]);
    // One slab.  Tile p uses ring slot (RB + p) % 2 and prefetches the next tile (of this slab at OFF, or tile 0 of
    // the next slab at OFF + 64): one tile = 3 * NC MFMAs (>= 150 cycles) of cover for the LDS latency.  The issue
    // order is pinned with sched_group_barriers - left alone, the scheduler sinks every load to just before its
    // first use (LDS reads then wait with lgkmcnt(0) in front of each tile, weight loads lose their one-slab lead):
    // per tile {LDS reads of the next tile} first, then this tile's MFMAs with the slab's 2 * NC weight loads (for
    // the NEXT slab) dealt out one per three MFMAs.
    constexpr int NVS = PT < 3 ? PT : 3;               // the weight loads go out during the first NVS tiles of a slab (lead >= 2 tiles)
    constexpr int NVT = (2 * NC + NVS - 1) / NVS;      // weight loads threaded through one of those tiles' MFMAs
#define TAE_H_SLAB(ACUR, OFF, RB)                                                                          \
    _Pragma("unroll") for (int p = 0; p < PT; ++p) {                                                       \
        if (p + 1 < PT) load_xh<(OFF), B128>(b[((RB) + p + 1) % 2], ch[p + 1 < PT ? p + 1 : 0], cl[p + 1 < PT ? p + 1 : 0]);   \
        else load_xh<(OFF) + 64, B128>(b[((RB) + p + 1) % 2], ch[0], cl[0]);                                     \
        mma_tile_h<NC, PROD>(acc[p], ACUR, b[((RB) + p) % 2]);                                                   \
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);                                                 \
        if (p < NVS) {                                                                                     \
            _Pragma("unroll") for (int v = 0; v < NVT; ++v) {                                              \
                __builtin_amdgcn_sched_group_barrier(0x008, PROD * NC / NVT, 0);                              \
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                                         \
            }                                                                                              \
            __builtin_amdgcn_sched_group_barrier(0x008, 3 * NC - NVT * (PROD * NC / NVT), 0);                 \
        } else {                                                                                           \
            __builtin_amdgcn_sched_group_barrier(0x008, PROD * NC, 0);                                        \
        }                                                                                                  \
    }
    int s = 0;
#pragma unroll 1
    for (; s + 1 < nslab; s += 2) {
        load_wh<CTT, C0, NC>(a1, rsrc, voff, soff + SB);
        TAE_H_SLAB(a0, 0, 0)
        if (s + 2 < nslab) load_wh<CTT, C0, NC>(a0, rsrc, voff, soff + 2 * SB);   // no fetch past the layer: the caller refills a0
        TAE_H_SLAB(a1, 64, PT % 2)
        soff += 2 * SB;
#pragma unroll
        for (int p = 0; p < PT; ++p) { ch[p] += 128; cl[p] += 128; }
    }
    if constexpr (TAIL20) {
        {
            // the last full slab (a0, fetched by the loop's last pair) with the tail slab's fragments and first B tile on their way,
            // then the tail slab: per tile three 8-byte LDS reads (of the next tile) and 2 * NC MFMAs
            load_wh<CTT, C0, NC>(a1, rsrc, voff, soff + SB);
            _Pragma("unroll") for (int p = 0; p + 1 < PT; ++p) {
                load_xh<0, B128>(b[(p + 1) % 2], ch[p + 1], cl[p + 1]);
                mma_tile_h<NC, PROD>(acc[p], a0, b[p % 2]);
                __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
                if (p < NVS) {
                    _Pragma("unroll") for (int v = 0; v < NVT; ++v) {
                        __builtin_amdgcn_sched_group_barrier(0x008, PROD * NC / NVT, 0);
                        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x008, 3 * NC - NVT * (PROD * NC / NVT), 0);
                } else {
                    __builtin_amdgcn_sched_group_barrier(0x008, PROD * NC, 0);
                }
            }
            {   // last tile of the last full slab: the tail slab's first B tile is fetched behind the products that end the hi fragments'
                // life (sixteen registers fewer at the kernel's tightest point than fetching it ahead of the tile)
                constexpr int p = PT - 1;
                const OpsHB& bb = b[p % 2];
                static_assert(TAE_MMA_ORDER == 1, "the split below keeps mma_tile_h's per-accumulator product order (hi*lo, hi*hi, lo*hi)");
                if constexpr (PROD == 3) {
#pragma unroll
                    for (int ct = 0; ct < NC; ++ct) {
                        acc[p][ct] = mfma16x16x32h(a0.hi[ct], bb.lo, acc[p][ct]);
                        acc[p][ct] = mfma16x16x32h(a0.hi[ct], bb.hi, acc[p][ct]);
                    }
                }
                const uint32_t k8 = (voff >> 5) & ~7u;
                _Pragma("unroll") for (int r = 0; r < PT; ++r) { ch[r] -= k8; cl[r] -= k8; }      // the full slabs are done with them
                load_xh_tail20(b[(p + 1) % 2], ch[0], cl[0], voff);
#pragma unroll
                for (int ct = 0; ct < NC; ++ct) acc[p][ct] = PROD == 3 ? mfma16x16x32h(a0.lo[ct], bb.hi, acc[p][ct]) : mfma16x16x32h(a0.hi[ct], bb.hi, acc[p][ct]);
                __builtin_amdgcn_sched_group_barrier(0x008, (PROD - 1) * NC, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, NC, 0);
            }
            _Pragma("unroll") for (int p = 0; p < PT; ++p) {
                if (p + 1 < PT) load_xh_tail20(b[(PT + p + 1) % 2], ch[p + 1 < PT ? p + 1 : 0], cl[p + 1 < PT ? p + 1 : 0], voff);
                mma_tile_h_tail20<NC>(acc[p], a1, b[(PT + p) % 2]);
                __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 2 * NC, 0);
            }
            return;
        }
    }
    if (nslab & 1) { TAE_H_SLAB(a0, 0, 0) }
#undef TAE_H_SLAB
}

}  // namespace tae
