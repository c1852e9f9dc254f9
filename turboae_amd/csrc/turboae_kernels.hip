// TurboAE rate-1/3 CNN hot path for MI355X (gfx950 / CDNA4) - device kernels.
//
// What the reference computes on this path (all fp32):
//   SameShapeConv1d.forward   cnn_utils.py:36-46   x = ELU(conv1d(x, W_l, b_l, pad=2)) per layer
//   ENC_interCNN.forward      encoders.py:351-377  3 conv stacks + Linear(U->1) + ELU, branch 3 interleaved
//   power_constraint          encoders.py:102-125  (x - mean) / std over the whole batch
//   Channel_AE.forward        channel_ae.py:38-42  received = codes + noise
//   DEC_LargeCNN.forward      decoders.py:206-269  2*num_iteration conv stacks + Linear(U->F|1),
//                                                  extrinsic subtract, (de)interleave, sigmoid
//   Interleaver/DeInterleaver interleavers.py:15-21,43-48
//
// MI355X design (not a translation of the PyTorch graph):
//  * One workgroup (4 waves, one per SIMD) owns `nb` whole codeword blocks and keeps ALL their
//    activations in LDS for the whole network: the (rows x U) activation panel is updated IN PLACE
//    layer after layer (accumulators live in registers while the panel is being read, then ELU
//    results overwrite it), so a full 6-iteration decode touches HBM only for its 12 B/bit input,
//    4 B/bit output and the L2-resident weights.
//  * Each conv layer is an implicit GEMM on the exact-fp32 matrix cores
//    (v_mfma_f32_16x16x4_f32): A = weights (M = output channel), B = activations (N = position),
//    K = 5 taps x U channels.  Rows are stored position-major with U contiguous channels and NO row
//    padding, so the im2col row of position t is simply the 5*U contiguous floats starting at row
//    t-2: every B fragment is one ds_read_b64 at (row-2)*U*4 + 32*chunk + 8*kq, bank-conflict
//    free for U=100 (stride 100 dwords -> 18*n mod 32 distinct even slots).
//  * Weights are pre-tiled on the host into MFMA A-fragment order, so a wave fetches a fragment with
//    one fully coalesced 512-B global_load_dwordx2; all workgroups stream the same ~225 KB layer at
//    about the same time, so it is served from L2/L1.
//  * Zero padding (Conv1d padding=2) is two permanently-zero rows between consecutive blocks in the
//    panel; interleaving is an LDS scatter of the 5 extrinsic values per position between the two
//    8-float-per-row input panels XA (natural order) and XB (interleaved order).
//  * Linear(U->F) + bias + extrinsic subtraction + (de)interleave are fused into the last conv
//    layer's epilogue straight from the accumulators (the last layer never touches the panel).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "turboae_internal.hpp"
#include "philox.hpp"

namespace tae {

using f32x4 = __attribute__((ext_vector_type(4))) float;
// constant address space: wave-uniform loads through it become scalar (s_load) instructions
using cfloat = const float __attribute__((address_space(4)));

// ELU(alpha=1) = x > 0 ? x : expm1(x)  (F.elu, cnn_utils.py:26,43).  Branch-free expm1 for x <= 0:
// degree-5 Taylor for x >= -0.125 (truncation x^5/720 < 5e-8 relative), exp(x) - 1 below that, where
// |result| >= 0.1175 so the cancellation costs < 6e-7 relative (absolute error <= 7e-8 everywhere).
// 12 VALU ops, no divergence.  Large positive x may produce inf in the discarded branches, never NaN.
__device__ __forceinline__ float elu1(float x) {
    float p = fmaf(x, 8.3333333e-3f, 4.1666667e-2f);   // 1/5!, 1/4!
    p = fmaf(p, x, 1.6666667e-1f);                     // 1/3!
    p = fmaf(p, x, 0.5f);
    p = fmaf(p, x, 1.0f);
    p = p * x;
    const float e = __builtin_amdgcn_exp2f(x * 1.44269504088896341f) - 1.0f;
    const float neg = x < -0.125f ? e : p;
    return x > 0.0f ? x : neg;
}

__device__ __forceinline__ f32x4 mfma16x16x4(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// 16 independent 4x4 outer products: lane 4*blk + i supplies A[blk][i] and B[blk][i];
// lane 4*blk + j receives D[blk][0..3][j].
__device__ __forceinline__ f32x4 mfma4x4x1(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
}

// Build-time experiment switches (defaults = shipped configuration).
#ifndef TAE_VALU_REM
#define TAE_VALU_REM 0      // 0 (shipped): remainder channels (U % 16) ride a padded extra 16x16 tile; 1: on 4x4x1 MFMAs (measured slower: ~44 cycles per v_mfma_f32_4x4x1_16b_f32 on gfx950)
#endif

template <int U>
struct Geo {
    static constexpr int CTM = TAE_VALU_REM ? U / 16 : (U + 15) / 16;   // 16-wide output-channel tiles on the matrix cores
    static constexpr int VCH = TAE_VALU_REM ? U - 16 * (U / 16) : 0;    // remainder channels, computed with 4x4x1 MFMAs
    static constexpr int CP = ((U + 15) / 16) * 16;  // padded channel count (bias / Linear head rows)
    static constexpr int NCH_MID = (5 * U + 7) / 8;  // K chunks (8 k each) of a U->U layer
    static constexpr int NCH_L0 = 5;                 // K chunks of the first layer (5 taps x 8 padded inputs)
    static constexpr int MIDF = NCH_MID * CTM * 128; // floats of A fragments per U->U layer
    static constexpr int L0F = NCH_L0 * CTM * 128;
    static constexpr int MIDR = NCH_MID * VCH * 8;   // floats of remainder-channel weights [chunk][VCH][8]
    static constexpr int L0R = NCH_L0 * VCH * 8;
    static_assert(VCH == 0 || VCH == 4, "remainder path = one 4-row MFMA block");
};

constexpr int kWaves = 4;
constexpr int kThreads = 64 * kWaves;
constexpr int kXW = 8;  // floats per row of the XA / XB input panels

// Operands of one K-chunk (8 k values = 2 MFMA k-steps) for a wave's PT x CTM grid of 16x16 tiles,
// plus the remainder channels: U = 16*CTM + VCH, and the VCH = 4 remainder channels (96..99 for
// U = 100) would waste 75 % of a 7th 16x16 tile, so they run on v_mfma_f32_4x4x1_16b_f32 instead
// (16 independent 4x4 outer products per instruction, K = 1, 2 passes): the 4 channels are the A
// rows of every block and each lane brings its OWN position as a B column, so one instruction
// covers 64 positions x 4 channels at full matrix-pipe efficiency:
//   group A: lane l owns position 16*(l>>4) + (l&15) of tiles 0..3;
//   group B: lane l owns position (l&15) of tile PT-1 (the four lane groups duplicate each other).
// Both groups run the same k-ordered chain, so a block's result does not depend on where in a
// workgroup it sits.  (Measured alternative: the same FMAs on the vector ALU cost ~8 matrix-pipe
// cycles each - fp32 VALU FMAs do not overlap fp32 MFMAs - and were a net loss.)
template <int CTM, int PT, int VCH>
struct Ops {
    float2 a[CTM];
    float2 b[PT];
    f32x4 xa[2], xb[2];      // 8 consecutive im2col values of the lane's group-A / group-B position
    f32x4 wr[2];             // 8 consecutive weights of remainder channel (lane & 3)
};

template <int VCH>
struct VAddr {
    uint32_t a, b;            // LDS byte address of (row-2)*stride for the lane's group-A / group-B position
    const float* remv;        // per lane: remainder weights [chunk][VCH][8] + (lane & 3) * 8
};

// A fragments (global, L2-resident weights) of chunk c
template <int CTM, int PT, int VCH>
__device__ __forceinline__ void load_w(Ops<CTM, PT, VCH>& o, const float2* __restrict__ wf, const VAddr<VCH>& va, int c) {
#pragma unroll
    for (int ct = 0; ct < CTM; ++ct) o.a[ct] = wf[(c * CTM + ct) * 64];
    if constexpr (VCH > 0) {
        o.wr[0] = *reinterpret_cast<const f32x4*>(va.remv + c * VCH * 8);
        o.wr[1] = *reinterpret_cast<const f32x4*>(va.remv + c * VCH * 8 + 4);
    }
}

// B fragments (LDS activations) of chunk c
template <int CTM, int PT, int VCH>
__device__ __forceinline__ void load_x(Ops<CTM, PT, VCH>& o, const char* lds, const uint32_t (&baddr)[PT],
                                       const VAddr<VCH>& va, int c) {
#pragma unroll
    for (int p = 0; p < PT; ++p) o.b[p] = *reinterpret_cast<const float2*>(lds + baddr[p] + 32u * (uint32_t)c);
    if constexpr (VCH > 0) {
        o.xa[0] = *reinterpret_cast<const f32x4*>(lds + va.a + 32u * (uint32_t)c);
        o.xa[1] = *reinterpret_cast<const f32x4*>(lds + va.a + 32u * (uint32_t)c + 16u);
        o.xb[0] = *reinterpret_cast<const f32x4*>(lds + va.b + 32u * (uint32_t)c);
        o.xb[1] = *reinterpret_cast<const f32x4*>(lds + va.b + 32u * (uint32_t)c + 16u);
    }
}

template <int CTM, int PT, int VCH>
__device__ __forceinline__ void mma_chunk(f32x4 (&acc)[PT][CTM], f32x4 (&accA)[8], f32x4 (&accB)[8], const Ops<CTM, PT, VCH>& o) {
#pragma unroll
    for (int p = 0; p < PT; ++p)
#pragma unroll
        for (int ct = 0; ct < CTM; ++ct) acc[p][ct] = mfma16x16x4(o.a[ct].x, o.b[p].x, acc[p][ct]);
    if constexpr (VCH > 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            accA[k] = mfma4x4x1(o.wr[0][k], o.xa[0][k], accA[k]);
            accB[k] = mfma4x4x1(o.wr[0][k], o.xb[0][k], accB[k]);
        }
    }
#pragma unroll
    for (int p = 0; p < PT; ++p)
#pragma unroll
        for (int ct = 0; ct < CTM; ++ct) acc[p][ct] = mfma16x16x4(o.a[ct].y, o.b[p].y, acc[p][ct]);
    if constexpr (VCH > 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            accA[4 + k] = mfma4x4x1(o.wr[1][k], o.xa[1][k], accA[4 + k]);
            accB[4 + k] = mfma4x4x1(o.wr[1][k], o.xb[1][k], accB[4 + k]);
        }
    }
}

// acc += W (16*CTM x 8*nch) * im2col (8*nch x PT*16) (+ the remainder channels), software-pipelined
// one chunk ahead.  `o0` arrives with the chunk-0 WEIGHT fragments already loaded (prefetched across
// the previous layer's epilogue and barriers).
// wf   : this lane's pointer into the layer's A fragments ([chunk][ct][lane] float2)
// baddr: per position tile, LDS byte address of (row-2)*stride + 8*kq for this lane
template <int CTM, int PT, int VCH>
__device__ __forceinline__ void conv_accumulate(f32x4 (&acc)[PT][CTM], f32x4 (&accA)[8], f32x4 (&accB)[8], Ops<CTM, PT, VCH>& o0,
                                                const float2* __restrict__ wf, const char* lds,
                                                const uint32_t (&baddr)[PT], const VAddr<VCH>& va, int nch) {
    Ops<CTM, PT, VCH> o1;
    load_x<CTM, PT, VCH>(o0, lds, baddr, va, 0);
    int c = 0;
    for (; c + 2 <= nch; c += 2) {
        load_w<CTM, PT, VCH>(o1, wf, va, c + 1);
        load_x<CTM, PT, VCH>(o1, lds, baddr, va, c + 1);
        mma_chunk<CTM, PT, VCH>(acc, accA, accB, o0);
        const int cn = (c + 2 < nch) ? c + 2 : nch - 1;   // clamp: harmless re-load of the last chunk
        load_w<CTM, PT, VCH>(o0, wf, va, cn);
        load_x<CTM, PT, VCH>(o0, lds, baddr, va, cn);
        mma_chunk<CTM, PT, VCH>(acc, accA, accB, o1);
    }
    if (c < nch) mma_chunk<CTM, PT, VCH>(acc, accA, accB, o0);
}

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

// Weight-side state a stack hands to the next one: chunk-0 A fragments of the NEXT conv layer,
// fetched before the current layer's epilogue so their L2 latency hides behind it.
template <int U, int PT>
struct Prefetch {
    Ops<Geo<U>::CTM, PT, Geo<U>::VCH> o;
};

// Runs one SameShapeConv1d stack (cnn_utils.py:36-46) followed by its Linear head for the
// workgroup's blocks.  `epi(p, f, value)` is called for output feature f (0..7) of this lane's
// position in tile p by the lane that owns (p, f): lane group q owns f = 2q and f = 2q + 1.
// `wnext` = packed base of the stack that runs next in this kernel (nullptr if none): its first
// layer's chunk-0 weights are prefetched into `pf` before the head epilogue.
//
// Packed stack layout (floats), written by turboae_api.hip::pack_stack:
//   per layer: A fragments [chunk][CTM][64][2] | bias [CP] | remainder weights [chunk][VCH][8]
//   then Linear head weights [8][CP] | Linear bias [8]
template <int U, int PT, class Epi>
__device__ __forceinline__ void run_stack(const float* __restrict__ wstack, const float* __restrict__ wnext, int n_layer,
                                          char* smem, float* ACT, const float* Xin, const TileCtx<PT>& tc, int lane,
                                          Prefetch<U, PT>& pf, Epi epi) {
    using G = Geo<U>;
    constexpr int CTM = G::CTM, VCH = G::VCH;
    static_assert(VCH == 0 || PT == 5, "remainder-channel mapping assumes 4 + 1 position tiles per wave");
    const int q = lane >> 4;
    f32x4 acc[PT][CTM];
    f32x4 accA = {0.f, 0.f, 0.f, 0.f}, accB = {0.f, 0.f, 0.f, 0.f};
    f32x4 accA8[8], accB8[8];   // one accumulator per k mod 8: no back-to-back dependent 4x4x1 MFMAs
    // group-A position of this lane = its position in tile q; group-B position = its position in tile PT-1
    int rowA = tc.row[0];
    bool validA = tc.valid[0];
    if constexpr (VCH > 0) {
        rowA = (q == 0) ? tc.row[0] : (q == 1) ? tc.row[1] : (q == 2) ? tc.row[2] : tc.row[3];
        validA = (q == 0) ? tc.valid[0] : (q == 1) ? tc.valid[1] : (q == 2) ? tc.valid[2] : tc.valid[3];
    }
    const float* wl = wstack;
    for (int l = 0; l < n_layer; ++l) {
        const bool first = (l == 0);
        const int fragf = first ? G::L0F : G::MIDF;
        const int remf = first ? G::L0R : G::MIDR;
        const float* bias = wl + fragf;
        // accumulators start at the bias (Conv1d bias=True, cnn_utils.py:15-17)
        {
            f32x4 b4[CTM];
#pragma unroll
            for (int ct = 0; ct < CTM; ++ct) b4[ct] = *reinterpret_cast<const f32x4*>(bias + ct * 16 + 4 * q);
#pragma unroll
            for (int p = 0; p < PT; ++p)
#pragma unroll
                for (int ct = 0; ct < CTM; ++ct) acc[p][ct] = b4[ct];
            if constexpr (VCH > 0) {
                accA8[0] = accB8[0] = *reinterpret_cast<const f32x4*>(bias + 16 * CTM);
#pragma unroll
                for (int k = 1; k < 8; ++k) accA8[k] = accB8[k] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
        uint32_t baddr[PT];
        VAddr<VCH> va;
        va.remv = bias + G::CP + (lane & 3) * 8;
        const uint32_t stride = first ? (uint32_t)(kXW * 4) : (uint32_t)(U * 4);
        const uint32_t poff = (uint32_t)(reinterpret_cast<const char*>(first ? Xin : ACT) - smem);
#pragma unroll
        for (int p = 0; p < PT; ++p) baddr[p] = poff + (uint32_t)(tc.row[p] - 2) * stride + 8u * q;
        va.a = poff + (uint32_t)(rowA - 2) * stride;
        va.b = poff + (uint32_t)(tc.row[PT - 1] - 2) * stride;
        conv_accumulate<CTM, PT, VCH>(acc, accA8, accB8, pf.o, reinterpret_cast<const float2*>(wl) + lane, smem, baddr, va,
                                      first ? G::NCH_L0 : G::NCH_MID);
        if constexpr (VCH > 0) {   // fixed-order pairwise sum of the 8 partial accumulators
            accA = ((accA8[0] + accA8[1]) + (accA8[2] + accA8[3])) + ((accA8[4] + accA8[5]) + (accA8[6] + accA8[7]));
            accB = ((accB8[0] + accB8[1]) + (accB8[2] + accB8[3])) + ((accB8[4] + accB8[5]) + (accB8[6] + accB8[7]));
        }
        wl += fragf + G::CP + remf;
        // prefetch the next conv layer's chunk-0 weights (this stack's next layer, or the next stack's first)
        {
            const bool more = (l + 1 < n_layer);
            const float* wn = more ? wl : wnext;
            if (wn != nullptr) {
                const int nfrag = more ? G::MIDF : G::L0F;
                VAddr<VCH> vn;
                vn.a = vn.b = 0;
                vn.remv = wn + nfrag + G::CP + (lane & 3) * 8;
                load_w<CTM, PT, VCH>(pf.o, reinterpret_cast<const float2*>(wn) + lane, vn, 0);
            }
        }
        if (l + 1 < n_layer) {
            // in-place panel update: everyone must have finished reading the old activations
            if (!first) __syncthreads();
#pragma unroll
            for (int p = 0; p < PT; ++p) {
#pragma unroll
                for (int ct = 0; ct < CTM; ++ct) {
                    f32x4 v = acc[p][ct];
                    v.x = elu1(v.x); v.y = elu1(v.y); v.z = elu1(v.z); v.w = elu1(v.w);
                    if (tc.valid[p] && ct * 16 + 4 * q < U) *reinterpret_cast<f32x4*>(ACT + tc.row[p] * U + ct * 16 + 4 * q) = v;
                }
            }
            if constexpr (VCH > 0) {
                f32x4 va4 = {elu1(accA.x), elu1(accA.y), elu1(accA.z), elu1(accA.w)};
                f32x4 vb4 = {elu1(accB.x), elu1(accB.y), elu1(accB.z), elu1(accB.w)};
                if (validA) *reinterpret_cast<f32x4*>(ACT + rowA * U + 16 * CTM) = va4;
                if (tc.valid[PT - 1] && q == 0) *reinterpret_cast<f32x4*>(ACT + tc.row[PT - 1] * U + 16 * CTM) = vb4;
            }
            __syncthreads();
        }
    }
    // ---- Linear head fused on the accumulators of the last conv layer (decoders.py:233,243;
    //      encoders.py:364-371).  wl now points at lin_w [8][CP], then lin_b [8].
    float part[PT][8];
#pragma unroll
    for (int p = 0; p < PT; ++p)
#pragma unroll
        for (int f = 0; f < 8; ++f) part[p][f] = 0.0f;
#pragma unroll
    for (int ct = 0; ct < CTM; ++ct) {
        f32x4 w4[8];
#pragma unroll
        for (int f = 0; f < 8; ++f) w4[f] = *reinterpret_cast<const f32x4*>(wl + f * G::CP + ct * 16 + 4 * q);
#pragma unroll
        for (int p = 0; p < PT; ++p) {
            f32x4 v = acc[p][ct];
            v.x = elu1(v.x); v.y = elu1(v.y); v.z = elu1(v.z); v.w = elu1(v.w);
#pragma unroll
            for (int f = 0; f < 8; ++f) {
                float s = part[p][f];
                s = fmaf(v.x, w4[f].x, s); s = fmaf(v.y, w4[f].y, s);
                s = fmaf(v.z, w4[f].z, s); s = fmaf(v.w, w4[f].w, s);
                part[p][f] = s;
            }
        }
    }
    // remainder channels: one lane per position holds their contribution (group A: the lane of tile q,
    // group B: lane group 0 of tile PT-1); it is routed to the owning lane through a second butterfly
    // whose other inputs are exact zeros, so the result does not depend on which lane produced it.
    float ca[8], cb[8];
    if constexpr (VCH > 0) {
        const f32x4 ea = {elu1(accA.x), elu1(accA.y), elu1(accA.z), elu1(accA.w)};
        const f32x4 eb = {elu1(accB.x), elu1(accB.y), elu1(accB.z), elu1(accB.w)};
#pragma unroll
        for (int f = 0; f < 8; ++f) {
            const f32x4 w = *reinterpret_cast<const f32x4*>(wl + f * G::CP + 16 * CTM);
            ca[f] = fmaf(ea.w, w.w, fmaf(ea.z, w.z, fmaf(ea.y, w.y, ea.x * w.x)));
            cb[f] = fmaf(eb.w, w.w, fmaf(eb.z, w.z, fmaf(eb.y, w.y, eb.x * w.x)));
        }
    }
    const float* lb = wl + 8 * G::CP;
    const float bq0 = lb[2 * q], bq1 = lb[2 * q + 1];
    const bool hi32 = (q & 2) != 0, hi16 = (q & 1) != 0;
#pragma unroll
    for (int p = 0; p < PT; ++p) {
        float k2[2];
        butterfly8(part[p], hi32, hi16, k2);
        if constexpr (VCH > 0) {
            float rp[8], r2[2];
            const bool mine = (p < PT - 1) ? (q == p) : (q == 0);
#pragma unroll
            for (int f = 0; f < 8; ++f) rp[f] = mine ? ((p < PT - 1) ? ca[f] : cb[f]) : 0.0f;
            butterfly8(rp, hi32, hi16, r2);
            k2[0] += r2[0];
            k2[1] += r2[1];
        }
        if (tc.center[p]) {
            epi(p, 2 * q, k2[0] + bq0);
            epi(p, 2 * q + 1, k2[1] + bq1);
        }
    }
    __syncthreads();
}

// chunk-0 weights of a stack's first layer (used once per kernel, before the first stack)
template <int U, int PT>
__device__ __forceinline__ void prefetch_first(Prefetch<U, PT>& pf, const float* wstack, int lane) {
    using G = Geo<U>;
    VAddr<G::VCH> vn;
    vn.a = vn.b = 0;
    vn.remv = wstack + G::L0F + G::CP + (lane & 3) * 8;
    load_w<G::CTM, PT, G::VCH>(pf.o, reinterpret_cast<const float2*>(wstack) + lane, vn, 0);
}

template <int PT>
__device__ __forceinline__ void make_tiles(TileCtx<PT>& tc, int wave, int lane, int L, int npos) {
    const int n = lane & 15;
#pragma unroll
    for (int p = 0; p < PT; ++p) {
        const int m = (wave * PT + p) * 16 + n;
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

struct Panels {
    float* ACT;
    float* XA;
    float* XB;
    int* PERM;
    int* INV;
};

template <int U>
__device__ __forceinline__ Panels carve(char* smem, int rows, int L) {
    Panels pn;
    pn.ACT = reinterpret_cast<float*>(smem);
    pn.XA = pn.ACT + (size_t)(rows + 1) * U;
    pn.XB = pn.XA + (size_t)(rows + 1) * kXW;
    pn.PERM = reinterpret_cast<int*>(pn.XB + (size_t)(rows + 1) * kXW);
    pn.INV = pn.PERM + L;
    return pn;
}

__device__ __forceinline__ void zero_lds(char* smem, int bytes, int tid) {
    f32x4 z = {0.f, 0.f, 0.f, 0.f};
    for (int i = tid; i < bytes / 16; i += kThreads) reinterpret_cast<f32x4*>(smem)[i] = z;
}

// =============================================================================================
// Decoder: DEC_LargeCNN.forward (decoders.py:206-269) for nb blocks per workgroup.
template <int U, int PT>
__global__ __launch_bounds__(kThreads, 1) void dec_kernel(FusedParams P) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using G = Geo<U>;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int L = P.L, nb = P.nb;
    const int rows = nb * (L + 2) + 2;
    const Panels pn = carve<U>(smem, rows, L);
    const int blk0 = blockIdx.x * nb;
    const int nblk = min(nb, P.B - blk0);
    const int npos = nblk * L;

    zero_lds(smem, P.lds_bytes, tid);
    __syncthreads();
    for (int i = tid; i < L; i += kThreads) { pn.PERM[i] = P.perm[i]; pn.INV[i] = P.inv[i]; }
    __syncthreads();
    // r_sys, r_par1 -> XA ch 0,1 (natural order); r_sys_int, r_par2 -> XB ch 0,1 (decoders.py:221-224)
    const float* rx = P.in + (size_t)blk0 * L * 3;
    for (int m = tid; m < npos; m += kThreads) {
        const int b = m / L, t = m - b * L;
        const int row = b * (L + 2) + 2 + t;
        const float* r = rx + (size_t)m * 3;
        pn.XA[row * kXW + 0] = r[0];
        pn.XA[row * kXW + 1] = r[1];
        pn.XB[row * kXW + 0] = rx[((size_t)b * L + pn.PERM[t]) * 3 + 0];
        pn.XB[row * kXW + 1] = r[2];
    }
    __syncthreads();

    TileCtx<PT> tc;
    make_tiles<PT>(tc, wave, lane, L, npos);

    const int n_stack = 2 * P.n_iter;
    const int F = P.F;
    const bool extrinsic = P.extrinsic != 0;
    float* xdec = P.out + (size_t)blk0 * L;
    Prefetch<U, PT> pf;
    prefetch_first<U, PT>(pf, P.wpack, lane);
    for (int s = 0; s < n_stack; ++s) {
        const float* Xin = (s & 1) ? pn.XB : pn.XA;
        float* Xout = (s & 1) ? pn.XA : pn.XB;
        // dec1 output q[t] feeds dec2 at row inv[t] (interleave, decoders.py:238);
        // dec2 output q2[i] becomes prior[p[i]] (deinterleave, decoders.py:249)
        const int* ptab = (s & 1) ? pn.PERM : pn.INV;
        const float* wstack = P.wpack + (size_t)s * P.stack_stride;
        if (s + 1 < n_stack) {
            run_stack<U, PT>(wstack, wstack + P.stack_stride, P.n_layer, smem, pn.ACT, Xin, tc, lane, pf, [&](int p, int f, float v) {
                if (f < F) {
                    if (extrinsic) v -= Xin[tc.row[p] * kXW + 2 + f];   // decoders.py:235-236,246-247
                    Xout[(tc.rowbase[p] + ptab[tc.t[p]]) * kXW + 2 + f] = v;
                }
            });
        } else {
            // last half-iteration: Linear(U->1), no extrinsic subtraction, sigmoid(deinterleave) (decoders.py:262-267)
            run_stack<U, PT>(wstack, nullptr, P.n_layer, smem, pn.ACT, Xin, tc, lane, pf, [&](int p, int f, float v) {
                if (f == 0) xdec[tc.blk[p] * L + ptab[tc.t[p]]] = 1.0f / (1.0f + expf(-v));
            });
        }
    }
}

// =============================================================================================
// Encoder before power normalisation: ENC_interCNN.forward (encoders.py:362-373) + per-workgroup
// partial sums for power_constraint (encoders.py:107-108).
template <int U, int PT>
__global__ __launch_bounds__(kThreads, 1) void enc_kernel(FusedParams P) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int L = P.L, nb = P.nb;
    const int rows = nb * (L + 2) + 2;
    const Panels pn = carve<U>(smem, rows, L);
    const int blk0 = blockIdx.x * nb;
    const int nblk = min(nb, P.B - blk0);
    const int npos = nblk * L;

    zero_lds(smem, P.lds_bytes, tid);
    __syncthreads();
    for (int i = tid; i < L; i += kThreads) { pn.PERM[i] = P.perm[i]; pn.INV[i] = P.inv[i]; }
    __syncthreads();
    // inputs = 2u - 1 (encoders.py:362); XB holds the interleaved copy (encoders.py:369)
    const float* u = P.in + (size_t)blk0 * L;
    for (int m = tid; m < npos; m += kThreads) {
        const int b = m / L, t = m - b * L;
        const int row = b * (L + 2) + 2 + t;
        pn.XA[row * kXW] = 2.0f * u[m] - 1.0f;
        pn.XB[row * kXW] = 2.0f * u[b * L + pn.PERM[t]] - 1.0f;
    }
    __syncthreads();

    TileCtx<PT> tc;
    make_tiles<PT>(tc, wave, lane, L, npos);

    float* xtx = P.out + (size_t)blk0 * L * 3;
    const bool act_elu = P.act == 0;
    double sum = 0.0, sumsq = 0.0;
    Prefetch<U, PT> pf;
    prefetch_first<U, PT>(pf, P.wpack, lane);
    for (int s = 0; s < 3; ++s) {
        const float* Xin = (s == 2) ? pn.XB : pn.XA;
        const float* wstack = P.wpack + (size_t)s * P.stack_stride;
        run_stack<U, PT>(wstack, s < 2 ? wstack + P.stack_stride : nullptr, P.n_layer, smem, pn.ACT, Xin, tc, lane, pf,
                         [&](int p, int f, float v) {
            if (f == 0) {
                if (act_elu) v = elu1(v);                      // enc_act (encoders.py:364)
                xtx[(size_t)(tc.blk[p] * L + tc.t[p]) * 3 + s] = v;   // x_p2 stays in interleaved order (encoders.py:371-373)
                sum += (double)v;
                sumsq += (double)v * (double)v;
            }
        });
    }
    // block reduction of the fp64 partial sums (deterministic order); the panels are dead after the
    // last stack's closing barrier, so the reduction scratch aliases them (no static LDS: G17)
    double* red = reinterpret_cast<double*>(smem);
    red[tid] = sum;
    red[kThreads + tid] = sumsq;
    __syncthreads();
    for (int off = kThreads / 2; off > 0; off >>= 1) {
        if (tid < off) { red[tid] += red[tid + off]; red[kThreads + tid] += red[kThreads + tid + off]; }
        __syncthreads();
    }
    if (tid == 0) { P.partials[2 * blockIdx.x] = red[0]; P.partials[2 * blockIdx.x + 1] = red[kThreads]; }
}


// =============================================================================================
// Long blocks (block_len > 320, e.g. BASELINE configs[3] block_len=1000): a block's activations no
// longer fit one workgroup's LDS, so each SameShapeConv1d stack runs as its own launch over
// (block, segment) workgroups.  A segment owns T centre positions and loads H = 2*n_layer halo
// positions on each side (one conv layer widens the receptive field by 2); the panel is updated in
// place exactly as in the whole-block kernels, rows outside the block stay zero (Conv1d zero
// padding), and garbage from the panel edges creeps inwards 2 rows per layer, never reaching the
// centre.  The F extrinsic values per position travel between stacks through the (B, L, 8) fp32
// exchange buffers in HBM; (de)interleaving is the gather on the read side.
template <int U, int PT>
__global__ __launch_bounds__(kThreads, 1) void seg_kernel(SegParams P) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int L = P.L, H = 2 * P.n_layer, NP = P.T + 2 * H;
    const int rows = NP + 4;
    float* ACT = reinterpret_cast<float*>(smem);
    float* X = ACT + (size_t)(rows + 1) * U;
    int bid = blockIdx.x;
    int stack = P.stack;
    if (P.mode == 0) { stack = bid % 3; bid /= 3; }
    const int seg = bid % P.nseg, b = bid / P.nseg;
    const int s0 = seg * P.T;
    const int tlen = min(P.T, L - s0);
    const bool odd = (stack & 1) != 0;

    zero_lds(smem, P.lds_bytes, tid);
    __syncthreads();
    for (int m = tid; m < NP; m += kThreads) {
        const int t = s0 - H + m;
        if (t < 0 || t >= L) continue;
        float* xr = X + (size_t)(2 + m) * kXW;
        if (P.mode == 0) {
            const int src = (stack == 2) ? P.perm[t] : t;                           // encoders.py:369
            xr[0] = 2.0f * P.in[(size_t)b * L + src] - 1.0f;                         // encoders.py:362
        } else {
            const float* rx = P.in + (size_t)b * L * 3;
            xr[0] = odd ? rx[(size_t)P.perm[t] * 3] : rx[(size_t)t * 3];            // r_sys_int / r_sys
            xr[1] = rx[(size_t)t * 3 + (odd ? 2 : 1)];                              // r_par2 / r_par1
            if (stack > 0) {
                // dec2 reads q[p[i]] (interleave, decoders.py:238); dec1 reads q2[inv[j]] (deinterleave, :249)
                const int g = odd ? P.perm[t] : P.inv[t];
                const float* e = P.eprev + ((size_t)b * L + g) * 8;
                for (int f = 0; f < P.F; ++f) xr[2 + f] = e[f];
            }
        }
    }
    __syncthreads();

    TileCtx<PT> tc;
    {
        const int n = lane & 15;
#pragma unroll
        for (int p = 0; p < PT; ++p) {
            const int m = (wave * PT + p) * 16 + n;
            const int t = s0 - H + m;
            const bool v = (m < NP) && (t >= 0) && (t < L);
            tc.valid[p] = v;
            tc.center[p] = v && (t >= s0) && (t < s0 + tlen);
            tc.blk[p] = b;
            tc.t[p] = v ? t : 0;
            tc.rowbase[p] = 2;
            tc.row[p] = v ? 2 + m : 2;
        }
    }
    const float* wstack = P.wpack + (size_t)stack * P.stack_stride;
    Prefetch<U, PT> pf;
    prefetch_first<U, PT>(pf, wstack, lane);
    if (P.mode == 0) {
        const bool act_elu = P.act == 0;
        double sum = 0.0, sumsq = 0.0;
        float* xtx = P.out + (size_t)b * L * 3;
        run_stack<U, PT>(wstack, nullptr, P.n_layer, smem, ACT, X, tc, lane, pf, [&](int p, int f, float v) {
            if (f == 0) {
                if (act_elu) v = elu1(v);
                xtx[(size_t)tc.t[p] * 3 + stack] = v;
                sum += (double)v;
                sumsq += (double)v * (double)v;
            }
        });
        double* red = reinterpret_cast<double*>(smem);
        red[tid] = sum;
        red[kThreads + tid] = sumsq;
        __syncthreads();
        for (int off = kThreads / 2; off > 0; off >>= 1) {
            if (tid < off) { red[tid] += red[tid + off]; red[kThreads + tid] += red[kThreads + tid + off]; }
            __syncthreads();
        }
        if (tid == 0) { P.partials[2 * blockIdx.x] = red[0]; P.partials[2 * blockIdx.x + 1] = red[kThreads]; }
    } else if (!P.last) {
        const int F = P.F;
        const bool extrinsic = P.extrinsic != 0;
        float* ecur = P.ecur + (size_t)b * L * 8;
        run_stack<U, PT>(wstack, nullptr, P.n_layer, smem, ACT, X, tc, lane, pf, [&](int p, int f, float v) {
            if (f < F) {
                if (extrinsic) v -= X[tc.row[p] * kXW + 2 + f];
                ecur[(size_t)tc.t[p] * 8 + f] = v;
            }
        });
    } else {
        float* xdec = P.out + (size_t)b * L;
        run_stack<U, PT>(wstack, nullptr, P.n_layer, smem, ACT, X, tc, lane, pf, [&](int p, int f, float v) {
            if (f == 0) xdec[P.perm[tc.t[p]]] = 1.0f / (1.0f + expf(-v));    // sigmoid(deinterleave), decoders.py:267
        });
    }
}

// =============================================================================================
// stats[0..2] = (sum, sumsq, count) over this rank's shard, summed in fixed order.
__global__ void reduce_partials_kernel(const double* __restrict__ partials, int n, double count, double* __restrict__ stats) {
    __shared__ double red[2 * 256];
    double a = 0.0, b = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) { a += partials[2 * i]; b += partials[2 * i + 1]; }
    red[threadIdx.x] = a; red[256 + threadIdx.x] = b;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (threadIdx.x < off) { red[threadIdx.x] += red[threadIdx.x + off]; red[256 + threadIdx.x] += red[256 + threadIdx.x + off]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { stats[0] = red[0]; stats[1] = red[256]; stats[2] = count; }
}

// power_constraint (encoders.py:107-116): codes = (x - mean) * 1.0 / std, unbiased std, fp32 arithmetic
// on fp32 mean/std; channel_ae.py:42: received = codes + noise.
__global__ void normalize_kernel(const float* __restrict__ xtx, const double* __restrict__ stats,
                                 const float* __restrict__ noise, float* __restrict__ codes, float* __restrict__ rx,
                                 size_t n) {
    const double sum = stats[0], sumsq = stats[1], cnt = stats[2];
    const double mean_d = sum / cnt;
    double var_d = (sumsq - sum * mean_d) / (cnt - 1.0);
    if (var_d < 0.0) var_d = 0.0;
    const float mean = (float)mean_d;
    const float sd = (float)sqrt(var_d);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float c = __fdiv_rn(xtx[i] - mean, sd);
        if (codes) codes[i] = c;
        if (rx) rx[i] = c + noise[i];
    }
}

// errors_ber / errors_bler (utils.py:6-18,49-66) as integer counts: bit errors and blocks with >=1 error.
// torch.round is half-to-even, so round(sigmoid) == 1 iff sigmoid > 0.5.
__global__ void count_errors_kernel(const float* __restrict__ xdec, const float* __restrict__ u, int B, int L,
                                    unsigned long long* __restrict__ counts) {
    const int lane = threadIdx.x & 63;
    const int wave_global = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    unsigned long long bit_err = 0, blk_err = 0;
    for (int b = wave_global; b < B; b += nwaves) {
        int e = 0;
        for (int t = lane; t < L; t += 64) {
            const float xh = xdec[(size_t)b * L + t] > 0.5f ? 1.0f : 0.0f;
            const float xt = u[(size_t)b * L + t] > 0.5f ? 1.0f : 0.0f;
            e += (xh != xt);
        }
        for (int off = 32; off > 0; off >>= 1) e += __shfl_xor(e, off);
        bit_err += e;
        blk_err += (e > 0);
    }
    if (lane == 0 && (bit_err | blk_err)) {
        atomicAdd(&counts[0], bit_err);
        atomicAdd(&counts[1], blk_err);
    }
}

// Test inputs on device (replaces trainer.py:167-169): u ~ Bernoulli(0.5), noise = sigma * N(0,1),
// element e of block-major tensors keyed by the GLOBAL element index so any shard matches the
// single-device stream.  Box-Muller in fp64 (see turboae_amd/philox.py).
__global__ void gen_inputs_kernel(float* __restrict__ u, float* __restrict__ noise, size_t n_bits, size_t bit_offset,
                                  unsigned long long seed_bits, unsigned long long seed_noise, float sigma) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    // bits: 4 per philox call
    const size_t first_call = bit_offset >> 2, last_call = (bit_offset + n_bits - 1) >> 2;
    for (size_t c = first_call + tid; u != nullptr && c <= last_call; c += stride) {
        const u32x4 w = philox_call(seed_bits, STREAM_BITS, c);
        const uint32_t ww[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const size_t e = (c << 2) + k;
            if (e >= bit_offset && e < bit_offset + n_bits) u[e - bit_offset] = (float)(ww[k] & 1u);
        }
    }
    // normals: 3 per bit; one philox call -> 2 (u1,u2) pairs -> 4 normals
    const size_t n0 = 3 * bit_offset, nn = 3 * n_bits;
    const size_t fc = n0 >> 2, lc = (n0 + nn - 1) >> 2;
    for (size_t c = fc + tid; noise != nullptr && c <= lc; c += stride) {
        const u32x4 w = philox_call(seed_noise, STREAM_NOISE, c);
        const double r0 = sqrt(-2.0 * log(u32_to_unit_open(w.x))), th0 = 6.283185307179586476925 * u32_to_unit_open(w.y);
        const double r1 = sqrt(-2.0 * log(u32_to_unit_open(w.z))), th1 = 6.283185307179586476925 * u32_to_unit_open(w.w);
        const float z[4] = {(float)(r0 * cos(th0)), (float)(r0 * sin(th0)), (float)(r1 * cos(th1)), (float)(r1 * sin(th1))};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const size_t e = (c << 2) + k;
            if (e >= n0 && e < n0 + nn) noise[e - n0] = sigma * z[k];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// host-side launch helpers (called from turboae_api.cpp through turboae_internal.hpp)
template <int U>
static hipError_t launch_fused_u(bool decoder, const FusedParams& P, int grid, hipStream_t st) {
    constexpr int PT = 5;
    auto kd = dec_kernel<U, PT>;
    auto ke = enc_kernel<U, PT>;
    const void* fn = decoder ? reinterpret_cast<const void*>(kd) : reinterpret_cast<const void*>(ke);
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, P.lds_bytes);
    if (e != hipSuccess) return e;
    if (decoder) hipLaunchKernelGGL(kd, dim3(grid), dim3(kThreads), P.lds_bytes, st, P);
    else hipLaunchKernelGGL(ke, dim3(grid), dim3(kThreads), P.lds_bytes, st, P);
    return hipGetLastError();
}

hipError_t launch_fused(int U, bool decoder, const FusedParams& P, int grid, hipStream_t st) {
    switch (U) {
        case 100: return launch_fused_u<100>(decoder, P, grid, st);
        case 64: return launch_fused_u<64>(decoder, P, grid, st);
        case 32: return launch_fused_u<32>(decoder, P, grid, st);
        default: return hipErrorInvalidValue;
    }
}


template <int U>
static hipError_t launch_seg_u(const SegParams& P, int grid, hipStream_t st) {
    constexpr int PT = 5;
    auto k = seg_kernel<U, PT>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, P.lds_bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k, dim3(grid), dim3(kThreads), P.lds_bytes, st, P);
    return hipGetLastError();
}

hipError_t launch_seg(int U, const SegParams& P, int grid, hipStream_t st) {
    switch (U) {
        case 100: return launch_seg_u<100>(P, grid, st);
        case 64: return launch_seg_u<64>(P, grid, st);
        case 32: return launch_seg_u<32>(P, grid, st);
        default: return hipErrorInvalidValue;
    }
}

int seg_lds_bytes(int U, int T, int n_layer) {
    const int rows = T + 4 * n_layer + 4;
    size_t b = (size_t)(rows + 1) * U * 4 + (size_t)(rows + 1) * kXW * 4;
    b = (b + 15) & ~(size_t)15;
    if (b < 2 * kThreads * sizeof(double)) b = 2 * kThreads * sizeof(double);
    return (int)b;
}

hipError_t launch_reduce_partials(const double* partials, int n, double count, double* stats, hipStream_t st) {
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(1), dim3(256), 0, st, partials, n, count, stats);
    return hipGetLastError();
}

hipError_t launch_normalize(const float* xtx, const double* stats, const float* noise, float* codes, float* rx, size_t n,
                            hipStream_t st) {
    const int grid = (int)std::min<size_t>((n + 255) / 256, 2048);
    hipLaunchKernelGGL(normalize_kernel, dim3(grid), dim3(256), 0, st, xtx, stats, noise, codes, rx, n);
    return hipGetLastError();
}

hipError_t launch_count_errors(const float* xdec, const float* u, int B, int L, unsigned long long* counts, hipStream_t st) {
    const int grid = std::min((B + 3) / 4, 2048);
    hipLaunchKernelGGL(count_errors_kernel, dim3(grid), dim3(256), 0, st, xdec, u, B, L, counts);
    return hipGetLastError();
}

hipError_t launch_gen_inputs(float* u, float* noise, size_t n_bits, size_t bit_offset, unsigned long long seed_bits,
                             unsigned long long seed_noise, float sigma, hipStream_t st) {
    hipLaunchKernelGGL(gen_inputs_kernel, dim3(1024), dim3(256), 0, st, u, noise, n_bits, bit_offset, seed_bits, seed_noise, sigma);
    return hipGetLastError();
}

// LDS bytes and packed-weight geometry shared with the host packer
int fused_lds_bytes(int U, int L, int nb) {
    const int rows = nb * (L + 2) + 2;
    size_t b = (size_t)(rows + 1) * U * 4 + 2 * (size_t)(rows + 1) * kXW * 4 + 2 * (size_t)L * 4;
    b = (b + 15) & ~(size_t)15;
    return (int)b;
}

int fused_max_positions() { return kWaves * 5 * 16; }

}  // namespace tae
