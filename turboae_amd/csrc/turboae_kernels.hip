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
#include "../../include/turboae_hip.h"      // TAE_NOISE_* kinds
#include "turboae_internal.hpp"
#include "turboae_device.hpp"
#include "philox.hpp"

namespace tae {

// Weight-side state shared by the stacks of one kernel: the buffer resource over the packed
// weights and the chunk-0 A fragments of the NEXT conv layer, fetched before the current layer's
// epilogue so that their L2 latency hides behind it.
template <int U, int PT, int C0, int NC>
struct WeightStream {
    __amdgpu_buffer_rsrc_t rsrc;
    uint32_t voff;                       // lane * 16
    Ops<NC, PT> o;
    __device__ __forceinline__ void init(const float* wpack, uint32_t bytes, int lane) {
        rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wpack), 0, (int)bytes, 0x00020000);
        voff = (uint32_t)lane * 16u;
    }
    __device__ __forceinline__ void prefetch(uint32_t soff) { load_w<Geo<U>::CT, C0, NC, PT>(o, rsrc, voff, soff); }
};

// Runs one SameShapeConv1d stack (cnn_utils.py:36-46) followed by its Linear head for the
// workgroup's blocks; this wave owns channel tiles [C0, C0 + NC) of position group g.
// `epi(p, f, value)` is called (by the C0 == 0 wave) for output feature f (0..7) of this lane's
// position in tile p by the lane that owns (p, f): lane group q owns f = 2q and f = 2q + 1.
// `soff` = byte offset of this stack inside the packed weights; `snext` = byte offset of the stack
// that runs next in this kernel (or 0xffffffff): its first chunk is prefetched before the head.
// `HS` = head-combine scratch [kHeadSlots][8] floats: the upper channel half parks its partial
// Linear outputs there and the lower half adds them in a fixed order.
//
// Packed stack layout (floats), written by turboae_api_create.hip::pack_stack:
//   per layer: A fragments [chunk][...] (see load_w) | bias [CP];  then Linear weights [8][CP] | bias [8]
// SUPER: this (upper-half) wave additionally computes the 4 remainder channels of its position group with
// two super-tiles (see super_accumulate); its NC then excludes the padded last channel tile.
template <int U, int PT, int C0, int NC, bool SUPER, class Epi>
__device__ __forceinline__ void run_stack(const float* __restrict__ wpack, uint32_t soff, uint32_t snext, int n_layer,
                                          char* smem, float* ACT, const float* Xin, float* HS, const TileCtx<PT>& tc,
                                          const SuperCtx& sc, int g, int lane, WeightStream<U, PT, C0, NC>& ws, Epi epi) {
    using G = Geo<U>;
    constexpr int CTT = G::CT;
    constexpr int CREM = 16 * (CTT - 1);      // first remainder channel (96 for U = 100)
    static_assert(!SUPER || (G::SUP && C0 + NC <= CTT - 1), "with super-tiles the padded last channel tile is not computed");
    const int q = lane >> 4;
    f32x4 acc[PT][NC];
    f32x4 accS[2];
    uint32_t lo = soff;      // byte offset of the current layer
    for (int l = 0; l < n_layer; ++l) {
        const bool first = (l == 0);
        const uint32_t fragb = (first ? G::L0F : G::MIDF) * 4u;
        const float* bias = wpack + (lo + fragb) / 4;
        // accumulators start at the bias (Conv1d bias=True, cnn_utils.py:15-17)
        {
            f32x4 b4[NC];
#pragma unroll
            for (int i = 0; i < NC; ++i) b4[i] = *reinterpret_cast<const f32x4*>(bias + (C0 + i) * 16 + 4 * q);
#pragma unroll
            for (int p = 0; p < PT; ++p)
#pragma unroll
                for (int i = 0; i < NC; ++i) acc[p][i] = b4[i];
            if constexpr (SUPER) {
                accS[0] = *reinterpret_cast<const f32x4*>(bias + CREM);   // rows (s, c): bias[96 + c]
                accS[1] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
        uint32_t baddr[PT];
        const uint32_t stride = first ? (uint32_t)(kXW * 4) : (uint32_t)(U * 4);
        const uint32_t poff = (uint32_t)(reinterpret_cast<const char*>(first ? Xin : ACT) - smem);
#pragma unroll
        for (int p = 0; p < PT; ++p) baddr[p] = poff + (uint32_t)(tc.row[p] - 2) * stride + 8u * q;
        uint32_t sb = 0u;
        if constexpr (SUPER) sb = poff + (uint32_t)(sc.row0 - 2) * stride + 8u * q;
        const uint32_t so = lo + fragb + G::CP * 4u;     // super A fragments follow the bias
        if (first) conv_accumulate<CTT, C0, NC, PT, G::NCH_L0, SUPER ? G::SCH_L0 / 2 : 0>(acc, accS, ws.o, ws.rsrc, ws.voff, lo, so, smem, baddr, sb);
        else conv_accumulate<CTT, C0, NC, PT, G::NCH_MID, SUPER ? G::SCH_MID / 2 : 0>(acc, accS, ws.o, ws.rsrc, ws.voff, lo, so, smem, baddr, sb);
        lo += fragb + G::CP * 4u + (first ? G::SF0 : G::SFM) * 4u;
        // prefetch the next conv layer's chunk-0 weights (this stack's next layer, or the next stack's first)
        {
            const uint32_t nxt = (l + 1 < n_layer) ? lo : snext;
            if (nxt != 0xffffffffu) ws.prefetch(nxt);
        }
        if (l + 1 < n_layer) {
            // in-place panel update: everyone must have finished reading the old activations
            if (!first && !(TAE_X & 1)) __syncthreads();
#pragma unroll
            for (int p = 0; p < PT; ++p) {
#pragma unroll
                for (int i = 0; i < NC; ++i) {
                    f32x4 v = acc[p][i];
                    if (!(TAE_X & 4)) { v.x = elu1(v.x); v.y = elu1(v.y); v.z = elu1(v.z); v.w = elu1(v.w); }
                    const int ch = (C0 + i) * 16 + 4 * q;
                    if (TAE_X & 2) asm volatile("" :: "v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w));
                    else if (tc.valid[p] && ch < U) *reinterpret_cast<f32x4*>(ACT + tc.row[p] * U + ch) = v;
                }
            }
            if constexpr (SUPER) {
                f32x4 v = accS[0] + accS[1];
                v.x = elu1(v.x); v.y = elu1(v.y); v.z = elu1(v.z); v.w = elu1(v.w);
                if (sc.valid) *reinterpret_cast<f32x4*>(ACT + (sc.row0 + q) * U + CREM) = v;
            }
            if (!(TAE_X & 1)) __syncthreads();
        }
    }
    // ---- Linear head fused on the accumulators of the last conv layer (decoders.py:233,243;
    //      encoders.py:364-371): lin_w [8][CP], then lin_b [8].
    const float* wl = wpack + lo / 4;
    float part[PT][8];
#pragma unroll
    for (int p = 0; p < PT; ++p)
#pragma unroll
        for (int f = 0; f < 8; ++f) part[p][f] = 0.0f;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        f32x4 w4[8];
#pragma unroll
        for (int f = 0; f < 8; ++f) w4[f] = *reinterpret_cast<const f32x4*>(wl + f * G::CP + (C0 + i) * 16 + 4 * q);
#pragma unroll
        for (int p = 0; p < PT; ++p) {
            f32x4 v = acc[p][i];
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
    const bool hi32 = (q & 2) != 0, hi16 = (q & 1) != 0;
    float k2[PT][2];
#pragma unroll
    for (int p = 0; p < PT; ++p) butterfly8(part[p], hi32, hi16, k2[p]);
    // combine the two channel halves: upper half parks its partials, lower half adds them
    const int n = lane & 15;
    // remainder channels (super-tile): this lane holds all 4 of them for its quad position; their Linear
    // contribution is added to that position's parked partial sums (one wave: LDS operations are in order).
    // The sum parked for a position is always  k2(upper half) + c_super, whichever wave computed c_super.
    auto add_super = [&]() {
        f32x4 e = accS[0] + accS[1];
        e.x = elu1(e.x); e.y = elu1(e.y); e.z = elu1(e.z); e.w = elu1(e.w);
        float* hs = HS + sc.slot * 8;
        f32x4 h0 = *reinterpret_cast<const f32x4*>(hs), h1 = *reinterpret_cast<const f32x4*>(hs + 4);
        float c8[8];
#pragma unroll
        for (int f = 0; f < 8; ++f) {
            const f32x4 w = *reinterpret_cast<const f32x4*>(wl + f * G::CP + CREM);
            c8[f] = fmaf(e.w, w.w, fmaf(e.z, w.z, fmaf(e.y, w.y, e.x * w.x)));
        }
        h0 += f32x4{c8[0], c8[1], c8[2], c8[3]};
        h1 += f32x4{c8[4], c8[5], c8[6], c8[7]};
        if (sc.center) {
            *reinterpret_cast<f32x4*>(hs) = h0;
            *reinterpret_cast<f32x4*>(hs + 4) = h1;
        }
    };
    if constexpr (C0 != 0) {
#pragma unroll
        for (int p = 0; p < PT; ++p)
            *reinterpret_cast<float2*>(HS + ((g * PT + p) * 16 + n) * 8 + 2 * q) = float2{k2[p][0], k2[p][1]};
        if constexpr (SUPER) add_super();
    }
    __syncthreads();
    if constexpr (C0 == 0 && SUPER) add_super();       // the lower half's super-tile (quads 0..15), after the upper half parked
    if constexpr (C0 == 0) {
        const float* lb = wl + 8 * G::CP;
        const float bq0 = lb[2 * q], bq1 = lb[2 * q + 1];
#pragma unroll
        for (int p = 0; p < PT; ++p) {
            float2 other = float2{0.0f, 0.0f};
            if constexpr (NC < CTT) other = *reinterpret_cast<const float2*>(HS + ((g * PT + p) * 16 + n) * 8 + 2 * q);
            if (tc.center[p]) {
                epi(p, 2 * q, (k2[p][0] + other.x) + bq0);
                epi(p, 2 * q + 1, (k2[p][1] + other.y) + bq1);
            }
        }
    }
    __syncthreads();
}

// super-tile view of position group g (whole-block layout): the lower channel half (h = 0) takes quads 0..15,
// the upper half (h = 1) quads 16..19 of the group's 20 (its lanes n >= 4 duplicate them and never write back)
template <int PT>
__device__ __forceinline__ void make_super(SuperCtx& sc, int g, int h, int lane, int L, int npos) {
    const int n = lane & 15, q = lane >> 4;
    const int Q = h == 0 ? n : 16 + (n & 3);
    const int m0 = g * PT * 16 + 4 * Q;
    const bool ok = (h == 0 || n < 4) && (m0 + q < npos);
    const int mm0 = m0 < npos ? m0 : 0;
    const int b = mm0 / L, t0 = mm0 - b * L;
    sc.row0 = b * (L + 2) + 2 + t0;
    sc.slot = mm0 + q;
    sc.valid = ok;
    sc.center = ok;
}

struct Panels {
    float* ACT;
    float* XA;
    float* XB;
    int* PERM;
    int* INV;
    float* HS;       // head-combine scratch [kHeadSlots][8]
};

template <int U>
__device__ __forceinline__ Panels carve(char* smem, int rows, int L) {
    Panels pn;
    pn.ACT = reinterpret_cast<float*>(smem);
    pn.XA = pn.ACT + (size_t)(rows + 1) * U;
    pn.XB = pn.XA + (size_t)(rows + 1) * kXW;
    pn.PERM = reinterpret_cast<int*>(pn.XB + (size_t)(rows + 1) * kXW);
    pn.INV = pn.PERM + L;
    pn.HS = reinterpret_cast<float*>(smem + (((reinterpret_cast<char*>(pn.INV + L) - smem) + 15) & ~15));
    return pn;
}

// =============================================================================================
// Decoder: DEC_LargeCNN.forward (decoders.py:206-269) for nb blocks per workgroup.
template <int U, int PT, int C0, int NC, bool SUPER, bool TAPS>
__device__ __forceinline__ void dec_body(const FusedParams& P, char* smem, const Panels& pn, const TileCtx<PT>& tc,
                                         const SuperCtx& sc, int g, int lane, int blk0) {
    const int L = P.L;
    const int n_stack = 2 * P.n_iter;
    const int F = P.F;
    const bool extrinsic = P.extrinsic != 0;
    float* xdec = P.out + (size_t)blk0 * L;
    WeightStream<U, PT, C0, NC> ws;
    ws.init(P.wpack, P.wpack_bytes, lane);
    ws.prefetch(0);
    const uint32_t sstride = P.stack_stride * 4u;
    for (int s = 0; s < n_stack; ++s) {
        const float* Xin = (s & 1) ? pn.XB : pn.XA;
        float* Xout = (s & 1) ? pn.XA : pn.XB;
        // dec1 output q[t] feeds dec2 at row inv[t] (interleave, decoders.py:238);
        // dec2 output q2[i] becomes prior[p[i]] (deinterleave, decoders.py:249)
        const int* ptab = (s & 1) ? pn.PERM : pn.INV;
        if (s + 1 < n_stack) {
            run_stack<U, PT, C0, NC, SUPER>(P.wpack, s * sstride, (s + 1) * sstride, P.n_layer, smem, pn.ACT, Xin, pn.HS, tc, sc, g, lane, ws,
                                     [&](int p, int f, float v) {
                if (f < F) {
                    if (extrinsic) v -= Xin[tc.row[p] * kXW + 2 + f];   // decoders.py:235-236,246-247
                    if constexpr (TAPS) P.tap_out[(((size_t)s * P.B + blk0 + tc.blk[p]) * L + tc.t[p]) * F + f] = v;
                    Xout[(tc.rowbase[p] + ptab[tc.t[p]]) * kXW + 2 + f] = v;
                }
            });
        } else {
            // last half-iteration: Linear(U->1), no extrinsic subtraction, sigmoid(deinterleave) (decoders.py:262-267)
            run_stack<U, PT, C0, NC, SUPER>(P.wpack, s * sstride, 0xffffffffu, P.n_layer, smem, pn.ACT, Xin, pn.HS, tc, sc, g, lane, ws,
                                     [&](int p, int f, float v) {
                if (f == 0) xdec[tc.blk[p] * L + ptab[tc.t[p]]] = 1.0f / (1.0f + expf(-v));
            });
        }
    }
}

template <int U, int PT, bool TAPS = false>
__global__ __launch_bounds__(kThreads, 2) void dec_kernel(FusedParams P) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = wave & (kGroups - 1), h = wave / kGroups;
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
    make_tiles<PT>(tc, g, lane, L, npos);
    SuperCtx sc;
    make_super<PT>(sc, g, h, lane, L, npos);
    const bool upper = __builtin_amdgcn_readfirstlane(h) != 0;
    if constexpr (Geo<U>::SUP) {
        if (P.super) {
            if (!upper) dec_body<U, PT, 0, Split<U>::SA, true, TAPS>(P, smem, pn, tc, sc, g, lane, blk0);
            else dec_body<U, PT, Split<U>::SA, Split<U>::SB, true, TAPS>(P, smem, pn, tc, sc, g, lane, blk0);
            return;
        }
    }
    if (!upper) dec_body<U, PT, 0, Split<U>::CTA, false, TAPS>(P, smem, pn, tc, sc, g, lane, blk0);
    else dec_body<U, PT, Split<U>::CTA, Split<U>::CTB, false, TAPS>(P, smem, pn, tc, sc, g, lane, blk0);
}

// =============================================================================================
// Encoder before power normalisation: ENC_interCNN.forward (encoders.py:362-373) + per-workgroup
// partial sums for power_constraint (encoders.py:107-108).
template <int U, int PT, int C0, int NC, bool SUPER>
__device__ __forceinline__ void enc_body(const FusedParams& P, char* smem, const Panels& pn, const TileCtx<PT>& tc,
                                         const SuperCtx& sc, int g, int lane, int blk0, double& sum, double& sumsq) {
    const int L = P.L;
    float* xtx = P.out + (size_t)blk0 * L * 3;
    const int act = P.act;
    WeightStream<U, PT, C0, NC> ws;
    ws.init(P.wpack, P.wpack_bytes, lane);
    ws.prefetch(0);
    const uint32_t sstride = P.stack_stride * 4u;
    for (int s = 0; s < 3; ++s) {
        const float* Xin = (s == 2) ? pn.XB : pn.XA;
        run_stack<U, PT, C0, NC, SUPER>(P.wpack, s * sstride, s < 2 ? (s + 1) * sstride : 0xffffffffu, P.n_layer, smem, pn.ACT,
                                        Xin, pn.HS, tc, sc, g, lane, ws, [&](int p, int f, float v) {
            if (f == 0) {
                v = act_apply(v, act);                         // enc_act (encoders.py:364)
                xtx[(size_t)(tc.blk[p] * L + tc.t[p]) * 3 + s] = v;   // x_p2 stays in interleaved order (encoders.py:371-373)
                sum += (double)v;
                sumsq += (double)v * (double)v;
            }
        });
    }
}

template <int U, int PT>
__global__ __launch_bounds__(kThreads, 2) void enc_kernel(FusedParams P) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = wave & (kGroups - 1), h = wave / kGroups;
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
    make_tiles<PT>(tc, g, lane, L, npos);
    double sum = 0.0, sumsq = 0.0;
    SuperCtx sc;
    make_super<PT>(sc, g, h, lane, L, npos);
    const bool upper = __builtin_amdgcn_readfirstlane(h) != 0;
    bool done = false;
    if constexpr (Geo<U>::SUP) {
        if (P.super) {
            if (!upper) enc_body<U, PT, 0, Split<U>::SA, true>(P, smem, pn, tc, sc, g, lane, blk0, sum, sumsq);
            else enc_body<U, PT, Split<U>::SA, Split<U>::SB, true>(P, smem, pn, tc, sc, g, lane, blk0, sum, sumsq);
            done = true;
        }
    }
    if (!done) {
        if (!upper) enc_body<U, PT, 0, Split<U>::CTA, false>(P, smem, pn, tc, sc, g, lane, blk0, sum, sumsq);
        else enc_body<U, PT, Split<U>::CTA, Split<U>::CTB, false>(P, smem, pn, tc, sc, g, lane, blk0, sum, sumsq);
    }
    block_reduce_stats(smem, tid, sum, sumsq, P.partials);
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
template <int U, int PT, int C0, int NC, bool SUPER>
__device__ __forceinline__ void seg_body(const SegParams& P, char* smem, float* ACT, float* X, float* HS, const TileCtx<PT>& tc,
                                         const SuperCtx& sc, int g, int lane, int stack, int b, double& sum, double& sumsq) {
    const int L = P.L;
    WeightStream<U, PT, C0, NC> ws;
    ws.init(P.wpack, P.wpack_bytes, lane);
    const uint32_t soff = (uint32_t)stack * P.stack_stride * 4u;
    ws.prefetch(soff);
    if (P.mode == 0) {
        const int act = P.act;
        float* xtx = P.out + (size_t)b * L * 3;
        run_stack<U, PT, C0, NC, SUPER>(P.wpack, soff, 0xffffffffu, P.n_layer, smem, ACT, X, HS, tc, sc, g, lane, ws, [&](int p, int f, float v) {
            if (f == 0) {
                v = act_apply(v, act);
                xtx[(size_t)tc.t[p] * 3 + stack] = v;
                sum += (double)v;
                sumsq += (double)v * (double)v;
            }
        });
    } else if (!P.last) {
        const int F = P.F;
        const bool extrinsic = P.extrinsic != 0;
        float* ecur = P.ecur + (size_t)b * L * 8;
        run_stack<U, PT, C0, NC, SUPER>(P.wpack, soff, 0xffffffffu, P.n_layer, smem, ACT, X, HS, tc, sc, g, lane, ws, [&](int p, int f, float v) {
            if (f < F) {
                if (extrinsic) v -= X[tc.row[p] * kXW + 2 + f];
                ecur[(size_t)tc.t[p] * 8 + f] = v;
            }
        });
    } else {
        float* xdec = P.out + (size_t)b * L;
        run_stack<U, PT, C0, NC, SUPER>(P.wpack, soff, 0xffffffffu, P.n_layer, smem, ACT, X, HS, tc, sc, g, lane, ws, [&](int p, int f, float v) {
            if (f == 0) xdec[P.perm[tc.t[p]]] = 1.0f / (1.0f + expf(-v));    // sigmoid(deinterleave), decoders.py:267
        });
    }
}

template <int U, int PT>
__global__ __launch_bounds__(kThreads, 2) void seg_kernel(SegParams P) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = wave & (kGroups - 1), h = wave / kGroups;
    const int L = P.L, H = 2 * P.n_layer;
    int bid = blockIdx.x;
    int stack = P.stack;
    if (P.mode == 0) { stack = bid % 3; bid /= 3; }
    const int seg = bid % P.nseg, b = bid / P.nseg;
    // segment 0 owns T0 centre positions, the others T (the last one what is left); only positions inside the block are
    // walked: no halo in front of position 0 or behind position L - 1 (those rows are the Conv1d zero padding)
    const int s0 = seg == 0 ? 0 : P.T0 + (seg - 1) * P.T;
    const int tlen = min(seg == 0 ? P.T0 : P.T, L - s0);
    // panel position m <-> block index t = tstart + m; tstart is floored to a multiple of 4 so that the
    // super-tile's shift index (m mod 4) equals t mod 4, exactly as in the whole-block kernels
    const int tstart = max(s0 - H, 0) & ~3;
    const int NP = min(s0 + tlen + H, L) - tstart;     // <= T + 2H + 3
    const int rows = P.T + 2 * H + 3 + 4;              // allocation-independent of the segment
    float* ACT = reinterpret_cast<float*>(smem);
    float* X = ACT + (size_t)(rows + 1) * U;
    float* HS = X + (size_t)(rows + 1) * kXW;
    const bool odd = (stack & 1) != 0;

    zero_lds(smem, P.lds_bytes, tid);
    __syncthreads();
    for (int m = tid; m < NP; m += kThreads) {
        const int t = tstart + m;
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
                const int gi = odd ? P.perm[t] : P.inv[t];
                const float* e = P.eprev + ((size_t)b * L + gi) * 8;
                for (int f = 0; f < P.F; ++f) xr[2 + f] = e[f];
            }
        }
    }
    __syncthreads();

    TileCtx<PT> tc;
    SuperCtx sc;
    {
        const int n = lane & 15, q = lane >> 4;
#pragma unroll
        for (int p = 0; p < PT; ++p) {
            const int m = (g * PT + p) * 16 + n;
            const int t = tstart + m;
            const bool v = (m < NP) && (t >= 0) && (t < L);
            tc.valid[p] = v;
            tc.center[p] = v && (t >= s0) && (t < s0 + tlen);
            tc.blk[p] = b;
            tc.t[p] = v ? t : 0;
            tc.rowbase[p] = 2;
            tc.row[p] = v ? 2 + m : 2;
        }
        {
            const int Q = h == 0 ? n : 16 + (n & 3);
            const int m0 = g * PT * 16 + 4 * Q;
            const int t = tstart + m0 + q;
            const bool ok = (h == 0 || n < 4) && (m0 + q < NP) && (t >= 0) && (t < L);
            const int mm0 = m0 < NP ? m0 : 0;
            sc.row0 = 2 + mm0;
            sc.slot = mm0 + q;
            sc.valid = ok;
            sc.center = ok && (t >= s0) && (t < s0 + tlen);
        }
    }
    double sum = 0.0, sumsq = 0.0;
    const bool upper = __builtin_amdgcn_readfirstlane(h) != 0;
    bool done = false;
    if constexpr (Geo<U>::SUP) {
        if (P.super) {
            if (!upper) seg_body<U, PT, 0, Split<U>::SA, true>(P, smem, ACT, X, HS, tc, sc, g, lane, stack, b, sum, sumsq);
            else seg_body<U, PT, Split<U>::SA, Split<U>::SB, true>(P, smem, ACT, X, HS, tc, sc, g, lane, stack, b, sum, sumsq);
            done = true;
        }
    }
    if (!done) {
        if (!upper) seg_body<U, PT, 0, Split<U>::CTA, false>(P, smem, ACT, X, HS, tc, sc, g, lane, stack, b, sum, sumsq);
        else seg_body<U, PT, Split<U>::CTA, Split<U>::CTB, false>(P, smem, ACT, X, HS, tc, sc, g, lane, stack, b, sum, sumsq);
    }
    if (P.mode == 0) block_reduce_stats(smem, tid, sum, sumsq, P.partials);
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

// STEQuantize.forward (encoders.py:20-36 / ste.py:9-23): clamp to +-limit, then sign (level 2) or a uniform
// `level`-point grid; torch.round is round-half-to-even (rintf), the fp32 operation order is the reference's.
__device__ __forceinline__ float ste_quantize(float x, float lim, float level) {
    const float range = 2.0f * lim;
    const float c = fminf(fmaxf(x, -lim), lim);
    if (level == 2.0f) return c > 0.0f ? 1.0f : (c < 0.0f ? -1.0f : 0.0f);
    const float k = (float)(((double)level - 1.0) / (double)range);
    return __fdiv_rn(rintf((c + lim) * k) * range, level - 1.0f) - lim;
}

// power_constraint (encoders.py:102-125): codes = (x - mean) * 1.0 / std with the batch's mean / unbiased std
// (fp32 arithmetic on fp32 mean/std), or no normalisation, or fixed statistics; optional STE quantisation and
// truncation; then the channel of Channel_AE.forward (channel_ae.py:41-49) and the optional receive quantiser (:67-69).
__global__ void normalize_kernel(const float* __restrict__ xtx, const double* __restrict__ stats,
                                 const float* __restrict__ noise, float* __restrict__ codes, float* __restrict__ rx,
                                 size_t n, NormOpts o) {
    float mean = o.mean, sd = o.std;
    if (o.norm_mode == 0) {
        const double sum = stats[0], sumsq = stats[1], cnt = stats[2];
        const double mean_d = sum / cnt;
        double var_d = (sumsq - sum * mean_d) / (cnt - 1.0);
        if (var_d < 0.0) var_d = 0.0;
        mean = (float)mean_d;
        sd = (float)sqrt(var_d);
    }
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float c = xtx[i];
        if (o.norm_mode != 1) {
            c = __fdiv_rn(c - mean, sd);
            if (o.ste) c = ste_quantize(c, o.enc_value_limit, o.enc_quantize_level);
            if (o.enc_truncate_limit > 0.0f) c = fminf(fmaxf(c, -o.enc_truncate_limit), o.enc_truncate_limit);
        }
        if (codes) codes[i] = c;
        if (rx) {
            const float nz = noise[i];
            float r = o.channel == 0 ? c + nz : (o.channel == 1 ? c * nz : c * (2.0f * nz - 1.0f));
            if (o.channel == 3) r = nz * c + noise[n + i];      // fading: `noise` = [fading_h | additive noise] (channel_ae.py:51-56)
            if (o.rec_quantize) r = ste_quantize(r, o.rec_quantize_limit, o.rec_quantize_level);
            rx[i] = r;
        }
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

// The same for block lengths that are a multiple of 4 (rows of both tensors 16-byte aligned): G = 2^k >= L / 4 lanes (at most a
// wave) share one block, every lane compares four positions per 16-byte load of each tensor, 64 / G blocks per wave and
// iteration.  The counts are reduced per WORKGROUP before they touch the two global counters: same-address atomics serialise
// (~12 ns each on MI355X), and one pair per wave - 8 192 waves at 50 000 blocks - was all of the old kernel's 203 us.
__global__ __launch_bounds__(256) void count_errors_vec4_kernel(const float4* __restrict__ xdec, const float4* __restrict__ u, int B, int L4, int G,
                                                                unsigned long long* __restrict__ counts) {
    __shared__ unsigned long long red[2][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane & (G - 1), slot = lane / G, per_wave = 64 / G;
    const int wave_global = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    unsigned long long bit_err = 0, blk_err = 0;
    for (long b0 = (long)wave_global * per_wave; b0 < B; b0 += (long)nwaves * per_wave) {
        const long b = b0 + slot;
        int e = 0;
        if (b < B) {
            for (int t = sub; t < L4; t += G) {
                const float4 x = xdec[(size_t)b * L4 + t], y = u[(size_t)b * L4 + t];
                e += ((x.x > 0.5f) != (y.x > 0.5f)) + ((x.y > 0.5f) != (y.y > 0.5f)) + ((x.z > 0.5f) != (y.z > 0.5f)) + ((x.w > 0.5f) != (y.w > 0.5f));
            }
        }
        for (int off = G >> 1; off > 0; off >>= 1) e += __shfl_xor(e, off);
        if (sub == 0) { bit_err += e; blk_err += (e > 0); }
    }
    for (int off = 32; off > 0; off >>= 1) {       // sum the 64 / G sub-group leaders of the wave
        bit_err += __shfl_xor(bit_err, off);
        blk_err += __shfl_xor(blk_err, off);
    }
    if (lane == 0) { red[0][wave] = bit_err; red[1][wave] = blk_err; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long a = red[0][0] + red[0][1] + red[0][2] + red[0][3], c = red[1][0] + red[1][1] + red[1][2] + red[1][3];
        if (a | c) {
            atomicAdd(&counts[0], a);
            atomicAdd(&counts[1], c);
        }
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

// Test-time noise of the reference's other channels on the device (replaces generate_noise, channels.py:37-109, and the fading
// coefficients of channel_ae.py:51-56).  Every value is a function of (seed, global element index e = ((block * L) + t) * 3 + c) on
// named Philox streams; turboae_amd/channels.py::generate_noise is the numpy mirror.  fp64 inside, one rounding to fp32.
// chi-square with vv degrees of freedom = 2 * Gamma(vv / 2), Marsaglia-Tsang (shape >= 1 because vv > 2), attempt k of element e
// draws Philox counter (e, STREAM_GAMMA, k): normal from words 0, 1, uniform from word 2.
__device__ __forceinline__ double chi_square_at(unsigned long long seed, uint64_t e, double vv) {
    const double d = 0.5 * vv - 1.0 / 3.0, c = 1.0 / sqrt(9.0 * d);
    for (uint32_t k = 0; k < 32; ++k) {
        const u32x4 w = philox4x32_10((uint32_t)e, (uint32_t)(e >> 32), STREAM_GAMMA, k, (uint32_t)seed, (uint32_t)(seed >> 32));
        const double x = sqrt(-2.0 * log(u32_to_unit_open(w.x))) * cos(6.283185307179586476925 * u32_to_unit_open(w.y));
        const double t = 1.0 + c * x, v = t * t * t;
        if (v > 0.0 && log(u32_to_unit_open(w.z)) < 0.5 * x * x + d - d * v + d * log(v)) return 2.0 * d * v;
    }
    return 2.0 * d;      // unreachable in practice (rejection probability per attempt < 5 %)
}

__device__ __forceinline__ float noise_value(const NoiseGen& g, unsigned long long seed, uint64_t e, bool good) {
    switch (g.kind) {
        case TAE_NOISE_TDIST: {
            const double t = philox_normal(seed, STREAM_NOISE, e) / sqrt(chi_square_at(seed, e, (double)g.vv) / (double)g.vv);
            return g.sigma * (float)(sqrt(((double)g.vv - 2.0) / (double)g.vv) * t);
        }
        case TAE_NOISE_RADAR: {
            const float base = g.sigma * (float)philox_normal(seed, STREAM_NOISE, e);
            const bool hit = philox_uniform(seed, STREAM_MASK, e) < (double)g.radar_prob;
            return hit ? base + (float)((double)g.radar_power * philox_normal(seed, STREAM_AUX_A, e)) : base;
        }
        case TAE_NOISE_GE_AWGN: return (good ? g.s_good : g.s_bad) * (float)philox_normal(seed, STREAM_NOISE, e);
        case TAE_NOISE_BEC:
        case TAE_NOISE_BSC: return philox_uniform(seed, STREAM_MASK, e) >= (double)g.p ? 1.0f : 0.0f;
        case TAE_NOISE_GE: return good ? 1.0f : (philox_uniform(seed, STREAM_MASK, e) < (double)g.p ? 1.0f : 0.0f);
        default: return g.sigma * (float)philox_normal(seed, STREAM_NOISE, e);        // awgn, fading
    }
}

// memoryless kinds: one element per thread and step
__global__ void gen_noise_kernel(NoiseGen g, float* __restrict__ noise, float* __restrict__ fading, size_t n, size_t e0, unsigned long long seed) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint64_t e = e0 + i;
        noise[i] = noise_value(g, seed, e, true);
        if (g.kind == TAE_NOISE_FADING) {      // channel_ae.py:53: sqrt(randn^2 + randn^2) / sqrt(3.14 / 2)
            const double a = philox_normal(seed, STREAM_AUX_A, e), b = philox_normal(seed, STREAM_AUX_B, e);
            fading[i] = (float)(sqrt(a * a + b * b) / sqrt(3.14 / 2.0));
        }
    }
}

// Gilbert-Elliott kinds: one thread walks the chain of one (block, code symbol) along time (channels.py:66-81 / 93-107): every
// chain starts good; from the good state the next state is good with probability p_gg, from the bad state with probability p_bb
__global__ void gen_noise_chain_kernel(NoiseGen g, float* __restrict__ noise, size_t n_chains, size_t first_block, int L, unsigned long long seed) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t ch = (size_t)blockIdx.x * blockDim.x + threadIdx.x; ch < n_chains; ch += stride) {
        const size_t b = ch / 3, c = ch % 3;
        bool good = true;
        for (int t = 0; t < L; ++t) {
            const uint64_t e = ((uint64_t)(first_block + b) * L + t) * 3 + c;
            noise[(b * L + t) * 3 + c] = noise_value(g, seed, e, good);
            good = philox_uniform(seed, STREAM_CHAIN, e) < (double)(good ? g.p_gg : g.p_bb);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// host-side launch helpers (called from turboae_api.cpp through turboae_internal.hpp)
template <int U>
static hipError_t launch_fused_u(bool decoder, const FusedParams& P, int grid, hipStream_t st) {
    constexpr int PT = 5;
    auto kd = dec_kernel<U, PT>;
    auto kt = dec_kernel<U, PT, true>;         // debug instantiation exporting every stack's extrinsic outputs (tae_decode_taps)
    auto ke = enc_kernel<U, PT>;
    const bool taps = decoder && P.tap_out != nullptr;
    const void* fn = decoder ? (taps ? reinterpret_cast<const void*>(kt) : reinterpret_cast<const void*>(kd)) : reinterpret_cast<const void*>(ke);
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, P.lds_bytes);
    if (e != hipSuccess) return e;
    if (taps) hipLaunchKernelGGL(kt, dim3(grid), dim3(kThreads), P.lds_bytes, st, P);
    else if (decoder) hipLaunchKernelGGL(kd, dim3(grid), dim3(kThreads), P.lds_bytes, st, P);
    else hipLaunchKernelGGL(ke, dim3(grid), dim3(kThreads), P.lds_bytes, st, P);
    return hipGetLastError();
}

hipError_t launch_fused(int U, bool decoder, const FusedParams& P, int grid, hipStream_t st) {
    switch (U) {
        case 124: return launch_fused_u<124>(decoder, P, grid, st);
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
        case 124: return launch_seg_u<124>(P, grid, st);
        case 100: return launch_seg_u<100>(P, grid, st);
        case 64: return launch_seg_u<64>(P, grid, st);
        case 32: return launch_seg_u<32>(P, grid, st);
        default: return hipErrorInvalidValue;
    }
}

int seg_lds_bytes(int U, int T, int n_layer) {
    const int rows = T + 4 * n_layer + 3 + 4;     // + up to 3 alignment rows (panel origin floored to a multiple of 4)
    size_t b = (size_t)(rows + 1) * U * 4 + (size_t)(rows + 1) * kXW * 4;
    b += (size_t)kHeadSlots * 8 * 4;     // head-combine scratch
    b = (b + 15) & ~(size_t)15;
    if (b < 2 * kThreads * sizeof(double)) b = 2 * kThreads * sizeof(double);
    return (int)b;
}

hipError_t launch_reduce_partials(const double* partials, int n, double count, double* stats, hipStream_t st) {
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(1), dim3(256), 0, st, partials, n, count, stats);
    return hipGetLastError();
}

hipError_t launch_normalize(const float* xtx, const double* stats, const float* noise, float* codes, float* rx, size_t n,
                            const NormOpts& o, hipStream_t st) {
    const int grid = (int)std::min<size_t>((n + 255) / 256, 2048);
    hipLaunchKernelGGL(normalize_kernel, dim3(grid), dim3(256), 0, st, xtx, stats, noise, codes, rx, n, o);
    return hipGetLastError();
}

hipError_t launch_count_errors(const float* xdec, const float* u, int B, int L, unsigned long long* counts, hipStream_t st) {
    if (L % 4 == 0 && ((reinterpret_cast<uintptr_t>(xdec) | reinterpret_cast<uintptr_t>(u)) & 15) == 0) {
        const int L4 = L / 4;
        int G = 1;
        while (G < L4 && G < 64) G <<= 1;
        const long waves = ((long)B * G + 63) / 64;
        const int grid = (int)std::min<long>((waves + 3) / 4, 1024);       // 4 workgroups per CU, grid-stride over the blocks
        hipLaunchKernelGGL(count_errors_vec4_kernel, dim3(grid), dim3(256), 0, st, reinterpret_cast<const float4*>(xdec),
                           reinterpret_cast<const float4*>(u), B, L4, G, counts);
        return hipGetLastError();
    }
    const int grid = std::min((B + 3) / 4, 2048);
    hipLaunchKernelGGL(count_errors_kernel, dim3(grid), dim3(256), 0, st, xdec, u, B, L, counts);
    return hipGetLastError();
}

hipError_t launch_gen_noise(const NoiseGen& g, float* noise, float* fading, size_t n_blocks, size_t first_block, int L,
                            unsigned long long seed, hipStream_t st) {
    if (g.kind == TAE_NOISE_GE || g.kind == TAE_NOISE_GE_AWGN) {
        const size_t chains = n_blocks * 3;
        const int grid = (int)((chains + 255) / 256 < 4096 ? (chains + 255) / 256 : 4096);
        hipLaunchKernelGGL(gen_noise_chain_kernel, dim3(grid), dim3(256), 0, st, g, noise, chains, first_block, L, seed);
    } else {
        const size_t n = n_blocks * (size_t)L * 3;
        const int grid = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
        hipLaunchKernelGGL(gen_noise_kernel, dim3(grid), dim3(256), 0, st, g, noise, fading, n, first_block * (size_t)L * 3, seed);
    }
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
    b += (size_t)kHeadSlots * 8 * 4;     // head-combine scratch
    if (b < 2 * kThreads * sizeof(double)) b = 2 * kThreads * sizeof(double);
    return (int)b;
}

int fused_max_positions() { return kGroups * 5 * 16; }

}  // namespace tae
