// Layer 1 of the DeepTurbo GRU stacks (DEC_LargeRNN, decoders.py:41-49,84-149; ENC_interRNN, encoders.py:251-298) as ONE kernel
// in the f16x2 representation: input projection W_ih1 * y0_t, recurrence W_hh * h_{t-1}, gate arithmetic and this direction's
// half of the Linear head - the projections GI (4 GB per stack at 16 384 blocks, written by gru_proj_h and read back by
// gru_rec_h<layer 1> through r04) never exist.
//
// The workgroup is split by hidden UNIT, not by block (gru_rec_h: one wave = 16 blocks x all 300 gate rows, A fragments from
// LDS, which is why W_ih1 could not join them): 8 waves work on ONE group of 16 blocks (N = 16 columns of every MFMA) of one
// direction,
//   * 6 "unit" waves: wave ut owns units 16 ut .. 16 ut + 15, i.e. the r, z, n row tiles of those units.  Register-resident for the
//     whole launch (168 VGPRs): W_hh hi + lo (3 tiles x (3 slabs x 8 + remainder 4)), W_ih1 hi of k-slabs 0..3 + remainder, W_ih1 lo
//     of slabs 0, 1; the lo fragments of slabs 2, 3 come from LDS.  Per step: 33 MFMAs of recurrence (the critical path) + 42 of
//     projection for the NEXT step (off it: they fill the pipe while the gates are computed and the other waves arrive);
//   * the remainder wave: units 96..99 as one mixed row tile (row 4 qq + i = r, z, n_h, n_i of unit 96 + qq; all in registers),
//     plus the head tile: W_lin[:, dir H .. dir H + H) * h_t - 16 floats per position leave the kernel, as before; its four
//     single-accumulator chains are issued product by product in turn;
//   * the staging wave: y0 rows (16 x 800 bytes per step, in the three regions of turboae_y0.hpp) from HBM into LDS in B-fragment order by LDS-DMA (buffer_load ... lds: the
//     fragment tiles are lane-linear), issued at the top of the step before the one that reads them and waited for at its barrier.
// Waves w and w + 4 share a SIMD, so the two light waves are 3 and 7 and every other SIMD carries two unit waves.  To level the
// SIMDs (r05 v1 had 2 x 93 MFMAs per step on three of them and 42 on the fourth, and ran at exactly the 81 % of the sustained
// MFMA rate that split allows) the two light waves are also HELPERS: k-slabs 4 and 5 of the projection of EVERY unit tile
// (1 + 5 unit tiles: 18 + 90 MFMAs, W_ih1 hi of those slabs in their registers, lo from LDS) are computed by them, two steps ahead
// of the recurrence, and handed over through LDS as accumulator-init tiles (`pb`, which replace the unit waves' bias rows):
// 2 x 75 MFMAs on every SIMD.
// h_t is exchanged through LDS as B fragments (hi | lo halves, 8 KB, double-buffered): one s_barrier per step.
// LDS: W_ih1 lo (slabs 2..5) 73 728 + bias rows 1 616 + h 2 x 8 192 + y0 2 x (10 240 + 4 096) + partial tiles 2 x 18 432
//      = 157 264 of 163 840 bytes.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <type_traits>
#include "turboae_internal.hpp"
#include "turboae_device.hpp"
#include "turboae_y0.hpp"

// timing experiments on this kernel (results wrong; -DTAE_EXPERIMENT builds only): 1 linear gates (no exp / rcp), 2 no step barrier,
// 4 no projection MFMAs in the unit waves' step, 8 no recurrence MFMAs in the unit waves' step, 16 no y0 loads in the staging wave's step,
// 32 no helper partials in the step loops, 128 cycle stamps (s_memtime at the phase boundaries of steps
// 40..47 of workgroup (0, 0)'s first group, printed per wave at the end of the kernel; results stay correct)
#if defined(TAE_EXPERIMENT) && defined(TAE_L1F_X)
constexpr int kL1fX = TAE_L1F_X;
#else
#ifdef TAE_L1F_X
#error "TAE_L1F_X is a timing experiment that breaks the results: build with -DTAE_EXPERIMENT as well"
#endif
constexpr int kL1fX = 0;
#endif

namespace tae {

namespace {

using u32x4v = __attribute__((ext_vector_type(4))) uint32_t;
using lds_q4 = const u32x4v __attribute__((address_space(3)));
using lds_f4c = const f32x4 __attribute__((address_space(3)));
using lds_w4 = u32x4v __attribute__((address_space(3)));
using lds_w2 = u32x2v __attribute__((address_space(3)));
using lds_ptr = char __attribute__((address_space(3)))*;

constexpr int kLdsW = 6 * 3 * 4 * 1024;             // W_ih1 lo fragments of k-slabs 2..5: [ut][gate][slab - 2][lane][8 halves]
constexpr int kBiasB = 6 * 4 * 64 + 64 + 16;        // [ut][r, z, n_i, n_h][16] + remainder-tile init row + (2^-S, 2^-S_head, 0, 0)
constexpr int kHB = kLdsW + kBiasB;                 // h exchange: 3 slabs x (hi | lo) + remainder (b1 | b2)
constexpr int kHBsz = 8192;
constexpr int kYM = kHB + 2 * kHBsz;                // y0 of a step, k-slabs 0..3 x (hi | lo) + remainder (b1 | b2): read by the owners' projection
#ifndef TAE_L1F_HS
#define TAE_L1F_HS 2
#endif
constexpr int kHS = TAE_L1F_HS;                     // k-slabs of every unit tile's projection the helpers compute (the last kHS of 6);
                                                    // 1 measured 2 % slower than 2 (profiles/r05_gru_l1f_hs1_ab.txt): the staging wave's SIMD idles less than the unit waves' gain
constexpr int kMS = 6 - kHS;                        // ... and the owners' share
constexpr int kYMsz = (2 * kMS + 2) * 1024;
constexpr int kYH = kYM + 2 * kYMsz;                // y0 of a step, k-slabs 4, 5 x (hi | lo): read by the helpers, two steps ahead
constexpr int kYHsz = 2 * kHS * 1024;
constexpr int kPB = kYH + 2 * kYHsz;                // helper partials = accumulator-init tiles [ut][r, z, n_i][lane][4 floats]
constexpr int kPBsz = 18 * 1024;
constexpr int kLds = kPB + 2 * kPBsz;
constexpr int kUnitB = 42 * 1024, kRemB = 27 * 1024, kLo01B = 36 * 1024;
constexpr int kDbgSteps = 8, kDbgS0 = 40, kDbgStamps = 8;
constexpr int kDbgB = (kL1fX & 128) ? 8 * kDbgSteps * kDbgStamps * 4 : 0;
static_assert(kLds + kDbgB <= 160 * 1024, "LDS budget");
static_assert(kUnitB == GruL1fLayout::kUnitB && kRemB == GruL1fLayout::kRemB && kLo01B == GruL1fLayout::kLo01B &&
              kLdsW + kBiasB == GruL1fLayout::kLdsImgB, "host packing");

// sigmoid / tanh of x 2^-S given cs = -log2(e) 2^-S, ct = 2 log2(e) 2^-S: the accumulators' power-of-two scale rides in the
// exp2 argument's constant - bit for bit what scaling x first gives (a power of two commutes with every rounding here)
__device__ __forceinline__ float sigm_s(float x, float cs) {
    if (kL1fX & 1) return fmaf(x * cs, -0.17f, 0.5f);
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * cs));
}
__device__ __forceinline__ float tanh_s(float x, float ct) {
    if (kL1fX & 1) return x * ct * 0.17f;
    return fmaf(-2.0f, __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * ct)), 1.0f);
}
__device__ __forceinline__ h8 lds_h8(lds_cptr p) { return __builtin_bit_cast(h8, *reinterpret_cast<lds_q4*>(p)); }
__device__ __forceinline__ h8 glb_h8(const char* p) { return __builtin_bit_cast(h8, *reinterpret_cast<const u32x4v*>(p)); }

// LDS writes of this wave are done and visible, then the workgroup barrier.  NOT __syncthreads(): that also drains vmcnt, and the
// staging wave's loads / the head stores are meant to stay in flight across steps.
__device__ __forceinline__ void step_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    if (!(kL1fX & 2)) __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// timing experiments: keep the operands alive, issue no MFMA
__device__ __forceinline__ void skip3(f32x4 (&acc)[3], const h8 (&a)[3], h8 b0, h8 b1) {
#pragma unroll
    for (int g = 0; g < 3; ++g) asm volatile("" : "+v"(acc[g]) : "v"(a[g]), "v"(b0), "v"(b1));
}
// one 32-k slab, three gate tiles: hi*lo, lo*hi, hi*hi, each product over the three accumulators (independent chains)
__device__ __forceinline__ void mma3(f32x4 (&acc)[3], const h8 (&ah)[3], const h8 (&al)[3], h8 bh, h8 bl) {
#pragma unroll
    for (int g = 0; g < 3; ++g) acc[g] = mfma16x16x32h(ah[g], bl, acc[g]);
#pragma unroll
    for (int g = 0; g < 3; ++g) acc[g] = mfma16x16x32h(al[g], bh, acc[g]);
#pragma unroll
    for (int g = 0; g < 3; ++g) acc[g] = mfma16x16x32h(ah[g], bh, acc[g]);
}
// ... the products of the hi fragments (registers) first, the lo fragments' last: they arrive from LDS during the first six
__device__ __forceinline__ void mma3_lo_last(f32x4 (&acc)[3], const h8 (&ah)[3], const h8 (&al)[3], h8 bh, h8 bl) {
#pragma unroll
    for (int g = 0; g < 3; ++g) acc[g] = mfma16x16x32h(ah[g], bl, acc[g]);
#pragma unroll
    for (int g = 0; g < 3; ++g) acc[g] = mfma16x16x32h(ah[g], bh, acc[g]);
#pragma unroll
    for (int g = 0; g < 3; ++g) acc[g] = mfma16x16x32h(al[g], bh, acc[g]);
}
// K = 8 remainder slab, the three products of a slab in ONE MFMA (late r06; through r06 two, on gru_rec_h's [4 hi | 4 lo] fragments):
// per lane A = [hi k0 k1 | hi k0 k1 | lo k0 k1 | 0 0], B = [hi k0 k1 | lo k0 k1 | hi k0 k1 | 0 0] - the 32 k slots of the instruction
// hold hi*hi, hi*lo and lo*hi of the lane group's two real k (recurrence: unit 96 + kq and nothing; projection: y0 halves 192 + 2 kq, + 1)
__device__ __forceinline__ void mma3r(f32x4 (&acc)[3], const h8 (&ar)[3], h8 b) {
#pragma unroll
    for (int g = 0; g < 3; ++g) acc[g] = mfma16x16x32h(ar[g], b, acc[g]);
}
__device__ __forceinline__ void mma1(f32x4& acc, h8 ah, h8 al, h8 bh, h8 bl) {
    acc = mfma16x16x32h(ah, bl, acc);
    acc = mfma16x16x32h(al, bh, acc);
    acc = mfma16x16x32h(ah, bh, acc);
}
__device__ __forceinline__ void mma1r(f32x4& acc, h8 ar, h8 b) { acc = mfma16x16x32h(ar, b, acc); }

template <int N, class F, int I = 0>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<N, F, I + 1>(static_cast<F&&>(f));
    }
}

// issue-order hint for one stage: ND LDS reads (operands of a LATER stage) first, then NM MFMAs with NV vector-ALU / transcendental
// instructions after each
template <int ND, int NM, int NV>
__device__ __forceinline__ void pin() {
#ifdef TAE_L1F_NOPIN
    return;
#endif
    if (ND > 0) __builtin_amdgcn_sched_group_barrier(0x100, ND, 0);
#pragma unroll
    for (int m = 0; m < NM; ++m) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if (NV > 0) __builtin_amdgcn_sched_group_barrier(0x402, NV, 0);
    }
}

struct Ctx {
    const GruL1fParams& P;
    const char* wdir;
    lds_cptr lds;
    int lane, n, q, dir, L;
    float inv, inv_head;
    int wave;
};

// experiment 128: shader-clock stamp i of step s of this wave (first group of workgroup (0, 0) only)
__device__ __forceinline__ void stamp(const Ctx& c, bool first, int s, int i) {
    if constexpr ((kL1fX & 128) != 0) {
        if (first && blockIdx.x == 0 && blockIdx.y == 0 && s >= kDbgS0 && s < kDbgS0 + kDbgSteps) {
            __builtin_amdgcn_sched_barrier(0);
            const uint32_t t = (uint32_t)__builtin_readcyclecounter();
            if (c.lane == 0) *reinterpret_cast<uint32_t __attribute__((address_space(3)))*>((lds_ptr)c.lds + kLds + ((c.wave * kDbgSteps + (s - kDbgS0)) * kDbgStamps + i) * 4) = t;
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}
__device__ __forceinline__ void stamp_print(const Ctx& c, int nst) {
    if constexpr ((kL1fX & 128) != 0) {
        if (blockIdx.x == 0 && blockIdx.y == 0 && c.lane == 0) {
            for (int s = 0; s < kDbgSteps; ++s) {
                const uint32_t __attribute__((address_space(3)))* r = reinterpret_cast<const uint32_t __attribute__((address_space(3)))*>((lds_ptr)c.lds + kLds + ((c.wave * kDbgSteps + s) * kDbgStamps) * 4);
                const uint32_t t0 = r[0];
                printf("l1f wave %d step %d: t0 %u  +%u +%u +%u +%u +%u +%u (%d stamps)\n", c.wave, kDbgS0 + s, t0, r[1] - t0, r[2] - t0, r[3] - t0, r[4] - t0, r[5] - t0, r[6] - t0, nst);
            }
        }
    }
}

// Buffer parity: M[k & 1] / H[k & 1] hold the y0 fragments of step k, pb[k & 1] the partial tiles of step k, hb[k & 1] the B
// fragments of h_{k-1}.  During step s
//   unit waves   read hb[s], M[s + 1], pb[s + 1]              write hb[s + 1]
//   helpers      read H[s + 2]                                write pb[s + 2]
//   staging      (registers, fetched during step s - 1)       write M[s + 2], H[s + 3]
// so nothing written in a step is read in it, and one barrier per step orders everything.

// ---- unit wave ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void unit_wave(const Ctx& c, int ut) {
    const int lane = c.lane, L = c.L;
    const char* wr = c.wdir + (size_t)ut * kUnitB + lane * 16;
    const char* wlo = c.wdir + (size_t)6 * kUnitB + kRemB + (size_t)ut * (6 * 1024) + lane * 16;      // W_ih1 lo of slabs 0, 1: [gate][slab]
    h8 hh_hi[3][3], hh_lo[3][3], hh_r[3], ih_hi[kMS][3], ih_lo[2][3], ih_r[3];       // [slab][gate]
#pragma unroll
    for (int g = 0; g < 3; ++g) {
#pragma unroll
        for (int sl = 0; sl < 3; ++sl) {
            hh_hi[sl][g] = glb_h8(wr + (g * 7 + 2 * sl) * 1024);
            hh_lo[sl][g] = glb_h8(wr + (g * 7 + 2 * sl + 1) * 1024);
        }
        hh_r[g] = glb_h8(wr + (g * 7 + 6) * 1024);
#pragma unroll
        for (int sl = 0; sl < kMS; ++sl) ih_hi[sl][g] = glb_h8(wr + (21 + g * 7 + sl) * 1024);
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) ih_lo[sl][g] = glb_h8(wlo + (g * 2 + sl) * 1024);
        ih_r[g] = glb_h8(wr + (21 + g * 7 + 6) * 1024);
    }
    const lds_cptr wl = c.lds + ut * (12 * 1024) + lane * 16;           // W_ih1 lo of slabs 2..5: [gate][slab - 2] of this unit tile
    const lds_cptr bias_nh = c.lds + kLdsW + ut * 256 + 3 * 64 + c.q * 16;
    const lds_cptr hb = c.lds + kHB + lane * 16, ym = c.lds + kYM + lane * 16;
    const lds_cptr pb = c.lds + kPB + ut * (3 * 1024) + lane * 16;
    const lds_ptr hw = (lds_ptr)(c.lds + kHB + (ut >> 1) * 2048 + lane * 16 + (ut & 1) * 8);
    const float cs = -1.44269504088896341f * c.inv, ct = 2.88539008177792681f * c.inv;

    for (int grp = blockIdx.x; grp < c.P.ngroups; grp += gridDim.x) {
        // h_{-1} = 0: this wave's slots of buffer 0
        *reinterpret_cast<lds_w2*>(hw) = u32x2v{0, 0};
        *reinterpret_cast<lds_w2*>(hw + 1024) = u32x2v{0, 0};
        step_barrier();                                   // B0: y0 of steps 0 and 1 staged, h buffer 0 cleared
        step_barrier();                                   // B1: the helpers' partial tiles of steps 0 and 1 are in pb
        f32x4 gi[3];
        {   // the owner's share of step 0's projection
#pragma unroll
            for (int g = 0; g < 3; ++g) gi[g] = *reinterpret_cast<lds_f4c*>(pb + g * 1024);
#pragma unroll
            for (int sl = 0; sl < kMS; ++sl) {
                const h8 bh = lds_h8(ym + sl * 2048), bl = lds_h8(ym + sl * 2048 + 1024);
                h8 al[3];
#pragma unroll
                for (int g = 0; g < 3; ++g) al[g] = sl < 2 ? ih_lo[sl < 2 ? sl : 0][g] : lds_h8(wl + (g * 4 + sl - 2) * 1024);
                mma3(gi, ih_hi[sl], al, bh, bl);
            }
            mma3r(gi, ih_r, lds_h8(ym + kMS * 2048));
        }
        f32x4 h = {0.f, 0.f, 0.f, 0.f};
        step_barrier();                                   // B2: M[0], pb[0] may be refilled
#pragma unroll 1
        for (int s = 0; s < L; ++s) {
            // One step = ONE scheduling region with a pinned issue order:
            //   recurrence (the critical path): B fragments of h_{s-1} one slab ahead of their MFMAs;
            //   the owner's share of step s + 1's projection (off it): B fragments one slab ahead, the gate arithmetic of step s dealt
            //   out between its MFMAs (two vector-ALU instructions per MFMA): the matrix pipe never waits for the gates.
            const int p1 = (s + 1) & 1;
            stamp(c, grp == (int)blockIdx.x, s, 0);
            const lds_cptr hc = hb + (s & 1) * kHBsz, y = ym + p1 * kYMsz, pbn = pb + p1 * kPBsz;
            f32x4 acc[3] = {gi[0], gi[1], *reinterpret_cast<lds_f4c*>(bias_nh)};
            const f32x4 gin = gi[2];
            h8 xh[2], xl[2];                                   // B-fragment ring
            xh[0] = lds_h8(hc); xl[0] = lds_h8(hc + 1024);
            xh[1] = lds_h8(hc + 2048); xl[1] = lds_h8(hc + 3072);
            if (kL1fX & 8) skip3(acc, hh_hi[0], xh[0], xl[0]); else mma3(acc, hh_hi[0], hh_lo[0], xh[0], xl[0]);
            pin<5, 9, 0>();
            xh[0] = lds_h8(hc + 4096); xl[0] = lds_h8(hc + 5120);
            if (kL1fX & 8) skip3(acc, hh_hi[1], xh[1], xl[1]); else mma3(acc, hh_hi[1], hh_lo[1], xh[1], xl[1]);
            pin<2, 9, 0>();
            xh[1] = lds_h8(hc + 6144);                                  // remainder slab: one B tile
            if (kL1fX & 8) skip3(acc, hh_hi[2], xh[0], xl[0]); else mma3(acc, hh_hi[2], hh_lo[2], xh[0], xl[0]);
            pin<1, 9, 0>();
            xh[0] = lds_h8(y); xl[0] = lds_h8(y + 1024);
#pragma unroll
            for (int g = 0; g < 3; ++g) gi[g] = *reinterpret_cast<lds_f4c*>(pbn + g * 1024);       // helpers' partial (bias + slabs 4, 5)
            if (kL1fX & 8) skip3(acc, hh_r, xh[1], xh[1]); else mma3r(acc, hh_r, xh[1]);
            pin<5, 3, 0>();
            stamp(c, grp == (int)blockIdx.x, s, 1);
            f32x4 hn;
            static_for<kMS>([&](auto SL) {
                constexpr int sl = decltype(SL)::value, cur = sl & 1, nxt = cur ^ 1;
                xh[nxt] = lds_h8(y + (sl + 1) * 2048);                                                       // (the slab after the last one is the remainder's one tile)
                if constexpr (sl + 1 < kMS) xl[nxt] = lds_h8(y + (sl + 1) * 2048 + 1024);
                if constexpr (sl < 2) {
                    if (kL1fX & 4) skip3(gi, ih_hi[sl], xh[cur], xl[cur]); else mma3_lo_last(gi, ih_hi[sl], ih_lo[sl], xh[cur], xl[cur]);
                } else {
                    h8 al[3];                                  // lo fragments of THIS slab, used by its last three MFMAs
#pragma unroll
                    for (int g = 0; g < 3; ++g) al[g] = lds_h8(wl + (g * 4 + sl - 2) * 1024);
                    if (kL1fX & 4) skip3(gi, al, xh[cur], xl[cur]); else mma3_lo_last(gi, ih_hi[sl], al, xh[cur], xl[cur]);
                }
                if constexpr (sl < 4) {
                    const int i = sl;
                    const float r = sigm_s(acc[0][i], cs);
                    const float z = sigm_s(acc[1][i], cs);
                    const float nn = tanh_s(fmaf(r, acc[2][i], gin[i]), ct);
                    hn[i] = fmaf(z, h[i] - nn, nn);
                    pin<(sl < 2 ? 2 : 5), 9, 2>();
                } else pin<5, 9, 0>();
            });
            stamp(c, grp == (int)blockIdx.x, s, 2);
            h = hn;
            h4 nhi, nlo;
            split4(hn, nhi, nlo);
            const lds_ptr hn_w = hw + p1 * kHBsz;
            *reinterpret_cast<lds_w2*>(hn_w) = __builtin_bit_cast(u32x2v, nhi);
            *reinterpret_cast<lds_w2*>(hn_w + 1024) = __builtin_bit_cast(u32x2v, nlo);
            if (kL1fX & 4) skip3(gi, ih_r, xh[kMS & 1], xh[kMS & 1]); else mma3r(gi, ih_r, xh[kMS & 1]);
            pin<0, 3, 2>();
            stamp(c, grp == (int)blockIdx.x, s, 3);
            step_barrier();
        }
        step_barrier();                                   // the remainder wave's last head product has read the h buffer
    }
    stamp_print(c, 4);
}

// Helper share of the projection: for NU unit tiles starting at u0, tile (ut, g) <- bias + W_ih1[rows, k-slabs 4, 5] * y0, written
// to `pbw` as accumulator-init tiles.  `yh`: this step's H fragments; A hi fragments in registers, lo from LDS.
template <int NU>
__device__ __forceinline__ void helper_partials(const h8 (&a_hi)[NU][kHS][3], lds_cptr lds, int lane, int u0, int q, lds_cptr yh, lds_ptr pbw) {
    h8 bh[kHS], bl[kHS];
#pragma unroll
    for (int j = 0; j < kHS; ++j) { bh[j] = lds_h8(yh + j * 2048); bl[j] = lds_h8(yh + j * 2048 + 1024); }
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int ut = u0 + u;
        const lds_cptr wl = lds + ut * (12 * 1024) + lane * 16;
        const lds_cptr bias = lds + kLdsW + ut * 256 + q * 16;
        f32x4 acc[3];
        h8 al[kHS][3];
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            acc[g] = *reinterpret_cast<lds_f4c*>(bias + g * 64);
#pragma unroll
            for (int j = 0; j < kHS; ++j) al[j][g] = lds_h8(wl + (g * 4 + kMS - 2 + j) * 1024);
        }
#pragma unroll
        for (int j = 0; j < kHS; ++j) mma3_lo_last(acc, a_hi[u][j], al[j], bh[j], bl[j]);
#pragma unroll
        for (int g = 0; g < 3; ++g) *reinterpret_cast<f32x4 __attribute__((address_space(3)))*>(pbw + (ut * 3 + g) * 1024) = acc[g];
    }
}

// ---- remainder wave: units 96..99 + the head tile + helper for unit tile 0 -------------------------------------------------
__device__ __forceinline__ void rem_wave(const Ctx& c) {
    const int lane = c.lane, L = c.L, n = c.n, q = c.q, dir = c.dir;
    const char* wr = c.wdir + (size_t)6 * kUnitB + lane * 16;
    h8 hh_hi[3], hh_lo[3], hh_r, ih_hi[6], ih_lo[6], ih_r, hd_hi[3], hd_lo[3], hd_r;
#pragma unroll
    for (int sl = 0; sl < 3; ++sl) { hh_hi[sl] = glb_h8(wr + (2 * sl) * 1024); hh_lo[sl] = glb_h8(wr + (2 * sl + 1) * 1024); }
    hh_r = glb_h8(wr + 6 * 1024);
#pragma unroll
    for (int sl = 0; sl < 6; ++sl) { ih_hi[sl] = glb_h8(wr + (7 + 2 * sl) * 1024); ih_lo[sl] = glb_h8(wr + (8 + 2 * sl) * 1024); }
    ih_r = glb_h8(wr + 19 * 1024);
#pragma unroll
    for (int sl = 0; sl < 3; ++sl) { hd_hi[sl] = glb_h8(wr + (20 + 2 * sl) * 1024); hd_lo[sl] = glb_h8(wr + (21 + 2 * sl) * 1024); }
    hd_r = glb_h8(wr + 26 * 1024);
    h8 hp_hi[1][kHS][3];                                                  // helper: W_ih1 hi of k-slabs 4, 5 of unit tile 0
#pragma unroll
    for (int u = 0; u < 1; ++u)
#pragma unroll
        for (int sl = 0; sl < kHS; ++sl)
#pragma unroll
            for (int g = 0; g < 3; ++g) hp_hi[u][sl][g] = glb_h8(c.wdir + (size_t)u * kUnitB + (21 + g * 7 + kMS + sl) * 1024 + lane * 16);
    const lds_cptr bias = c.lds + kLdsW + 6 * 256 + q * 16;             // init row: (r, z, b_hn, b_in) of unit 96 + q
    const lds_cptr hb = c.lds + kHB + lane * 16, ym = c.lds + kYM + lane * 16, yh = c.lds + kYH + lane * 16;
    const lds_ptr hw = (lds_ptr)(c.lds + kHB + 6144 + lane * 16);
    const lds_ptr pbw = (lds_ptr)(c.lds + kPB + lane * 16);
    const float inv_head = c.inv_head, cs = -1.44269504088896341f * c.inv, ct = 2.88539008177792681f * c.inv;

    // this wave's own tile: bias + k-slabs 4, 5 (at helper time, two steps ahead) ...
    auto own_part = [&](lds_cptr yhk) {
        f32x4 g = *reinterpret_cast<lds_f4c*>(bias);
#pragma unroll
        for (int j = 0; j < kHS; ++j) mma1(g, ih_hi[kMS + j], ih_lo[kMS + j], lds_h8(yhk + j * 2048), lds_h8(yhk + j * 2048 + 1024));
        return g;
    };
    // ... + k-slabs 0..3 and the remainder (one step ahead, like the unit waves)
    auto own_main = [&](f32x4& gi, lds_cptr y) {
#pragma unroll
        for (int sl = 0; sl < kMS; ++sl) mma1(gi, ih_hi[sl], ih_lo[sl], lds_h8(y + sl * 2048), lds_h8(y + sl * 2048 + 1024));
        mma1r(gi, ih_r, lds_h8(y + kMS * 2048));
    };
    // head products of the state in h buffer `hc` -> hpart[(grp L + t) 16 + n][dir][8]
    auto head = [&](lds_cptr hc, const __amdgpu_buffer_rsrc_t& rs, int t) {
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int sl = 0; sl < 3; ++sl) {
            const h8 bh = lds_h8(hc + sl * 2048), bl = lds_h8(hc + sl * 2048 + 1024);
            mma1(a, hd_hi[sl], hd_lo[sl], bh, bl);
        }
        mma1r(a, hd_r, lds_h8(hc + 6144));
        const uint32_t v = q < 2 ? (uint32_t)(n * 64 + dir * 32 + q * 16) : 0x80000000u;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v, a * inv_head), rs, v, (uint32_t)t * 1024u, 0);
    };

    for (int grp = blockIdx.x; grp < c.P.ngroups; grp += gridDim.x) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(c.P.hpart + (size_t)grp * L * 256, 0, L * 1024, 0x00020000);
        *reinterpret_cast<lds_w4*>(hw) = u32x4v{0, 0, 0, 0};
        step_barrier();                                   // B0
        helper_partials<1>(hp_hi, c.lds, lane, 0, q, yh, pbw);                             // steps 0 and 1
        __builtin_amdgcn_sched_barrier(0);
        helper_partials<1>(hp_hi, c.lds, lane, 0, q, yh + kYHsz, pbw + kPBsz);
        f32x4 gi = own_part(yh), gp1 = own_part(yh + kYHsz);                            // own tile: partials of steps 0 and 1
        step_barrier();                                   // B1
        own_main(gi, ym);
        float hr = 0.0f;
        step_barrier();                                   // B2
#pragma unroll 1
        for (int s = 0; s < L; ++s) {
            const int p0 = s & 1, p1 = p0 ^ 1;
            stamp(c, grp == (int)blockIdx.x, s, 0);
            const lds_cptr hc = hb + p0 * kHBsz, y = ym + p1 * kYMsz, yk = yh + p0 * kYHsz;
            // Four single-accumulator chains - this tile's recurrence (acc), the head products of h_{s-1} (hd: the same B fragments),
            // the own tile's projection of step s + 1 (gi: k-slabs 0..3 + remainder) and of step s + 2 (gp: the helper slabs) - issued
            // product by product in turn: a chain's next MFMA is four issue slots behind the one it depends on, so none of them
            // waits for its accumulator (issued one after the other, each dependent MFMA held the pipe - and the staging wave - for its latency)
            f32x4 acc = gi, hd = {0.f, 0.f, 0.f, 0.f}, gp = *reinterpret_cast<lds_f4c*>(bias);
            gi = gp1;
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const h8 bh = lds_h8(hc + r * 2048), bl = lds_h8(hc + r * 2048 + 1024);
                const h8 mh = lds_h8(y + r * 2048), ml = lds_h8(y + r * 2048 + 1024);
                h8 kh = bh, kl = bl;
                if (r < kHS) { kh = lds_h8(yk + r * 2048); kl = lds_h8(yk + r * 2048 + 1024); }
                acc = mfma16x16x32h(hh_hi[r], bl, acc); hd = mfma16x16x32h(hd_hi[r], bl, hd); gi = mfma16x16x32h(ih_hi[r], ml, gi);
                if (r < kHS) gp = mfma16x16x32h(ih_hi[kMS + r], kl, gp);
                acc = mfma16x16x32h(hh_lo[r], bh, acc); hd = mfma16x16x32h(hd_lo[r], bh, hd); gi = mfma16x16x32h(ih_lo[r], mh, gi);
                if (r < kHS) gp = mfma16x16x32h(ih_lo[kMS + r], kh, gp);
                acc = mfma16x16x32h(hh_hi[r], bh, acc); hd = mfma16x16x32h(hd_hi[r], bh, hd); gi = mfma16x16x32h(ih_hi[r], mh, gi);
                if (r < kHS) gp = mfma16x16x32h(ih_hi[kMS + r], kh, gp);
            }
            {
                const h8 br = lds_h8(hc + 6144);
#pragma unroll
                for (int r = 3; r < kMS; ++r) {                     // the projection's remaining slabs, the remainders of the two h chains between its products
                    const h8 mh = lds_h8(y + r * 2048), ml = lds_h8(y + r * 2048 + 1024);
                    gi = mfma16x16x32h(ih_hi[r], ml, gi);
                    if (r == 3) acc = mfma16x16x32h(hh_r, br, acc);
                    gi = mfma16x16x32h(ih_lo[r], mh, gi);
                    if (r == 3) hd = mfma16x16x32h(hd_r, br, hd);
                    gi = mfma16x16x32h(ih_hi[r], mh, gi);
                }
            }
            stamp(c, grp == (int)blockIdx.x, s, 1);
            if (!(kL1fX & 32)) helper_partials<1>(hp_hi, c.lds, lane, 0, q, yk, pbw + p0 * kPBsz);   // unit tile 0, step s + 2
            mma1r(gi, ih_r, lds_h8(y + kMS * 2048));
            gp1 = gp;
            stamp(c, grp == (int)blockIdx.x, s, 2);
            if (s > 0) {                                            // Linear head on h_{s-1} (the state this step started from)
                const uint32_t v = q < 2 ? (uint32_t)(n * 64 + dir * 32 + q * 16) : 0x80000000u;
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v, hd * inv_head), rs, v, (uint32_t)(dir ? L - s : s - 1) * 1024u, 0);
            }
            stamp(c, grp == (int)blockIdx.x, s, 3);
            stamp(c, grp == (int)blockIdx.x, s, 4);
            const float r = sigm_s(acc[0], cs);
            const float z = sigm_s(acc[1], cs);
            const float nn = tanh_s(fmaf(r, acc[2], acc[3]), ct);
            hr = fmaf(z, hr - nn, nn);
            const _Float16 hi = (_Float16)hr;
            const _Float16 lo = (_Float16)(hr - (float)hi);
            const h8 br = {hi, 0, lo, 0, hi, 0, 0, 0};               // [hi k0 k1 | lo k0 k1 | hi k0 k1 | 0 0], k0 = unit 96 + q, no k1
            const lds_ptr hn_w = hw + p1 * kHBsz;
            *reinterpret_cast<lds_w4*>(hn_w) = __builtin_bit_cast(u32x4v, br);
            stamp(c, grp == (int)blockIdx.x, s, 5);
            step_barrier();
        }
        head(hb + (L & 1) * kHBsz, rs, dir ? 0 : L - 1);
        step_barrier();
    }
    stamp_print(c, 6);
}

// ---- staging wave: y0 rows -> B fragments in LDS; helper for unit tiles 1..5 ------------------------------------------------------
// LDS-DMA (buffer_load ... lds): every lane's 16 bytes land at the wave-uniform LDS address + lane * 16 - exactly a fragment tile.
// The staged rows never touch a register and cost no ds_write; the staging wave's registers go to its helper share instead.
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rs, lds_ptr dst, uint32_t v, uint32_t so) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, dst, 16, v, so, 0, 0);
}
// Lane (n, q) takes piece 4 sl + q of row n for k-slab sl, from the three regions of a step (turboae_y0.hpp): slabs 0..2 lie in A,
// slab 3 is the shared piece 12 (q = 0) | the first three of B, slabs 4 and 5 the next eight of B, the K = 16 remainder B's last one.
struct Y0Lane { uint32_t vA, v3, v3lo, vB; };
__device__ __forceinline__ Y0Lane y0_lane(int n, int q) {
    Y0Lane v;
    v.vA = y0_piece(n, q);
    v.v3 = y0_piece(n, 12 + q);
    v.v3lo = v.v3 + y0_lo_add(12 + q);
    v.vB = y0_piece(n, 16 + q);
    return v;
}
__device__ __forceinline__ void ym_dma(__amdgpu_buffer_rsrc_t rs, lds_ptr y, const Y0Lane& v, uint32_t so) {       // y: tile base (no lane term)
    static_assert(kMS == 4 && kHS == 2, "slab split of the Y0 regions");
#pragma unroll
    for (int sl = 0; sl < 3; ++sl) {
        dma16(rs, y + (2 * sl) * 1024, v.vA + sl * 64, so);
        dma16(rs, y + (2 * sl + 1) * 1024, v.vA + kY0PlaneAB + sl * 64, so);
    }
    dma16(rs, y + 6 * 1024, v.v3, so);
    dma16(rs, y + 7 * 1024, v.v3lo, so);
}
__device__ __forceinline__ void yh_dma(__amdgpu_buffer_rsrc_t rs, lds_ptr y, const Y0Lane& v, uint32_t so) {
#pragma unroll
    for (int sl = 0; sl < kHS; ++sl) {
        dma16(rs, y + (2 * sl) * 1024, v.vB + sl * 64, so);
        dma16(rs, y + (2 * sl + 1) * 1024, v.vB + kY0PlaneAB + sl * 64, so);
    }
}
// the K = 8 remainder of a step (y0 halves 192..199) goes through registers: lane (n, kq) takes k = 192 + 2 kq, + 1 of row n, re-paired
// into the one B tile of mma3r: [hi k0 k1 | lo k0 k1 | hi k0 k1 | 0 0]
struct YR { uint32_t rh, rl; };
__device__ __forceinline__ void yr_load(YR& r, __amdgpu_buffer_rsrc_t rs, uint32_t vr, uint32_t so) {
    r.rh = __builtin_amdgcn_raw_buffer_load_b32(rs, vr, so, 0);
    r.rl = __builtin_amdgcn_raw_buffer_load_b32(rs, vr + kY0PlaneAB, so, 0);
}
__device__ __forceinline__ void yr_store(const YR& r, lds_ptr y) {            // y: this lane's slot of the M part
    *reinterpret_cast<lds_w4*>(y + kMS * 2048) = u32x4v{r.rh, r.rl, r.rh, 0};
}
__device__ __forceinline__ void dma_landed() { __builtin_amdgcn_s_waitcnt(0x0F70); }       // vmcnt(0): an LDS-DMA is a pending LDS write on the VM counter

__device__ __forceinline__ void stage_wave(const Ctx& c) {
    const int lane = c.lane, L = c.L, n = c.n, q = c.q, dir = c.dir;
    h8 hp_hi[5][kHS][3];                                                  // helper: W_ih1 hi of k-slabs 4, 5 of unit tiles 1..5
#pragma unroll
    for (int u = 0; u < 5; ++u)
#pragma unroll
        for (int sl = 0; sl < kHS; ++sl)
#pragma unroll
            for (int g = 0; g < 3; ++g) hp_hi[u][sl][g] = glb_h8(c.wdir + (size_t)(1 + u) * kUnitB + (21 + g * 7 + kMS + sl) * 1024 + lane * 16);
    const lds_cptr yh = c.lds + kYH + lane * 16;
    const lds_ptr ymu = (lds_ptr)(c.lds + kYM), yhu = (lds_ptr)(c.lds + kYH);          // tile bases of the DMA
    const lds_ptr ymw = (lds_ptr)(c.lds + kYM + lane * 16);
    const lds_ptr pbw = (lds_ptr)(c.lds + kPB + lane * 16);
    const Y0Lane v0 = y0_lane(n, q);
    const uint32_t vr = y0_half(n, 192 + 2 * q);                                         // remainder k = 192 + 2 kq, + 1: piece 24, region B
    for (int grp = blockIdx.x; grp < c.P.ngroups; grp += gridDim.x) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(c.P.y0) + (size_t)grp * L * (16 * 800), 0, L * 16 * 800, 0x00020000);
        // byte offset of step k's 16 rows (steps past the end: the last step again - fetched, never used)
        const auto so = [&](int k) { k = k < L ? k : L - 1; return (uint32_t)__builtin_amdgcn_readfirstlane(dir ? L - 1 - k : k) * (16 * 800u); };
        YR r0, r1;
        ym_dma(rs, ymu, v0, so(0)); ym_dma(rs, ymu + kYMsz, v0, so(1));
        yh_dma(rs, yhu, v0, so(0)); yh_dma(rs, yhu + kYHsz, v0, so(1));
        yr_load(r0, rs, vr, so(0)); yr_load(r1, rs, vr, so(1));
        yr_store(r0, ymw); yr_store(r1, ymw + kYMsz);
        dma_landed();
        step_barrier();                                   // B0
        helper_partials<5>(hp_hi, c.lds, lane, 1, q, yh, pbw);                             // steps 0 and 1
        __builtin_amdgcn_sched_barrier(0);
        helper_partials<5>(hp_hi, c.lds, lane, 1, q, yh + kYHsz, pbw + kPBsz);
        __builtin_amdgcn_sched_barrier(0);
        step_barrier();                                   // B1: every helper has read H[0]
        yh_dma(rs, yhu, v0, so(2));                       // H[0] <- step 2
        dma_landed();
        step_barrier();                                   // B2
#pragma unroll 1
        for (int s = 0; s < L; ++s) {
            const int p0 = s & 1, p1 = p0 ^ 1;
            stamp(c, grp == (int)blockIdx.x, s, 0);
            // step s + 2 -> M[p0] (the owners read M[p1] now), step s + 3's helper slabs -> H[p1] (the helpers read H[p0] now):
            // in flight behind this wave's helper share, landed before the barrier that hands the buffers over
            if (!(kL1fX & 16)) {
                ym_dma(rs, ymu + p0 * kYMsz, v0, so(s + 2));
                yh_dma(rs, yhu + p1 * kYHsz, v0, so(s + 3));
                yr_load(r0, rs, vr, so(s + 2));
            }
            stamp(c, grp == (int)blockIdx.x, s, 1);
            __builtin_amdgcn_sched_barrier(0);
            if (!(kL1fX & 32)) helper_partials<5>(hp_hi, c.lds, lane, 1, q, yh + p0 * kYHsz, pbw + p0 * kPBsz);   // unit tiles 1..5, step s + 2
            __builtin_amdgcn_sched_barrier(0);
            stamp(c, grp == (int)blockIdx.x, s, 2);
            yr_store(r0, ymw + p0 * kYMsz);
            dma_landed();
            stamp(c, grp == (int)blockIdx.x, s, 3);
            step_barrier();
        }
        step_barrier();
    }
    stamp_print(c, 4);
}

__global__ __launch_bounds__(512) void gru_l1f_kernel(GruL1fParams P) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int dir = blockIdx.y;
    const char* wdir = P.w + (size_t)dir * P.w_dir_stride;
    {
        const f32x4* src = reinterpret_cast<const f32x4*>(wdir + 6 * kUnitB + kRemB + kLo01B);
        for (int i = tid; i < (kLdsW + kBiasB) / 16; i += 512) reinterpret_cast<f32x4*>(smem)[i] = src[i];
    }
    __syncthreads();
    const float inv = *reinterpret_cast<const float*>(smem + kLdsW + 6 * 256 + 64);
    const float inv_head = *reinterpret_cast<const float*>(smem + kLdsW + 6 * 256 + 68);
    const Ctx c{P, wdir, (lds_cptr)smem, lane, lane & 15, lane >> 4, dir, P.L, inv, inv_head, wave};
    if (wave == 3) rem_wave(c);
    else if (wave == 7) stage_wave(c);
    else unit_wave(c, wave < 3 ? wave : wave - 1);
}

}  // namespace

int gru_l1f_lds_bytes() { return kLds + kDbgB; }

hipError_t launch_gru_l1f(const GruL1fParams& P, hipStream_t st) {
    static int ncu = 0;
    if (!ncu) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return hipGetLastError();
        ncu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gru_l1f_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kLds + kDbgB);
    if (e != hipSuccess) return e;
    // one workgroup per CU (157 KB of LDS), half of them per direction; each walks its share of the 16-block groups
    const int per_dir = std::max(1, ncu / 2);
    const dim3 grid((unsigned)std::min(P.ngroups, per_dir), 2);
    hipLaunchKernelGGL(gru_l1f_kernel, grid, dim3(512), kLds + kDbgB, st, P);
    return hipGetLastError();
}

}  // namespace tae
