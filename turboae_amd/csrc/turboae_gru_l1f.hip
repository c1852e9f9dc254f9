// Layer 1 of the DeepTurbo GRU stacks (DEC_LargeRNN, decoders.py:41-49,84-149; ENC_interRNN, encoders.py:251-298) as ONE kernel
// in the f16x2 representation: input projection W_ih1 * y0_t, recurrence W_hh * h_{t-1}, gate arithmetic and this direction's
// half of the Linear head - the projections GI (4 GB per stack at 16 384 blocks, written by gru_proj_h and read back by
// gru_rec_h<layer 1> through r04) never exist.
//
// The workgroup is split by hidden UNIT, not by block (gru_rec_h: one wave = 16 blocks x all 300 gate rows, A fragments from
// LDS, which is why W_ih1 could not join them): 8 waves work on ONE group of 16 blocks (N = 16 columns of every MFMA) of one
// direction,
//   * 6 "unit" waves: wave ut owns units 16 ut .. 16 ut + 15, i.e. the r, z, n row tiles of those units.  Register-resident for the
//     whole launch: W_hh hi + lo (3 tiles x (3 slabs x 8 + remainder 4) = 84 VGPRs) and W_ih1 hi (3 x (6 x 4 + 4) = 84); the W_ih1
//     lo fragments (108 KB for the 18 tiles) stay in LDS.  Per step: 33 MFMAs of recurrence (on the critical path) + 60 of
//     projection for the NEXT step (off it: they fill the pipe while the gates are computed and the other waves arrive);
//   * the remainder wave: units 96..99 as one mixed row tile (row 4 qq + i = r, z, n_h, n_i of unit 96 + qq; everything in
//     registers), plus the head tile: W_lin[:, dir H .. dir H + H) * h_t, 16 floats per position leave the kernel as before;
//   * the staging wave: y0 of step s + 2 (16 rows of 800 bytes) from HBM into LDS in B-fragment order (loaded one step ahead).
// h_t is exchanged through LDS as B fragments (hi | lo halves, 8 KB, double-buffered): one s_barrier per step.  Waves w and w + 4
// share a SIMD, so the two light waves are 3 and 7 and every other SIMD carries two unit waves.
// LDS: W_ih1 lo 110 592 + bias rows 1 616 + h 2 x 8 192 + y0 2 x 14 336 = 157 264 of 163 840 bytes.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <type_traits>
#include "turboae_internal.hpp"
#include "turboae_device.hpp"

namespace tae {

namespace {

using u32x4v = __attribute__((ext_vector_type(4))) uint32_t;
using lds_q4 = const u32x4v __attribute__((address_space(3)));
using lds_f4c = const f32x4 __attribute__((address_space(3)));
using lds_w4 = u32x4v __attribute__((address_space(3)));
using lds_w2 = u32x2v __attribute__((address_space(3)));
using lds_ptr = char __attribute__((address_space(3)))*;

constexpr int kLdsW = 6 * 3 * 6 * 1024;             // W_ih1 lo fragments [ut][gate][slab][lane][8 halves]
constexpr int kBiasB = 6 * 4 * 64 + 64 + 16;        // [ut][r, z, n_i, n_h][16] + remainder-tile init row + (2^-S, 2^-S_head, 0, 0)
constexpr int kHB = kLdsW + kBiasB;                 // h exchange: 3 slabs x (hi | lo) + remainder (b1 | b2)
constexpr int kHBsz = 8192;
constexpr int kYB = kHB + 2 * kHBsz;                // y0 of a step: 6 slabs x (hi | lo) + remainder (b1 | b2)
constexpr int kYBsz = 14336;
constexpr int kLds = kYB + 2 * kYBsz;
constexpr int kUnitB = 42 * 1024, kRemB = 27 * 1024;
static_assert(kLds <= 160 * 1024, "LDS budget");
static_assert(kUnitB == GruL1fLayout::kUnitB && kRemB == GruL1fLayout::kRemB && kLdsW + kBiasB == GruL1fLayout::kLdsImgB, "host packing");

__device__ __forceinline__ float sigm_f(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896341f * x));
}
__device__ __forceinline__ float tanh_f(float x) {
    return fmaf(-2.0f, __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.88539008177792681f * x)), 1.0f);
}
__device__ __forceinline__ h8 lds_h8(lds_cptr p) { return __builtin_bit_cast(h8, *reinterpret_cast<lds_q4*>(p)); }
__device__ __forceinline__ h8 glb_h8(const char* p) { return __builtin_bit_cast(h8, *reinterpret_cast<const u32x4v*>(p)); }

// LDS writes of this wave are done and visible, then the workgroup barrier.  NOT __syncthreads(): that also drains vmcnt, and the
// staging wave's loads / the head stores are meant to stay in flight across steps.
__device__ __forceinline__ void step_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
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
// K = 16 remainder slab: A = [4 hi | 4 lo] of the lane's k = 0..3, b1 = [lo | hi], b2 = [hi | 0] (gru_rec_h's mma_rem)
__device__ __forceinline__ void mma3r(f32x4 (&acc)[3], const h8 (&ar)[3], h8 b1, h8 b2) {
#pragma unroll
    for (int g = 0; g < 3; ++g) acc[g] = mfma16x16x32h(ar[g], b1, acc[g]);
#pragma unroll
    for (int g = 0; g < 3; ++g) acc[g] = mfma16x16x32h(ar[g], b2, acc[g]);
}
__device__ __forceinline__ void mma1(f32x4& acc, h8 ah, h8 al, h8 bh, h8 bl) {
    acc = mfma16x16x32h(ah, bl, acc);
    acc = mfma16x16x32h(al, bh, acc);
    acc = mfma16x16x32h(ah, bh, acc);
}
__device__ __forceinline__ void mma1r(f32x4& acc, h8 ar, h8 b1, h8 b2) {
    acc = mfma16x16x32h(ar, b1, acc);
    acc = mfma16x16x32h(ar, b2, acc);
}

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
};

// ---- unit wave ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void unit_wave(const Ctx& c, int ut) {
    const int lane = c.lane, L = c.L;
    const char* wr = c.wdir + (size_t)ut * kUnitB + lane * 16;
    h8 hh_hi[3][3], hh_lo[3][3], hh_r[3], ih_hi[6][3], ih_r[3];       // [slab][gate]
#pragma unroll
    for (int g = 0; g < 3; ++g) {
#pragma unroll
        for (int sl = 0; sl < 3; ++sl) {
            hh_hi[sl][g] = glb_h8(wr + (g * 7 + 2 * sl) * 1024);
            hh_lo[sl][g] = glb_h8(wr + (g * 7 + 2 * sl + 1) * 1024);
        }
        hh_r[g] = glb_h8(wr + (g * 7 + 6) * 1024);
#pragma unroll
        for (int sl = 0; sl < 6; ++sl) ih_hi[sl][g] = glb_h8(wr + (21 + g * 7 + sl) * 1024);
        ih_r[g] = glb_h8(wr + (21 + g * 7 + 6) * 1024);
    }
    const lds_cptr wl = c.lds + ut * (18 * 1024) + lane * 16;           // W_ih1 lo: [gate][slab] of this unit tile
    const lds_cptr bias = c.lds + kLdsW + ut * 256 + c.q * 16;          // rows r, z, n_i, n_h
    const lds_cptr hb = c.lds + kHB + lane * 16, yb = c.lds + kYB + lane * 16;
    const lds_ptr hw = (lds_ptr)(c.lds + kHB + (ut >> 1) * 2048 + lane * 16 + (ut & 1) * 8);
    const float inv = c.inv;

    // gi = b + W_ih1 * y0 (r, z, n_i rows of this unit tile) from the staged fragments at `y`
    auto proj = [&](f32x4 (&gi)[3], lds_cptr y) {
#pragma unroll
        for (int g = 0; g < 3; ++g) gi[g] = *reinterpret_cast<lds_f4c*>(bias + g * 64);
#pragma unroll
        for (int sl = 0; sl < 6; ++sl) {
            const h8 bh = lds_h8(y + sl * 2048), bl = lds_h8(y + sl * 2048 + 1024);
            h8 al[3];
#pragma unroll
            for (int g = 0; g < 3; ++g) al[g] = lds_h8(wl + (g * 6 + sl) * 1024);
            mma3(gi, ih_hi[sl], al, bh, bl);
        }
        const h8 b1 = lds_h8(y + 12288), b2 = lds_h8(y + 13312);
        mma3r(gi, ih_r, b1, b2);
    };

    for (int grp = blockIdx.x; grp < c.P.ngroups; grp += gridDim.x) {
        // h_{-1} = 0: this wave's slots of buffer 0
        *reinterpret_cast<lds_w2*>(hw) = u32x2v{0, 0};
        *reinterpret_cast<lds_w2*>(hw + 1024) = u32x2v{0, 0};
        step_barrier();                                   // B0: y0 of steps 0 and 1 staged, h buffer 0 cleared
        f32x4 gi[3];
        proj(gi, yb);
        f32x4 h = {0.f, 0.f, 0.f, 0.f};
        step_barrier();                                   // B1: y0 buffer 0 may be refilled
#pragma unroll 1
        for (int s = 0; s < L; ++s) {
            // One step = ONE scheduling region, issue order pinned (left alone the scheduler puts every LDS read right in front of
            // its first use - 11 exposed LDS latencies per step - and the gate arithmetic behind the last MFMA):
            //   recurrence (the critical path): B fragments of h_{s-1} one slab ahead of their MFMAs;
            //   projection of step s + 1 (off it): operands one slab ahead, and the gate arithmetic of step s dealt out between its
            //   MFMAs, two vector-ALU instructions per MFMA - the matrix pipe never waits for the gates.
            const lds_cptr hc = hb + (s & 1) * kHBsz, y = yb + ((s + 1) & 1) * kYBsz;
            f32x4 acc[3] = {gi[0], gi[1], *reinterpret_cast<lds_f4c*>(bias + 3 * 64)};
            const f32x4 gin = gi[2];
            h8 xh[2], xl[2];                                   // B-fragment ring
            xh[0] = lds_h8(hc); xl[0] = lds_h8(hc + 1024);
            xh[1] = lds_h8(hc + 2048); xl[1] = lds_h8(hc + 3072);
            mma3(acc, hh_hi[0], hh_lo[0], xh[0], xl[0]);
            pin<5, 9, 0>();
            xh[0] = lds_h8(hc + 4096); xl[0] = lds_h8(hc + 5120);
            mma3(acc, hh_hi[1], hh_lo[1], xh[1], xl[1]);
            pin<2, 9, 0>();
            xh[1] = lds_h8(hc + 6144); xl[1] = lds_h8(hc + 7168);       // remainder slab: b1, b2
            mma3(acc, hh_hi[2], hh_lo[2], xh[0], xl[0]);
            pin<2, 9, 0>();
            xh[0] = lds_h8(y); xl[0] = lds_h8(y + 1024);
#pragma unroll
            for (int g = 0; g < 3; ++g) gi[g] = *reinterpret_cast<lds_f4c*>(bias + g * 64);
            mma3r(acc, hh_r, xh[1], xl[1]);
            pin<5, 6, 0>();
            f32x4 hn;
            h4 nhi, nlo;
            static_for<6>([&](auto SL) {
                constexpr int sl = decltype(SL)::value, cur = sl & 1, nxt = cur ^ 1;
                // W_ih1 lo fragments of THIS slab (used by its last three MFMAs), B fragments of the NEXT one
                h8 al[3];
#pragma unroll
                for (int g = 0; g < 3; ++g) al[g] = lds_h8(wl + (g * 6 + sl) * 1024);
                if (sl < 5) { xh[nxt] = lds_h8(y + (sl + 1) * 2048); xl[nxt] = lds_h8(y + (sl + 1) * 2048 + 1024); }
                else { xh[nxt] = lds_h8(y + 12288); xl[nxt] = lds_h8(y + 13312); }
                mma3_lo_last(gi, ih_hi[sl], al, xh[cur], xl[cur]);
                if (sl < 4) {
                    const int i = sl;
                    const float r = sigm_f(acc[0][i] * inv);
                    const float z = sigm_f(acc[1][i] * inv);
                    const float nn = tanh_f(fmaf(r, acc[2][i] * inv, gin[i] * inv));
                    hn[i] = fmaf(z, h[i] - nn, nn);
                } else if (sl == 4) {
                    h = hn;
                    split4(hn, nhi, nlo);
                    const lds_ptr hn_w = hw + ((s + 1) & 1) * kHBsz;
                    *reinterpret_cast<lds_w2*>(hn_w) = __builtin_bit_cast(u32x2v, nhi);
                    *reinterpret_cast<lds_w2*>(hn_w + 1024) = __builtin_bit_cast(u32x2v, nlo);
                }
                pin<5, 9, (sl < 4 ? 2 : (sl == 4 ? 1 : 0))>();
            });
            mma3r(gi, ih_r, xh[0], xl[0]);
            step_barrier();
        }
        step_barrier();                                   // the remainder wave's last head product has read the h buffer
    }
}

// ---- remainder wave: units 96..99 + the head tile ----------------------------------------------------------------------------
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
    const lds_cptr bias = c.lds + kLdsW + 6 * 256 + q * 16;             // init row: (r, z, b_hn, b_in) of unit 96 + q
    const lds_cptr hb = c.lds + kHB + lane * 16, yb = c.lds + kYB + lane * 16;
    const lds_ptr hw = (lds_ptr)(c.lds + kHB + 6144 + lane * 16);
    const float inv = c.inv, inv_head = c.inv_head;

    auto proj = [&](f32x4& gi, lds_cptr y) {
        gi = *reinterpret_cast<lds_f4c*>(bias);
#pragma unroll
        for (int sl = 0; sl < 6; ++sl) {
            const h8 bh = lds_h8(y + sl * 2048), bl = lds_h8(y + sl * 2048 + 1024);
            mma1(gi, ih_hi[sl], ih_lo[sl], bh, bl);
        }
        const h8 b1 = lds_h8(y + 12288), b2 = lds_h8(y + 13312);
        mma1r(gi, ih_r, b1, b2);
    };
    // head products of the state in h buffer `hc` -> hpart[(grp L + t) 16 + n][dir][8]
    auto head = [&](lds_cptr hc, const __amdgpu_buffer_rsrc_t& rs, int t) {
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int sl = 0; sl < 3; ++sl) {
            const h8 bh = lds_h8(hc + sl * 2048), bl = lds_h8(hc + sl * 2048 + 1024);
            mma1(a, hd_hi[sl], hd_lo[sl], bh, bl);
        }
        const h8 b1 = lds_h8(hc + 6144), b2 = lds_h8(hc + 7168);
        mma1r(a, hd_r, b1, b2);
        const uint32_t v = q < 2 ? (uint32_t)(n * 64 + dir * 32 + q * 16) : 0x80000000u;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v, a * inv_head), rs, v, (uint32_t)t * 1024u, 0);
    };

    for (int grp = blockIdx.x; grp < c.P.ngroups; grp += gridDim.x) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(c.P.hpart + (size_t)grp * L * 256, 0, L * 1024, 0x00020000);
        *reinterpret_cast<lds_w4*>(hw) = u32x4v{0, 0, 0, 0};
        *reinterpret_cast<lds_w4*>(hw + 1024) = u32x4v{0, 0, 0, 0};
        step_barrier();                                   // B0
        f32x4 gi;
        proj(gi, yb);
        float hr = 0.0f;
        step_barrier();                                   // B1
#pragma unroll 1
        for (int s = 0; s < L; ++s) {
            const lds_cptr hc = hb + (s & 1) * kHBsz;
            f32x4 acc = gi;
#pragma unroll
            for (int sl = 0; sl < 3; ++sl) {
                const h8 bh = lds_h8(hc + sl * 2048), bl = lds_h8(hc + sl * 2048 + 1024);
                mma1(acc, hh_hi[sl], hh_lo[sl], bh, bl);
            }
            {
                const h8 b1 = lds_h8(hc + 6144), b2 = lds_h8(hc + 7168);
                mma1r(acc, hh_r, b1, b2);
            }
            proj(gi, yb + ((s + 1) & 1) * kYBsz);
            if (s > 0) head(hc, rs, dir ? L - s : s - 1);          // Linear head on h_{s-1} (the state this step started from)
            const float r = sigm_f(acc[0] * inv);
            const float z = sigm_f(acc[1] * inv);
            const float nn = tanh_f(fmaf(r, acc[2] * inv, acc[3] * inv));
            hr = fmaf(z, hr - nn, nn);
            const _Float16 hi = (_Float16)hr;
            const _Float16 lo = (_Float16)(hr - (float)hi);
            const h8 b1 = {lo, 0, 0, 0, hi, 0, 0, 0}, b2 = {hi, 0, 0, 0, 0, 0, 0, 0};
            const lds_ptr hn_w = hw + ((s + 1) & 1) * kHBsz;
            *reinterpret_cast<lds_w4*>(hn_w) = __builtin_bit_cast(u32x4v, b1);
            *reinterpret_cast<lds_w4*>(hn_w + 1024) = __builtin_bit_cast(u32x4v, b2);
            step_barrier();
        }
        head(hb + (L & 1) * kHBsz, rs, dir ? 0 : L - 1);
        step_barrier();
    }
}

// ---- staging wave: y0 rows of one step -> B fragments in LDS ------------------------------------------------------------------
struct YRow { u32x4v v[12]; u32x2v rh, rl; };

__device__ __forceinline__ void y_load(YRow& r, __amdgpu_buffer_rsrc_t rs, uint32_t v0, uint32_t vr, uint32_t so) {
#pragma unroll
    for (int sl = 0; sl < 6; ++sl) {
        r.v[2 * sl] = __builtin_amdgcn_raw_buffer_load_b128(rs, v0 + sl * 64, so, 0);
        r.v[2 * sl + 1] = __builtin_amdgcn_raw_buffer_load_b128(rs, v0 + 400 + sl * 64, so, 0);
    }
    r.rh = __builtin_amdgcn_raw_buffer_load_b64(rs, vr, so, 0);               // out-of-range lanes read zeros
    r.rl = __builtin_amdgcn_raw_buffer_load_b64(rs, vr + 400, so, 0);
}
__device__ __forceinline__ void y_store(const YRow& r, lds_ptr y) {
#pragma unroll
    for (int i = 0; i < 12; ++i) *reinterpret_cast<lds_w4*>(y + i * 1024) = r.v[i];
    *reinterpret_cast<lds_w4*>(y + 12288) = u32x4v{r.rl.x, r.rl.y, r.rh.x, r.rh.y};       // b1 = [lo | hi]
    *reinterpret_cast<lds_w4*>(y + 13312) = u32x4v{r.rh.x, r.rh.y, 0, 0};                 // b2 = [hi | 0]
}

__device__ __forceinline__ void stage_wave(const Ctx& c) {
    const int lane = c.lane, L = c.L, n = c.n, q = c.q, dir = c.dir;
    const lds_ptr yw = (lds_ptr)(c.lds + kYB + lane * 16);
    // lane (n, kq) supplies k = 32 sl + 8 kq .. + 7 of block n: bytes plane * 400 + sl * 64 + kq * 16 of the block's row
    const uint32_t v0 = (uint32_t)(n * 800 + q * 16);
    const uint32_t vr = q < 2 ? (uint32_t)(n * 800 + 384 + q * 8) : 0x80000000u;        // remainder k = 192 + 4 kq .. + 3 (kq < 2)
    for (int grp = blockIdx.x; grp < c.P.ngroups; grp += gridDim.x) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(c.P.y0) + (size_t)grp * L * (16 * 800), 0, L * 16 * 800, 0x00020000);
        const auto so = [&](int s) { return (uint32_t)__builtin_amdgcn_readfirstlane(dir ? L - 1 - s : s) * (16 * 800u); };
        YRow r0, r1;
        y_load(r0, rs, v0, vr, so(0));
        y_load(r1, rs, v0, vr, so(L > 1 ? 1 : 0));
        y_store(r0, yw);
        y_store(r1, yw + kYBsz);
        if (L > 2) y_load(r0, rs, v0, vr, so(2));
        step_barrier();                                   // B0
        step_barrier();                                   // B1
#pragma unroll 1
        for (int s = 0; s < L; ++s) {
            // r0 holds y0 of step s + 2 (fetched a step ago): into buffer s & 1, whose last reader (the projection of step s) passed
            // the previous barrier; then fetch step s + 3
            if (s + 2 < L) y_store(r0, yw + (s & 1) * kYBsz);
            if (s + 3 < L) y_load(r0, rs, v0, vr, so(s + 3));
            step_barrier();
        }
        step_barrier();
    }
}

__global__ __launch_bounds__(512) void gru_l1f_kernel(GruL1fParams P) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int dir = blockIdx.y;
    const char* wdir = P.w + (size_t)dir * P.w_dir_stride;
    {
        const f32x4* src = reinterpret_cast<const f32x4*>(wdir + 6 * kUnitB + kRemB);
        for (int i = tid; i < (kLdsW + kBiasB) / 16; i += 512) reinterpret_cast<f32x4*>(smem)[i] = src[i];
    }
    __syncthreads();
    const float inv = *reinterpret_cast<const float*>(smem + kLdsW + 6 * 256 + 64);
    const float inv_head = *reinterpret_cast<const float*>(smem + kLdsW + 6 * 256 + 68);
    const Ctx c{P, wdir, (lds_cptr)smem, lane, lane & 15, lane >> 4, dir, P.L, inv, inv_head};
    if (wave == 3) rem_wave(c);
    else if (wave == 7) stage_wave(c);
    else unit_wave(c, wave < 3 ? wave : wave - 1);
}

}  // namespace

int gru_l1f_lds_bytes() { return kLds; }

hipError_t launch_gru_l1f(const GruL1fParams& P, hipStream_t st) {
    static int ncu = 0;
    if (!ncu) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return hipGetLastError();
        ncu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gru_l1f_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
    if (e != hipSuccess) return e;
    // one workgroup per CU (157 KB of LDS), half of them per direction; each walks its share of the 16-block groups
    const int per_dir = std::max(1, ncu / 2);
    const dim3 grid((unsigned)std::min(P.ngroups, per_dir), 2);
    hipLaunchKernelGGL(gru_l1f_kernel, grid, dim3(512), kLds, st, P);
    return hipGetLastError();
}

}  // namespace tae
