// DeepTurbo GRU decoder (DEC_LargeRNN, decoders.py:84-149) in the f16x2 representation: the kernels of
// turboae_gru.hip with every fp32 operand of the two big contractions (W_hh * h_t in the recurrence,
// W_ih1 * y0 in the layer-1 input projection) carried as fp16 hi + lo halves and three f16 MFMA products
// per 32 k (see turboae_device.hpp, "fp16-split contraction").  Gate arithmetic, state update, biases and
// the Linear head stay fp32.
//
// gru_rec_h: as gru_rec (one wave = 16 blocks x one direction, h_t never leaves the registers, W_hh
// fragments resident in LDS, no barriers), with
//   * k-slab s (32 k) of the recurrent product = unit tiles 2s and 2s+1: lane (n, kq) supplies the units
//     16(2s) + 4kq + j (j < 4) and 16(2s+1) + 4kq + (j-4) - exactly the 8 state values that lane holds in the
//     D layout of those two unit tiles, re-split into halves after every step;
//   * the 4 remainder units (96..99) as one K = 16 slab (two v_mfma_f32_16x16x32_f16 carrying hi and lo halves side by side, see mma_rem): lane kq supplies unit
//     96 + kq in k = 4kq, and - layer 0 - its three spare k slots carry the stack inputs x_t[3kq .. 3kq+2], so
//     the K = 7 input projection costs no extra MFMAs for the r / z rows (the n-gate input part needs its own
//     accumulators: 6 extra K = 16 tiles);
//   * unit-tile-major order: the r, z, n tiles of one unit tile (+ remainder slab) are finished together and
//     their gate arithmetic overlaps the next unit tile's MFMAs.
// gru_proj_h: GI = W_ih1 * Y0 + b on conv_accumulate_h (K = 200 -> 7 slabs), 160 positions per workgroup
// staged in LDS as hi / lo planes; the layer-0 recurrence writes Y0 directly as halves
// [hi 200 | lo 200] per (position, block) - since late r06 in the three-region layout of turboae_y0.hpp - so staging is a copy.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <algorithm>
#include <type_traits>
#include "turboae_internal.hpp"
#include "turboae_device.hpp"
#include "turboae_y0.hpp"

namespace tae {

using u32x4v = __attribute__((ext_vector_type(4))) uint32_t;
using h2v = __attribute__((ext_vector_type(2))) _Float16;

constexpr int kGH = 100;
constexpr int kGXW = 8;
constexpr int kRecTileB = 3 * 2048 + 1024;                 // bytes of A fragments per gate-row tile: 3 slabs (hi | lo) + K=16 remainder ([lane][4 hi | 4 lo])
constexpr int kRecFragB = 19 * kRecTileB;                   // 136 192
constexpr int kNiFragB = 6 * 1024;                          // layer 0: n-gate input tiles (K = 16, [lane][hi | lo])
constexpr int kRec0B = kRecFragB + kNiFragB + 25 * 64 + 16; // + accumulator-init rows + 2^-S
constexpr int kRec1B = kRecFragB + kRecTileB + 7 * 64 + 16; // + Linear-head tile (this direction's half of the head weights) + b_hn rows + (2^-S, 2^-S_head)

// The K = 16 remainder products run as v_mfma_f32_16x16x32_f16 (lane (i, kq) contracts its k = 0..3 in the instruction's k slots
// 8kq .. 8kq+3; see mma_rem for what the slots 8kq+4 .. 8kq+7 carry), NOT as v_mfma_f32_16x16x16_f16: mixed in one stream with
// 16x16x32, the K = 16 instruction read stale accumulators on gfx950 (a dependent 16x16x32 -> 16x16x16 pair through srcC, and - once
// the scheduler moved the K = 16 products between other chains - whole tiles; the compiler inserts no wait states for these pairs).
// Found in r03 when the head tile was added: results changed from run to run with two waves per SIMD.  One MFMA shape per kernel.
__device__ __forceinline__ float sigm_h(float x) {
    if (TAE_REC_X & 1) return fmaf(x, 0.25f, 0.5f);
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896341f * x));
}
// sigmoid / tanh of x 2^-S with the accumulators' power-of-two scale inside the exp2 constant (cs = -log2(e) 2^-S, ct = 2 log2(e) 2^-S):
// bit for bit what scaling x first gives (a power of two commutes with every rounding involved), two to three instructions fewer per gate
__device__ __forceinline__ float sigm_hs(float x, float cs) {
    if (TAE_REC_X & 1) return fmaf(x * cs, -0.17f, 0.5f);
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * cs));
}
__device__ __forceinline__ float tanh_hs(float x, float ct) {
    if (TAE_REC_X & 1) return x * ct * 0.17f;
    return fmaf(-2.0f, __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * ct)), 1.0f);
}
__device__ __forceinline__ float tanh_h(float x) {
    if (TAE_REC_X & 1) return x * 0.5f;
    return fmaf(-2.0f, __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.88539008177792681f * x)), 1.0f);
}

using lds_q4 = const u32x4v __attribute__((address_space(3)));
using lds_q2 = const u32x2v __attribute__((address_space(3)));
using lds_f4c = const f32x4 __attribute__((address_space(3)));

// A fragments of one k-slab for NG gate tiles (tile stride STRIDE bytes): 32-k slabs as 8 halves per lane, the K = 16
// remainder slab as [4 hi | 4 lo] halves per lane; slabs: hi then lo, 1024 bytes apart.  Loads and MFMAs are separate calls so the
// caller can issue the next slab's LDS reads before the current slab's MFMAs (pinned with sched_group_barrier: left to
// the scheduler every read ends up right in front of its first use and the LDS latency is exposed 28 times per step).
template <int NG> struct FragS { h8 ah[NG], al[NG]; };
template <int NG> struct FragR { h8 a[NG]; };                 // remainder slab: [4 hi halves | 4 lo halves] of the lane's k = 0..3, one 16-byte read

template <int NG>
__device__ __forceinline__ void load_slab(FragS<NG>& f, lds_cptr fr) {
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        if (TAE_REC_X & 2) { asm volatile("" : "+v"(f.ah[g]), "+v"(f.al[g])); continue; }
        f.ah[g] = __builtin_bit_cast(h8, *reinterpret_cast<lds_q4*>(fr + g * kRecTileB));
        f.al[g] = __builtin_bit_cast(h8, *reinterpret_cast<lds_q4*>(fr + g * kRecTileB + 1024));
    }
}
template <int NG>
__device__ __forceinline__ void mma_slab(f32x4 (&acc)[NG], const FragS<NG>& f, h8 bh, h8 bl) {
    if (TAE_REC_X & 8) {
#pragma unroll
        for (int g = 0; g < NG; ++g) asm volatile("" : "+v"(acc[g]) : "v"(f.ah[g]), "v"(f.al[g]), "v"(bh), "v"(bl));
        return;
    }
#pragma unroll
    for (int g = 0; g < NG; ++g) acc[g] = mfma16x16x32h(f.ah[g], bl, acc[g]);
#pragma unroll
    for (int g = 0; g < NG; ++g) acc[g] = mfma16x16x32h(f.al[g], bh, acc[g]);
#pragma unroll
    for (int g = 0; g < NG; ++g) acc[g] = mfma16x16x32h(f.ah[g], bh, acc[g]);
}
template <int NG, int STRIDE>
__device__ __forceinline__ void load_rem(FragR<NG>& f, lds_cptr fr) {
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        if (TAE_REC_X & 2) { asm volatile("" : "+v"(f.a[g])); continue; }
        f.a[g] = __builtin_bit_cast(h8, *reinterpret_cast<lds_q4*>(fr + g * STRIDE));
    }
}
// The K = 16 remainder in TWO K = 32 products: A = [a.hi | a.lo] (k slots 0..3 | 4..7 of every lane), b1 = [b.lo | b.hi] gives
// a.hi * b.lo + a.lo * b.hi in one instruction, b2 = [b.hi | 0] adds a.hi * b.hi (its a.lo half meets zeros).  One 16-byte LDS
// read per tile and no zero-extension of the A operand (the earlier form - three products on zero-extended 8-byte fragments -
// cost four v_mov per MFMA, 211 per step and wave: the register tuples' zero halves were re-made for every use).
template <int NG>
__device__ __forceinline__ void mma_rem(f32x4 (&acc)[NG], const FragR<NG>& f, h8 b1, h8 b2) {
    if (TAE_REC_X & 8) {
#pragma unroll
        for (int g = 0; g < NG; ++g) asm volatile("" : "+v"(acc[g]) : "v"(f.a[g]), "v"(b1), "v"(b2));
        return;
    }
#pragma unroll
    for (int g = 0; g < NG; ++g) acc[g] = mfma16x16x32h(f.a[g], b1, acc[g]);
#pragma unroll
    for (int g = 0; g < NG; ++g) acc[g] = mfma16x16x32h(f.a[g], b2, acc[g]);
}
// {ND LDS reads, then NM MFMAs}: the reads (for a LATER slab) go first, the MFMAs of the current slab cover their latency
template <int ND, int NM>
__device__ __forceinline__ void pin_ds_mma() {
    __builtin_amdgcn_sched_group_barrier(0x100, ND, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, NM, 0);
}

template <bool LAYER0>
__global__ __launch_bounds__(512) void gru_rec_h_kernel(GruRecParams P) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, n = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int dir = blockIdx.y, L = P.L;
    {
        const f32x4* src = reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(P.w) + (size_t)dir * P.w_dir_stride);
        constexpr int NV = (LAYER0 ? kRec0B : kRec1B) / 16;
        for (int i = tid; i < NV; i += (int)blockDim.x) reinterpret_cast<f32x4*>(smem)[i] = src[i];
    }
    __syncthreads();
    const int b0 = (blockIdx.x * (int)(blockDim.x >> 6) + wave) * 16;
    if (b0 >= P.B) return;                                  // no barrier below: waves are independent
    const int nb = min(16, P.B - b0);
    const bool valid = n < nb;
    const int nc = valid ? n : nb - 1;
    const lds_cptr lds3 = (lds_cptr)smem;
    const lds_cptr bias = lds3 + kRecFragB + (LAYER0 ? kNiFragB : kRecTileB) + q * 16;
    const float inv = *reinterpret_cast<const float*>(smem + kRecFragB + (LAYER0 ? kNiFragB + 25 * 64 : kRecTileB + 7 * 64));
    const float cs = -1.44269504088896341f * inv, ct = 2.88539008177792681f * inv;
    const float inv_head = LAYER0 ? 0.0f : *reinterpret_cast<const float*>(smem + kRecFragB + kRecTileB + 7 * 64 + 4);

    const __amdgpu_buffer_rsrc_t rs_in = LAYER0
        ? __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(P.x + (size_t)b0 * L * kGXW), 0, nb * L * kGXW * 4, 0x00020000)
        : __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(P.gi), 0, -1, 0x00020000);
    const uint32_t v_in = LAYER0 ? (uint32_t)(nc * L * kGXW * 4) : (uint32_t)((n * 4 + q) * 16);
    const uint32_t gi_wave = (uint32_t)(b0 / 16) * (uint32_t)L * (2 * 19 * 1024u) + (uint32_t)dir * (19 * 1024u);
    // outputs: layer 0 -> Y0 as halves, logically [pos][hi 200 | lo 200] (the projection kernel's operand; HBM layout: turboae_y0.hpp); layer 1 -> this direction's
    // share of the Linear head, W_lin[:, dir * H .. dir * H + H) * h_t, as [pos][dir][8] floats (Y1 itself is never written:
    // the head contracts it away, 16 instead of 200 floats per position leave the kernel)
    const __amdgpu_buffer_rsrc_t rs_y = LAYER0
        ? __builtin_amdgcn_make_buffer_rsrc(P.y + (size_t)b0 * L * 2 * kGH, 0, 16 * L * 2 * kGH * 4, 0x00020000)
        : __builtin_amdgcn_make_buffer_rsrc(P.hpart + (size_t)b0 * L * 16, 0, 16 * L * 16 * 4, 0x00020000);
    const uint32_t v_y = !valid ? 0x80000000u : (q < 2 ? (uint32_t)(n * 64 + dir * 32 + q * 16) : 0x80000000u);      // layer 1: head rows
    // layer 0: Y0 in the three-region layout of turboae_y0.hpp (unit tile 0 has the one lane whose units live in region C)
    Y0UnitOff yo_u = y0_unit_offsets(dir, n, q);
    if (!valid) yo_u.hi0 = yo_u.lo0 = yo_u.hi1 = 0x80000000u;
    const uint32_t v_yr = !valid ? 0x80000000u : y0_half(n, dir * kGH + 96 + q);
    const uint32_t v_yr_lo = !valid ? 0x80000000u : y0_half_lo(n, dir * kGH + 96 + q);

    f32x4 h[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) h[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float hr = 0.0f;
    h8 bh[3], bl[3];                        // B operands of the three 32-k slabs = halves of h (unit tiles 2s, 2s+1)
#pragma unroll
    for (int s = 0; s < 3; ++s) { bh[s] = h8{0, 0, 0, 0, 0, 0, 0, 0}; bl[s] = bh[s]; }
    h4 rh = {0, 0, 0, 0}, rl = {0, 0, 0, 0};   // remainder slab: k0 = h of unit 96 + q, k1..3 = this lane's share of x_t (layer 0)
    h8 rb1, rb2;                               // ... as the two B operands of mma_rem: [lo | hi] and [hi | 0]
    auto set_rb = [&]() {
        rb1 = h8{rl[0], rl[1], rl[2], rl[3], rh[0], rh[1], rh[2], rh[3]};
        rb2 = h8{rh[0], rh[1], rh[2], rh[3], 0, 0, 0, 0};
    };

    // x_t (layer 0): lane group q carries x[3q .. 3q+2] (q = 2: x[6], x[7] - the panel is 8 wide; q = 3: nothing)
    auto load_x = [&](int t, f32x4& xa, f32x4& xb) {
        const uint32_t so = (uint32_t)t * kGXW * 4;
        xa = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_in, v_in, so, 0));
        xb = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_in, v_in + 16, so, 0));
    };
    auto set_x = [&](const f32x4& xa, const f32x4& xb) {
        const float x0 = q == 0 ? xa.x : (q == 1 ? xa.w : (q == 2 ? xb.z : 0.0f));
        const float x1 = q == 0 ? xa.y : (q == 1 ? xb.x : (q == 2 ? xb.w : 0.0f));
        const float x2 = q == 0 ? xa.z : (q == 1 ? xb.y : 0.0f);
        const float xs[3] = {x0, x1, x2};
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const _Float16 hi = (_Float16)xs[j];
            rh[j + 1] = hi;
            rl[j + 1] = (_Float16)(xs[j] - (float)hi);
        }
    };
    f32x4 xa = {0.f, 0.f, 0.f, 0.f}, xb = xa;
    if (LAYER0) {
        load_x(dir ? L - 1 : 0, xa, xb);
        set_x(xa, xb);
    }
    set_rb();
    FragS<3> fa, fb;       // A-fragment ping-pong of the unit tiles
    FragS<1> f1;           // ... of the remainder tile
    load_slab<3>(fa, lds3 + lane * 16);
#pragma unroll 1
    for (int s = 0; s < L; ++s) {
        const int t = dir ? L - 1 - s : s;
        f32x4 g[19];
        if (LAYER0) {
            const int tn = dir ? (t > 0 ? t - 1 : 0) : (t + 1 < L ? t + 1 : t);
            load_x(tn, xa, xb);
        } else {
            const uint32_t so = gi_wave + (uint32_t)t * (2 * 19 * 1024u);
#pragma unroll
            for (int T = 0; T < 19; ++T) {
                if (TAE_REC_X & 4) { g[T] = f32x4{0.f, 0.f, 0.f, 0.f}; asm volatile("" : "+v"(g[T])); continue; }
                g[T] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_in, v_in, so + T * 1024u, 0));
            }
        }
        __builtin_amdgcn_sched_barrier(0);     // keep this step's loads up here (they are consumed by the gate arithmetic / next step)
        h4 nhi[6], nlo[6];
        const uint32_t yo = (uint32_t)t * (16 * 800u);
#pragma unroll
        for (int u = 0; u < 6; ++u) {
            constexpr int NG = LAYER0 ? 4 : 3;
            f32x4 acc[NG];
            if (LAYER0) {
#pragma unroll
                for (int gI = 0; gI < 3; ++gI) acc[gI] = *reinterpret_cast<lds_f4c*>(bias + (3 * u + gI) * 64);
                acc[NG - 1] = *reinterpret_cast<lds_f4c*>(bias + (19 + u) * 64);
            } else {
                acc[0] = f32x4{0.f, 0.f, 0.f, 0.f};
                acc[1] = acc[0];
                acc[2] = *reinterpret_cast<lds_f4c*>(bias + u * 64);
            }
            f32x4 a3[3] = {acc[0], acc[1], acc[2]};
            // slabs 0..2 + remainder of this unit tile; fa arrives with slab 0 loaded (end of the previous unit tile / step)
            const lds_cptr fr = lds3 + (3 * u) * kRecTileB + lane * 16;
            const lds_cptr frn = lds3 + (3 * (u + 1)) * kRecTileB + lane * 16;      // next unit tile (u = 5: the remainder tile 18)
            FragR<3> fq;
            load_slab<3>(fb, fr + 2048);
            mma_slab<3>(a3, fa, bh[0], bl[0]);
            pin_ds_mma<6, 9>();
            load_slab<3>(fa, fr + 4096);
            mma_slab<3>(a3, fb, bh[1], bl[1]);
            pin_ds_mma<6, 9>();
            load_rem<3, kRecTileB>(fq, lds3 + (3 * u) * kRecTileB + 6144 + lane * 16);
            FragR<1> fn;
            if (LAYER0) load_rem<1, 1024>(fn, lds3 + kRecFragB + u * 1024 + lane * 16);
            mma_slab<3>(a3, fa, bh[2], bl[2]);
            pin_ds_mma<LAYER0 ? 4 : 3, 9>();
            if (u < 5) load_slab<3>(fa, frn);
            else load_slab<1>(f1, frn);
            mma_rem<3>(a3, fq, rb1, rb2);
            f32x4 ani = {0.f, 0.f, 0.f, 0.f};
            if (LAYER0) {
                f32x4 a1[1] = {acc[NG - 1]};
                mma_rem<1>(a1, fn, rb1, rb2);
                ani = a1[0];
            }
            pin_ds_mma<6, LAYER0 ? 8 : 6>();
            f32x4 hn;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float r = LAYER0 ? sigm_hs(a3[0][i], cs) : sigm_h(fmaf(a3[0][i], inv, g[3 * u][i]));
                const float z = LAYER0 ? sigm_hs(a3[1][i], cs) : sigm_h(fmaf(a3[1][i], inv, g[3 * u + 1][i]));
                const float nn = LAYER0 ? tanh_hs(fmaf(r, a3[2][i], ani[i]), ct) : tanh_h(fmaf(r, a3[2][i] * inv, g[3 * u + 2][i]));
                hn[i] = fmaf(z, h[u][i] - nn, nn);
            }
            h[u] = hn;
            split4(hn, nhi[u], nlo[u]);
            if (LAYER0 && !(TAE_REC_X & 4)) {
                const uint32_t vh = u == 0 ? yo_u.hi0 : yo_u.hi1 + (u - 1) * 32, vl = u == 0 ? yo_u.lo0 : yo_u.hi1 + (u - 1) * 32 + kY0PlaneAB;
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2v, nhi[u]), rs_y, vh, yo, 0);
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2v, nlo[u]), rs_y, vl, yo, 0);
            }
        }
        // remainder tile: rows 4qq + i = gate i of unit 96 + qq (i = 3: layer-0 n-gate input part)
        {
            f32x4 a1[1];
            a1[0] = *reinterpret_cast<lds_f4c*>(bias + (LAYER0 ? 18 : 6) * 64);
            const lds_cptr fr = lds3 + 18 * kRecTileB + lane * 16;
            FragS<1> f2;
            FragR<1> fq;
            load_slab<1>(f2, fr + 2048);
            mma_slab<1>(a1, f1, bh[0], bl[0]);
            load_slab<1>(f1, fr + 4096);
            mma_slab<1>(a1, f2, bh[1], bl[1]);
            load_rem<1, kRecTileB>(fq, lds3 + 18 * kRecTileB + 6144 + lane * 16);
            mma_slab<1>(a1, f1, bh[2], bl[2]);
            load_slab<3>(fa, lds3 + lane * 16);          // slab 0 of unit tile 0 for the next step
            mma_rem<1>(a1, fq, rb1, rb2);
            const f32x4 a = a1[0];
            const float r = LAYER0 ? sigm_hs(a[0], cs) : sigm_h(fmaf(a[0], inv, g[18][0]));
            const float z = LAYER0 ? sigm_hs(a[1], cs) : sigm_h(fmaf(a[1], inv, g[18][1]));
            const float nn = LAYER0 ? tanh_hs(fmaf(r, a[2], a[3]), ct) : tanh_h(fmaf(r, a[2] * inv, g[18][2]));
            hr = fmaf(z, hr - nn, nn);
        }
        // next step's B operands
#pragma unroll
        for (int sl = 0; sl < 3; ++sl) {
            bh[sl] = h8{nhi[2 * sl][0], nhi[2 * sl][1], nhi[2 * sl][2], nhi[2 * sl][3], nhi[2 * sl + 1][0], nhi[2 * sl + 1][1], nhi[2 * sl + 1][2], nhi[2 * sl + 1][3]};
            bl[sl] = h8{nlo[2 * sl][0], nlo[2 * sl][1], nlo[2 * sl][2], nlo[2 * sl][3], nlo[2 * sl + 1][0], nlo[2 * sl + 1][1], nlo[2 * sl + 1][2], nlo[2 * sl + 1][3]};
        }
        {
            const _Float16 hi = (_Float16)hr;
            const _Float16 lo = (_Float16)(hr - (float)hi);
            rh[0] = hi;
            rl[0] = lo;
            if (LAYER0) {
                __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(uint16_t, hi), rs_y, v_yr, yo, 0);
                __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(uint16_t, lo), rs_y, v_yr_lo, yo, 0);
                set_x(xa, xb);
            }
            set_rb();
        }
        if (!LAYER0) {
            // Linear head on the new state (decoders.py:103,115,143; encoders.py:284-292): tile 19 = the <= 8 output rows of this
            // direction's half of the head weights, B operand = the halves of h_t just formed for the next step - 12 MFMAs off the
            // dependent chain (the next step's gate products do not wait for them)
            f32x4 ah[1] = {f32x4{0.f, 0.f, 0.f, 0.f}};
            const lds_cptr fr = lds3 + 19 * kRecTileB + lane * 16;
            FragS<1> fh0, fh1, fh2;
            FragR<1> fhq;
            load_slab<1>(fh0, fr);
            load_slab<1>(fh1, fr + 2048);
            load_slab<1>(fh2, fr + 4096);
            load_rem<1, kRecTileB>(fhq, lds3 + 19 * kRecTileB + 6144 + lane * 16);
            mma_slab<1>(ah, fh0, bh[0], bl[0]);
            mma_slab<1>(ah, fh1, bh[1], bl[1]);
            mma_slab<1>(ah, fh2, bh[2], bl[2]);
            mma_rem<1>(ah, fhq, rb1, rb2);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v, ah[0] * inv_head), rs_y, v_y, (uint32_t)t * 1024u, 0);
        }
    }
}

// ---- layer 0, unit-split twin for small batches -------------------------------------------------------------------------------
// gru_rec_h_kernel<true> walks all 19 gate-row tiles of its 16 blocks in ONE wave: 4.7 us per step whatever the batch, so a batch of
// 500 blocks keeps 64 waves of the chip busy for 100 x 4.7 us per stack.  Here the same tiles of the same 16 blocks are dealt out
// to seven waves of one workgroup (unit tile u to wave u, the remainder tile + the input x_t to wave 6), h_t is exchanged through
// LDS once per step (the B-fragment layout gru_l1f_kernel uses), and every accumulator sees EXACTLY the products gru_rec_h_kernel
// gives it, in the same order, from the same packed fragments (read from the image in global memory into registers, once): the
// two kernels are bit-identical, so the host may pick by batch size without results depending on it (tests/test_gpu_parity.py).
constexpr int kU0HB = 8192;                                   // one exchange buffer: 3 slabs x (hi | lo) + remainder (b1 | b2)

__device__ __forceinline__ h8 u0_glb(const char* p) { return __builtin_bit_cast(h8, *reinterpret_cast<const u32x4v*>(p)); }
__device__ __forceinline__ h8 u0_lds(lds_cptr p) { return __builtin_bit_cast(h8, *reinterpret_cast<lds_q4*>(p)); }
__device__ __forceinline__ void u0_barrier() {               // LDS-only fence: the Y0 stores and x loads stay in flight across steps
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

__global__ __launch_bounds__(448, 4) void gru_rec0u_kernel(GruRecParams P) {
    __shared__ __attribute__((aligned(16))) char smem[2 * kU0HB + 25 * 64];       // h exchange (two buffers) | accumulator-init rows
    using lds_w4 = u32x4v __attribute__((address_space(3)));
    using lds_w2 = u32x2v __attribute__((address_space(3)));
    using lds_ptr = char __attribute__((address_space(3)))*;
    const int tid = threadIdx.x, lane = tid & 63, n = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int dir = blockIdx.y, L = P.L;
    const char* img = reinterpret_cast<const char*>(P.w) + (size_t)dir * P.w_dir_stride;
    const int b0 = blockIdx.x * 16;
    const int nb = min(16, P.B - b0);
    const bool valid = n < nb;
    const int nc = valid ? n : nb - 1;
    const float inv = *reinterpret_cast<const float*>(img + kRecFragB + kNiFragB + 25 * 64);
    const float cs = -1.44269504088896341f * inv, ct = 2.88539008177792681f * inv;
    const lds_cptr biasl = (lds_cptr)smem + 2 * kU0HB + q * 16;
    const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(P.y + (size_t)b0 * L * 2 * kGH, 0, 16 * L * 2 * kGH * 4, 0x00020000);
    for (int i = tid; i < kU0HB / 16; i += 448) reinterpret_cast<f32x4*>(smem)[i] = f32x4{0.f, 0.f, 0.f, 0.f};       // h_{-1} = 0
    if (tid < 25 * 4) reinterpret_cast<f32x4*>(smem + 2 * kU0HB)[tid] = reinterpret_cast<const f32x4*>(img + kRecFragB + kNiFragB)[tid];
    __syncthreads();
    const lds_cptr hb = (lds_cptr)smem + lane * 16;

    if (wave < 6) {
        // ---- unit tile u = wave: gate tiles 3u .. 3u + 2 and the n-gate input tile, as iteration u of gru_rec_h_kernel's loop
        const int u = wave;
        const char* fr = img + (size_t)(3 * u) * kRecTileB + lane * 16;
        FragS<3> f0, f1, f2;
        FragR<3> fq;
        FragR<1> fn;
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            f0.ah[g] = u0_glb(fr + g * kRecTileB);        f0.al[g] = u0_glb(fr + g * kRecTileB + 1024);
            f1.ah[g] = u0_glb(fr + g * kRecTileB + 2048); f1.al[g] = u0_glb(fr + g * kRecTileB + 3072);
            f2.ah[g] = u0_glb(fr + g * kRecTileB + 4096); f2.al[g] = u0_glb(fr + g * kRecTileB + 5120);
            fq.a[g] = u0_glb(fr + g * kRecTileB + 6144);
        }
        fn.a[0] = u0_glb(img + kRecFragB + u * 1024 + lane * 16);
        uint32_t v_y = 0x80000000u, v_ylo = 0x80000000u;              // Y0: turboae_y0.hpp
        if (valid) y0_unit_tile(dir, u, n, q, v_y, v_ylo);
        const lds_ptr hw = (lds_ptr)smem + (u >> 1) * 2048 + lane * 16 + (u & 1) * 8;
        f32x4 h = {0.f, 0.f, 0.f, 0.f};
        u0_barrier();                                         // B0: the remainder wave's b1 | b2 of step 0 are in buffer 0
#pragma unroll 1
        for (int s = 0; s < L; ++s) {
            const int t = dir ? L - 1 - s : s;
            const lds_cptr hc = hb + (s & 1) * kU0HB;
            const h8 bh0 = u0_lds(hc), bl0 = u0_lds(hc + 1024), bh1 = u0_lds(hc + 2048), bl1 = u0_lds(hc + 3072);
            const h8 bh2 = u0_lds(hc + 4096), bl2 = u0_lds(hc + 5120), rb1 = u0_lds(hc + 6144), rb2 = u0_lds(hc + 7168);
            f32x4 a3[3];
#pragma unroll
            for (int g = 0; g < 3; ++g) a3[g] = *reinterpret_cast<lds_f4c*>(biasl + (3 * u + g) * 64);
            mma_slab<3>(a3, f0, bh0, bl0);
            mma_slab<3>(a3, f1, bh1, bl1);
            mma_slab<3>(a3, f2, bh2, bl2);
            mma_rem<3>(a3, fq, rb1, rb2);
            f32x4 a1[1] = {*reinterpret_cast<lds_f4c*>(biasl + (19 + u) * 64)};
            mma_rem<1>(a1, fn, rb1, rb2);
            const f32x4 ani = a1[0];
            f32x4 hn;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float r = sigm_hs(a3[0][i], cs);
                const float z = sigm_hs(a3[1][i], cs);
                const float nn = tanh_hs(fmaf(r, a3[2][i], ani[i]), ct);
                hn[i] = fmaf(z, h[i] - nn, nn);
            }
            h = hn;
            h4 nhi, nlo;
            split4(hn, nhi, nlo);
            const uint32_t yo = (uint32_t)t * (16 * 800u);
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2v, nhi), rs_y, v_y, yo, 0);
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2v, nlo), rs_y, v_ylo, yo, 0);
            const lds_ptr hn_w = hw + ((s + 1) & 1) * kU0HB;
            *reinterpret_cast<lds_w2*>(hn_w) = __builtin_bit_cast(u32x2v, nhi);
            *reinterpret_cast<lds_w2*>(hn_w + 1024) = __builtin_bit_cast(u32x2v, nlo);
            u0_barrier();
        }
    } else {
        // ---- remainder tile 18 (rows 4 qq + i = r, z, n_h, n_i of unit 96 + qq) + x_t: builds the remainder slab's B operands
        const char* fr = img + (size_t)18 * kRecTileB + lane * 16;
        FragS<1> f0, f1, f2;
        FragR<1> fq;
        f0.ah[0] = u0_glb(fr);        f0.al[0] = u0_glb(fr + 1024);
        f1.ah[0] = u0_glb(fr + 2048); f1.al[0] = u0_glb(fr + 3072);
        f2.ah[0] = u0_glb(fr + 4096); f2.al[0] = u0_glb(fr + 5120);
        fq.a[0] = u0_glb(fr + 6144);
        const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(P.x + (size_t)b0 * L * kGXW), 0, nb * L * kGXW * 4, 0x00020000);
        const uint32_t v_in = (uint32_t)(nc * L * kGXW * 4);
        const uint32_t v_yr = !valid ? 0x80000000u : y0_half(n, dir * kGH + 96 + q);
        const uint32_t v_yr_lo = !valid ? 0x80000000u : y0_half_lo(n, dir * kGH + 96 + q);
        h4 rh = {0, 0, 0, 0}, rl = {0, 0, 0, 0};           // k0 = h of unit 96 + q, k1..3 = this lane group's share of x_t
        h8 rb1, rb2;
        auto set_rb = [&]() {
            rb1 = h8{rl[0], rl[1], rl[2], rl[3], rh[0], rh[1], rh[2], rh[3]};
            rb2 = h8{rh[0], rh[1], rh[2], rh[3], 0, 0, 0, 0};
        };
        auto load_x = [&](int t, f32x4& xa, f32x4& xb) {
            const uint32_t so = (uint32_t)t * kGXW * 4;
            xa = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_in, v_in, so, 0));
            xb = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_in, v_in + 16, so, 0));
        };
        auto set_x = [&](const f32x4& xa, const f32x4& xb) {
            const float x0 = q == 0 ? xa.x : (q == 1 ? xa.w : (q == 2 ? xb.z : 0.0f));
            const float x1 = q == 0 ? xa.y : (q == 1 ? xb.x : (q == 2 ? xb.w : 0.0f));
            const float x2 = q == 0 ? xa.z : (q == 1 ? xb.y : 0.0f);
            const float xs[3] = {x0, x1, x2};
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const _Float16 hi = (_Float16)xs[j];
                rh[j + 1] = hi;
                rl[j + 1] = (_Float16)(xs[j] - (float)hi);
            }
        };
        const lds_ptr rw = (lds_ptr)smem + 6144 + lane * 16;
        f32x4 xa, xb;
        load_x(dir ? L - 1 : 0, xa, xb);
        set_x(xa, xb);
        set_rb();
        *reinterpret_cast<lds_w4*>(rw) = __builtin_bit_cast(u32x4v, rb1);
        *reinterpret_cast<lds_w4*>(rw + 1024) = __builtin_bit_cast(u32x4v, rb2);
        float hr = 0.0f;
        u0_barrier();                                         // B0
#pragma unroll 1
        for (int s = 0; s < L; ++s) {
            const int t = dir ? L - 1 - s : s;
            const int tn = dir ? (t > 0 ? t - 1 : 0) : (t + 1 < L ? t + 1 : t);
            load_x(tn, xa, xb);
            const lds_cptr hc = hb + (s & 1) * kU0HB;
            const h8 bh0 = u0_lds(hc), bl0 = u0_lds(hc + 1024), bh1 = u0_lds(hc + 2048), bl1 = u0_lds(hc + 3072);
            const h8 bh2 = u0_lds(hc + 4096), bl2 = u0_lds(hc + 5120);
            f32x4 a1[1] = {*reinterpret_cast<lds_f4c*>(biasl + 18 * 64)};
            mma_slab<1>(a1, f0, bh0, bl0);
            mma_slab<1>(a1, f1, bh1, bl1);
            mma_slab<1>(a1, f2, bh2, bl2);
            mma_rem<1>(a1, fq, rb1, rb2);
            const f32x4 a = a1[0];
            const float r = sigm_hs(a[0], cs);
            const float z = sigm_hs(a[1], cs);
            const float nn = tanh_hs(fmaf(r, a[2], a[3]), ct);
            hr = fmaf(z, hr - nn, nn);
            const _Float16 hi = (_Float16)hr;
            const _Float16 lo = (_Float16)(hr - (float)hi);
            rh[0] = hi;
            rl[0] = lo;
            const uint32_t yo = (uint32_t)t * (16 * 800u);
            __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(uint16_t, hi), rs_y, v_yr, yo, 0);
            __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(uint16_t, lo), rs_y, v_yr_lo, yo, 0);
            set_x(xa, xb);
            set_rb();
            const lds_ptr rn = rw + ((s + 1) & 1) * kU0HB;
            *reinterpret_cast<lds_w4*>(rn) = __builtin_bit_cast(u32x4v, rb1);
            *reinterpret_cast<lds_w4*>(rn + 1024) = __builtin_bit_cast(u32x4v, rb2);
            u0_barrier();
        }
    }
}

hipError_t launch_gru_rec0u(const GruRecParams& P, hipStream_t st) {
    hipLaunchKernelGGL(gru_rec0u_kernel, dim3((unsigned)((P.B + 15) / 16), 2), dim3(448), 0, st, P);
    return hipGetLastError();
}

// ---- layer-1 input projections, f16x2 GEMM ---------------------------------------------------------------
// Workgroup = 8 waves, 160 positions staged in LDS (hi plane | lo plane, rows of 200 halves); wave (pg, rq):
// position group pg (5 tiles), row quarter rq = (direction, half): gate tiles [0, 10) or [10, 19) of that
// direction in two passes of <= 5 tiles (accumulators 5 x 5 tiles).
// Workgroup geometry: NPG position groups of 5 tiles (80 positions each) x 4 row quarters = 4 * NPG waves.  NPG = 1 (4 waves, 80
// positions, 65 KB of LDS): two workgroups share a CU, one stages its Y0 rows while the other streams MFMAs; NPG = 2 (8 waves, 160
// positions): one workgroup per CU (the r02 geometry; TAE_GRU_PROJ_PG=2).  The A-fragment stream per position is the same.
constexpr int kProjHSlabs = 7;                                // K = 200 -> 224
constexpr uint32_t kProjHDirB = kProjHSlabs * 19 * 2048u;     // A fragments of one direction
template <int NPG, int PT = 5> struct ProjGeo {
    static constexpr int kPos = 16 * PT * NPG;
    static constexpr int kPlaneB = kPos * 400 + 512;          // + slack for the K padding over-read of the last row
    static constexpr int kLds = 2 * kPlaneB;
    static constexpr int kThreads = 256 * NPG;
};

template <int NPG, int C0, int NC, bool NT, int PT = 5>
__device__ __forceinline__ void proj_pass_h(const GruProjParams& P, const char* smem, int pg, int dir, int lane, size_t p0) {
    const int n = lane & 15, kq = lane >> 4;
    const char* wb = reinterpret_cast<const char*>(P.w);
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(wb), 0, (int)(2 * kProjHDirB + 2 * 19 * 64 + 16), 0x00020000);
    const uint32_t voff = (uint32_t)lane * 16u;
    const uint32_t soff = (uint32_t)dir * kProjHDirB;
    const float* bias = reinterpret_cast<const float*>(wb + 2 * kProjHDirB) + dir * (19 * 16);
    const float inv = reinterpret_cast<const float*>(wb + 2 * kProjHDirB)[2 * 19 * 16];
    uint32_t bh[PT], bl[PT];
#pragma unroll
    for (int p = 0; p < PT; ++p) {
        bh[p] = (uint32_t)(((pg * PT + p) * 16 + n) * 400 + 16 * kq);
        bl[p] = bh[p] + (uint32_t)ProjGeo<NPG, PT>::kPlaneB;
    }
    OpsHA<NC> a0;
    load_wh<19, C0, NC>(a0, rsrc, voff, soff);
    f32x4 acc[PT][NC];
#pragma unroll
    for (int ct = 0; ct < NC; ++ct) {
        const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + (C0 + ct) * 16 + 4 * kq);
#pragma unroll
        for (int p = 0; p < PT; ++p) acc[p][ct] = bv;
    }
#if !(TAE_PROJ_X & 2)
    conv_accumulate_h<19, C0, NC, PT, kProjHSlabs>(acc, a0, rsrc, voff, soff, smem, bh, bl);
#endif
#pragma unroll
    for (int p = 0; p < PT; ++p) {
        const size_t pos = p0 + (pg * PT + p) * 16 + n;
#if TAE_PROJ_X & 1
        if (pos == (size_t)-1) {
#else
        if (pos < P.npos) {
#endif
            float* dst = P.gi + ((pos >> 4) * 2 + dir) * (size_t)(19 * 256) + (size_t)C0 * 256 + (n * 4 + kq) * 4;
#pragma unroll
            for (int ct = 0; ct < NC; ++ct) {
                // GI is written once and read once by the layer-1 recurrence (4 GB per stack): NT = streaming stores that do
                // not displace the A fragments every workgroup re-reads from L2
                if (NT) __builtin_nontemporal_store(acc[p][ct] * inv, reinterpret_cast<f32x4*>(dst + ct * 256));
                else *reinterpret_cast<f32x4*>(dst + ct * 256) = acc[p][ct] * inv;
            }
        }
    }
}

template <int NPG, bool NT, int PT = 5, int MINW = 2>
__global__ __launch_bounds__(256 * NPG, MINW) void gru_proj_h_kernel(GruProjParams P) {
    using G = ProjGeo<NPG, PT>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const size_t p0 = (size_t)blockIdx.x * G::kPos;
    const int np = (int)min((size_t)G::kPos, P.npos - p0);
    {
        // Y0 arrives as halves, logically [pos][hi 200 | lo 200]: 50 16-byte pieces per position, 25 per plane
        for (int i = tid; i < G::kLds / 16; i += G::kThreads) reinterpret_cast<f32x4*>(smem)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        __syncthreads();
        for (int i = tid; i < ((TAE_PROJ_X & 4) ? 0 : np * 50); i += G::kThreads) {
            const int pos = i / 50, c = i - pos * 50, plane = c >= 25 ? 1 : 0, cc = c - plane * 25;
            const size_t row = p0 + pos;                     // (group, step, block) -> the step's regions (turboae_y0.hpp)
            const char* piece = reinterpret_cast<const char*>(P.yin) + (row >> 4) * kY0StepB + y0_piece((int)(row & 15), cc) + (plane ? y0_lo_add(cc) : 0u);
            *reinterpret_cast<f32x4*>(smem + plane * G::kPlaneB + pos * 400 + cc * 16) = *reinterpret_cast<const f32x4*>(piece);
        }
    }
    __syncthreads();
    const int pg = NPG == 1 ? 0 : (wave & 1), rq = NPG == 1 ? wave : (wave >> 1), dir = rq >> 1;
    if ((rq & 1) == 0) {
        proj_pass_h<NPG, 0, 5, NT, PT>(P, smem, pg, dir, lane, p0);
        proj_pass_h<NPG, 5, 5, NT, PT>(P, smem, pg, dir, lane, p0);
    } else {
        proj_pass_h<NPG, 10, 5, NT, PT>(P, smem, pg, dir, lane, p0);
        proj_pass_h<NPG, 15, 4, NT, PT>(P, smem, pg, dir, lane, p0);
    }
}

// ---- head epilogue: the Linear products arrive from the layer-1 recurrence ---------------------------------------------
// out[f] = fwd[f] + bwd[f] + b[f] -> dec_act -> extrinsic subtraction -> (de)interleave scatter into the other panel, or
// sigmoid + deinterleave for the last half-iteration (decoders.py:103-147); encoder mode: enc_act -> x_tx + partial sums
// (encoders.py:284-298).  One thread per position of the block-group-major workspace, pos' = ((b / 16) * L + t) * 16 + b % 16.
constexpr int kHeadPartThreads = 256;
__global__ __launch_bounds__(kHeadPartThreads) void gru_head_part_kernel(GruHeadParams P) {
    double esum = 0.0, esq = 0.0;
    for (size_t posy = (size_t)blockIdx.x * kHeadPartThreads + threadIdx.x; posy < P.npos; posy += (size_t)gridDim.x * kHeadPartThreads) {
        const size_t row = posy >> 4, grp = row / P.L;
        const int t = (int)(row - grp * P.L);
        const size_t b = grp * 16 + (posy & 15);
        if (b >= (size_t)P.B) continue;
        const f32x4* hp = reinterpret_cast<const f32x4*>(P.y + posy * 16);
        const f32x4 f0 = hp[0], b0 = hp[2];
        float o[8] = {f0.x + b0.x, f0.y + b0.y, f0.z + b0.z, f0.w + b0.w, 0.f, 0.f, 0.f, 0.f};
        if (P.nout > 4) {
            const f32x4 f1 = hp[1], b1 = hp[3];
            o[4] = f1.x + b1.x; o[5] = f1.y + b1.y; o[6] = f1.z + b1.z; o[7] = f1.w + b1.w;
        }
        const size_t pos = b * P.L + t;      // (block, t) order of the X panels
        if (P.enc_stack >= 0) {              // ENC_interRNN: x = enc_act(Linear(2H -> 1)) (encoders.py:284,287,292)
            float v = o[0] + P.b[0];
            v = P.act == 0 ? (v > 0.0f ? v : expm1f(v)) : act_apply(v, P.act);
            P.xtx[pos * 3 + P.enc_stack] = v;
            esum += (double)v;
            esq += (double)v * (double)v;
        } else if (!P.last) {
            // rows of the X panels are 32 bytes [x0, x1, f0 .. f5]: whole 16-byte pieces in and out (a lane's row is 3 200 bytes from its
            // neighbour's, so every 4-byte access was one cache line per lane: 12 such instructions per position, now 5).  The target
            // row's x0, x1 are constants of the call (gru_prep) and channels past F stay zero, so the row can be rewritten whole.
            const f32x4* xc4 = reinterpret_cast<const f32x4*>(P.xcur + pos * kGXW);
            f32x4* xn4 = reinterpret_cast<f32x4*>(P.xnext + (b * P.L + P.ptab[t]) * kGXW);
            const f32x4 c0 = P.extrinsic ? xc4[0] : f32x4{0.f, 0.f, 0.f, 0.f}, c1 = P.extrinsic ? xc4[1] : f32x4{0.f, 0.f, 0.f, 0.f};
            const float xc[6] = {c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
            float v[6];
#pragma unroll
            for (int f = 0; f < 6; ++f) v[f] = f < P.F ? act_apply(o[f] + P.b[f], P.act) - (P.extrinsic ? xc[f] : 0.0f) : 0.0f;      // dec_act, then extrinsic (decoders.py:103-106)
            f32x4 n0 = xn4[0];
            n0.z = v[0]; n0.w = v[1];
            xn4[0] = n0;
            xn4[1] = f32x4{v[2], v[3], v[4], v[5]};
            if (P.tap)
                for (int f = 0; f < P.F; ++f) P.tap[pos * P.F + f] = v[f];
        } else {
            const float v = act_apply(o[0] + P.b[0], P.act);
            P.xdec[b * P.L + P.ptab[t]] = 1.0f / (1.0f + expf(-v));      // sigmoid(deinterleave(dec_act(x_plr))), decoders.py:143-147
        }
    }
    if (P.enc_stack >= 0) {
        // per-workgroup partial sums for power_constraint (encoders.py:107-108), fixed-order tree
        __shared__ double red[2 * kHeadPartThreads];
        const int tid = threadIdx.x;
        red[tid] = esum;
        red[kHeadPartThreads + tid] = esq;
        __syncthreads();
        for (int off = kHeadPartThreads / 2; off > 0; off >>= 1) {
            if (tid < off) { red[tid] += red[tid + off]; red[kHeadPartThreads + tid] += red[kHeadPartThreads + tid + off]; }
            __syncthreads();
        }
        if (tid == 0) { P.partials[2 * blockIdx.x] = red[0]; P.partials[2 * blockIdx.x + 1] = red[kHeadPartThreads]; }
    }
}

hipError_t launch_gru_head_part(const GruHeadParams& P, hipStream_t st) {
    // the same grid as launch_gru_head (the encoder's partial-sum slots are sized by gru_head_grid)
    hipLaunchKernelGGL(gru_head_part_kernel, dim3(gru_head_grid(P.npos)), dim3(kHeadPartThreads), 0, st, P);
    return hipGetLastError();
}

int gru_rec_h_lds_bytes(bool layer0) { return layer0 ? kRec0B : kRec1B; }

hipError_t launch_gru_rec_h(bool layer0, const GruRecParams& P, hipStream_t st) {
    const int lds = gru_rec_h_lds_bytes(layer0);
    int nw = 8;
    while (nw > 1 && 2 * ((P.B + 16 * nw - 1) / (16 * nw)) < 256) nw >>= 1;
    static const int nw_env = [] { const char* e = tae::debug_knob("TAE_GRU_NW"); return e ? atoi(e) : 0; }();     // experiments: read once
    if (nw_env == 1 || nw_env == 2 || nw_env == 4 || nw_env == 8) nw = nw_env;
    const dim3 grid((P.B + 16 * nw - 1) / (16 * nw), 2);
    const void* fn = layer0 ? reinterpret_cast<const void*>(gru_rec_h_kernel<true>) : reinterpret_cast<const void*>(gru_rec_h_kernel<false>);
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return e;
    if (layer0) hipLaunchKernelGGL(gru_rec_h_kernel<true>, grid, dim3(64 * nw), lds, st, P);
    else hipLaunchKernelGGL(gru_rec_h_kernel<false>, grid, dim3(64 * nw), lds, st, P);
    return hipGetLastError();
}

template <int NPG, bool NT, int PT = 5, int MINW = 2>
static hipError_t launch_gru_proj_h_t(const GruProjParams& P, hipStream_t st) {
    using G = ProjGeo<NPG, PT>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gru_proj_h_kernel<NPG, NT, PT, MINW>), hipFuncAttributeMaxDynamicSharedMemorySize, G::kLds);
    if (e != hipSuccess) return e;
    const dim3 grid((unsigned)((P.npos + G::kPos - 1) / G::kPos));
    hipLaunchKernelGGL((gru_proj_h_kernel<NPG, NT, PT, MINW>), grid, dim3(G::kThreads), G::kLds, st, P);
    return hipGetLastError();
}

hipError_t launch_gru_proj_h(const GruProjParams& P, hipStream_t st) {
    static const int npg = [] { const char* e = tae::debug_knob("TAE_GRU_PROJ_PG"); return (e && atoi(e) == 2) ? 2 : 1; }();     // experiments
    static const int nt = [] { const char* e = tae::debug_knob("TAE_GRU_PROJ_NT"); return e ? atoi(e) : 1; }();
    if (npg == 2) return nt ? launch_gru_proj_h_t<2, true>(P, st) : launch_gru_proj_h_t<2, false>(P, st);
    return nt ? launch_gru_proj_h_t<1, true>(P, st) : launch_gru_proj_h_t<1, false>(P, st);
}

}  // namespace tae
