// Generic fp32 kernels: the reference's layer arithmetic one layer at a time, for every configuration the reference's argument
// parser accepts on this path but the MFMA kernels do not instantiate - channel widths above 124 (100 for recurrent cells), kernel sizes above 9 (any odd
// size), num_iter_ft above 6, LSTM / vanilla-RNN cells (decoders.py:27-32, encoders.py:242-253), ENC_interRNN with
// enc_num_layer != 2, an RNN encoder in front of the (then dense, decoders.py:173-176) CNN decoder - and `precision = f32` for the
// variants whose MFMA kernels exist in the fp16-split arithmetic only (DenseSameShapeConv1d, kernel sizes 7 / 9).
//
// One launch per layer, activations round-trip through HBM between layers, everything in plain fp32 (operands and accumulation: no
// range window - this is also where the range fall-back lands for dense stacks and 7 / 9 taps).  Since r04 the multiplications run
// on the fp32 matrix cores (v_mfma_f32_16x16x4_f32: gen_conv_mfma_kernel, gen_proj_mfma_kernel, gen_rnn_mfma_kernel); the r03
// vector-ALU kernels remain for recurrent widths above 128 / not a multiple of 4 and, behind TAE_GEN_CONV=valu / TAE_GEN_RNN=valu,
// as the A/B baseline.  Not the benchmark path (DESIGN.md 3.9 has the numbers).
// layers.  Every op follows the PyTorch definition the reference relies on:
//   SameShapeConv1d / DenseSameShapeConv1d   cnn_utils.py:6-82       y[co,t] = b[co] + sum_ci sum_j W[co,ci,j] x[ci,t+j-k//2]; ELU
//   torch.nn.GRU / LSTM / RNN (bidirectional, batch_first, n layers)   decoders.py:41-49, encoders.py:251-268
//   Linear heads, enc_act / dec_act, extrinsic subtraction, (de)interleave, sigmoid    decoders.py:84-149,206-269, encoders.py:281-377
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <string>
#include <vector>

#include "../../include/turboae_hip.h"
#include "turboae_internal.hpp"
#include "turboae_device.hpp"

#ifndef TAE_GEN_MJ
#define TAE_GEN_MJ 2        // taps of the weights staged together by gen_conv_mfma_kernel: 4 = 107 KB of LDS at k = 5 = one workgroup per CU, 17 % slower (profiles/r04_gen_mj_ab.txt)
#endif

namespace tae {

namespace {

constexpr int kConvPos = 32;        // positions per workgroup
constexpr int kConvCh = 64;         // output channels per workgroup
constexpr int kConvCi = 128;        // input channels staged per pass

// y[b, t, coff + co] = act(bias[co] + sum_{ci, j} wt[(ci * k + j) * cout + co] * x[b, t + j - k / 2, ci]), zero outside the block
// (Conv1d padding = k // 2, cnn_utils.py:16,58).  act: 0 none (Linear, RNN input projections), 1 ELU (every conv layer).
__global__ __launch_bounds__(256) void gen_conv_kernel(const float* __restrict__ x, int ldx, int cin, const float* __restrict__ wt,
                                                       const float* __restrict__ bias, float* __restrict__ y, int ldy, int coff, int cout,
                                                       int k, int L, int act) {
    extern __shared__ float xs[];                      // [(kConvPos + k - 1)][kConvCi]
    const int tid = threadIdx.x, col = tid & 63, pg = tid >> 6;
    const int t0 = blockIdx.x * kConvPos, co = blockIdx.z * kConvCh + col, pad = k / 2;
    const size_t b = blockIdx.y;
    const int rows = kConvPos + k - 1;
    float acc[8];
#pragma unroll
    for (int p = 0; p < 8; ++p) acc[p] = 0.0f;
    for (int c0 = 0; c0 < cin; c0 += kConvCi) {
        const int nc = min(kConvCi, cin - c0);
        __syncthreads();
        for (int i = tid; i < rows * nc; i += 256) {
            const int r = i / nc, c = i - r * nc, t = t0 - pad + r;
            xs[r * kConvCi + c] = (t >= 0 && t < L) ? x[(b * L + t) * (size_t)ldx + c0 + c] : 0.0f;
        }
        __syncthreads();
        if (co < cout) {
            for (int c = 0; c < nc; ++c)
                for (int j = 0; j < k; ++j) {
                    const float w = wt[((size_t)(c0 + c) * k + j) * cout + co];
                    const float* xr = xs + (pg * 8 + j) * kConvCi + c;
#pragma unroll
                    for (int p = 0; p < 8; ++p) acc[p] = fmaf(w, xr[p * kConvCi], acc[p]);
                }
        }
    }
    if (co >= cout) return;
    const float bv = bias[co];
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        const int t = t0 + pg * 8 + p;
        if (t < L) {
            float v = acc[p] + bv;
            if (act == 1) v = v > 0.0f ? v : expm1f(v);
            y[(b * L + t) * (size_t)ldy + coff + co] = v;
        }
    }
}

// The same convolution on the fp32 matrix cores (r04; the vector-ALU kernel above reached 13 TFLOP/s on the LSTM decoder's input
// projection, 585 of its 857 ms per forward): v_mfma_f32_16x16x4_f32 with A = weights (M = output channel), B = activations
// (N = position), fp32 operands and accumulation - no operand split, so no range limits (this is also the fall-back arithmetic).
// Workgroup = 8 waves = 256 positions (FLATTENED over the batch, so short blocks fill tiles; a tap that would reach across a block
// boundary is masked when the operand is read) x 128 output channels, 171 flops per staged float; wave = 32 positions x 128
// channels = 16 accumulator tiles, 2 + 2 LDS reads per 16 MFMAs.  Input channels are staged 32 at a time (rows padded to 36 floats:
// the 16 positions x 4 k of one operand read fall into 64 different banks), weights 32 channels x up to 2 taps x 128 outputs with
// the outputs of a row permuted (channel 16 mt + n at n * 8 + mt) so that a lane's 8 A operands are two 16-byte reads.
constexpr int kMP = 256, kMC = 128, kMK = 32, kMKP = 36, kMWP = 136, kMJ = TAE_GEN_MJ;

__global__ __launch_bounds__(512) void gen_conv_mfma_kernel(const float* __restrict__ x, int ldx, int cin, const float* __restrict__ wt,
                                                            const float* __restrict__ bias, float* __restrict__ y, int ldy, int coff, int cout,
                                                            int k, int L, size_t np, int act, unsigned ctiles, unsigned ntiles, int xcd) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int rows = kMP + k - 1, pad = k / 2;
    float* xs = sm;                                    // [rows][kMKP]: flattened positions p0 - pad ...
    float* ws = sm + rows * kMKP;                      // [taps staged][kMK][kMWP]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = lane & 15, q = lane >> 4;
    // 1-D grid, XCD-aware: workgroup i runs on XCD i % 8, so XCD k takes the k-th contiguous eighth of the (position tile, channel
    // tile) list, channel tile fastest - the workgroups that share an activation tile follow each other on ONE L2
    const unsigned per_xcd = gridDim.x / 8, tile = xcd ? (blockIdx.x % 8) * per_xcd + blockIdx.x / 8 : blockIdx.x;
    if (tile >= ntiles) return;
    const size_t p0 = (size_t)(tile / ctiles) * kMP;
    const int ch0 = (int)(tile % ctiles) * kMC;
    const bool xv = (ldx & 3) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0;
    const bool wv = (cout & 3) == 0 && (reinterpret_cast<uintptr_t>(wt) & 15) == 0;
    const bool yv = (ldy & 3) == 0 && (coff & 3) == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0;
    int tn[2];                                         // time index of this lane's two positions inside their blocks
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const size_t p = p0 + wave * 32 + nt * 16 + n;
        tn[nt] = p < np ? (int)(p % (size_t)L) : -(1 << 24);
    }
    f32x4 acc[8][2];
#pragma unroll
    for (int mt = 0; mt < 8; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int c0 = 0; c0 < cin; c0 += kMK) {
        const int nc = min(kMK, cin - c0), kend = (nc + 3) & ~3;
        __syncthreads();
        for (int i = tid; i < rows * (kMK / 4); i += 512) {
            const int r = i / (kMK / 4), c = (i % (kMK / 4)) * 4;
            const long long p = (long long)p0 - pad + r;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (p >= 0 && (size_t)p < np && c < nc) {
                const float* src = x + (size_t)p * ldx + c0 + c;
                if (xv && c + 4 <= nc) v = *reinterpret_cast<const f32x4*>(src);
                else
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = c + e < nc ? src[e] : 0.0f;
            }
            *reinterpret_cast<f32x4*>(xs + r * kMKP + c) = v;
        }
        for (int j0 = 0; j0 < k; j0 += kMJ) {
            const int nj = min(kMJ, k - j0);
            if (j0) __syncthreads();
            for (int i = tid; i < nj * kMK * (kMC / 4); i += 512) {
                const int m = (i % (kMC / 4)) * 4, c = (i / (kMC / 4)) % kMK, jj = i / ((kMC / 4) * kMK);
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (c < nc && ch0 + m < cout) {
                    const float* src = wt + ((size_t)(c0 + c) * k + j0 + jj) * cout + ch0 + m;
                    if (wv && ch0 + m + 4 <= cout) v = *reinterpret_cast<const f32x4*>(src);
                    else
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = ch0 + m + e < cout ? src[e] : 0.0f;
                }
                float* dst = ws + (jj * kMK + c) * kMWP + (m >> 4);      // m .. m + 3 share a channel tile (m is a multiple of 4)
#pragma unroll
                for (int e = 0; e < 4; ++e) dst[((m & 15) + e) * 8] = v[e];
            }
            __syncthreads();
            for (int jj = 0; jj < nj; ++jj) {
                const int j = j0 + jj;
                const bool ok0 = (unsigned)(tn[0] + j - pad) < (unsigned)L, ok1 = (unsigned)(tn[1] + j - pad) < (unsigned)L;
                const float* xr = xs + (wave * 32 + n + j) * kMKP + q;
                const float* wr = ws + (jj * kMK + q) * kMWP + n * 8;
                for (int kk = 0; kk < kend; kk += 4) {
                    const float b0 = ok0 ? xr[kk] : 0.0f, b1 = ok1 ? xr[16 * kMKP + kk] : 0.0f;
                    const f32x4 alo = *reinterpret_cast<const f32x4*>(wr + kk * kMWP), ahi = *reinterpret_cast<const f32x4*>(wr + kk * kMWP + 4);
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) {
                        acc[mt][0] = mfma16x16x4(alo[mt], b0, acc[mt][0]);
                        acc[mt][1] = mfma16x16x4(alo[mt], b1, acc[mt][1]);
                    }
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) {
                        acc[4 + mt][0] = mfma16x16x4(ahi[mt], b0, acc[4 + mt][0]);
                        acc[4 + mt][1] = mfma16x16x4(ahi[mt], b1, acc[4 + mt][1]);
                    }
                }
            }
        }
    }
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const size_t p = p0 + wave * 32 + nt * 16 + n;
        if (p >= np) continue;
        float* yr = y + p * (size_t)ldy + coff;
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) {
            const int ch = ch0 + mt * 16 + 4 * q;
            if (ch >= cout) continue;
            f32x4 v;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float t = acc[mt][nt][i] + (ch + i < cout ? bias[ch + i] : 0.0f);
                if (act == 1) t = t > 0.0f ? t : expm1f(t);
                v[i] = t;
            }
            if (yv && ch + 4 <= cout) *reinterpret_cast<f32x4*>(yr + ch) = v;
            else
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (ch + i < cout) yr[ch + i] = v[i];
        }
    }
}

// k = 1 (Linear heads, RNN input projections: the largest generic launches, K = 2 H -> 2 G H at every position): the same tiles
// without taps / halo / masks, and software-pipelined - the 16-byte loads of input-channel chunk c + 1 (4 of x, 2 of W per thread)
// are in flight while chunk c is multiplied, so a workgroup hides its own global latency instead of leaving that to its neighbour.
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void gen_proj_mfma_kernel(const float* __restrict__ x, int ldx, int cin, const float* __restrict__ wt,
                                                            const float* __restrict__ bias, float* __restrict__ y, int ldy, int coff, int cout,
                                                            size_t np, int act, unsigned ctiles, unsigned ntiles, int xcd) {
    __shared__ __attribute__((aligned(16))) float xs[kMP * kMKP];
    __shared__ __attribute__((aligned(16))) float ws[kMK * kMWP];
    const int tid = threadIdx.x, lane = tid & 63, n = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // 1-D grid, XCD-aware: workgroup i runs on XCD i % 8, so XCD k takes the k-th contiguous eighth of the (position tile, channel
    // tile) list, channel tile fastest - the workgroups that share an activation tile follow each other on ONE L2
    const unsigned per_xcd = gridDim.x / 8, tile = xcd ? (blockIdx.x % 8) * per_xcd + blockIdx.x / 8 : blockIdx.x;
    if (tile >= ntiles) return;
    const size_t p0 = (size_t)(tile / ctiles) * kMP;
    const int ch0 = (int)(tile % ctiles) * kMC;
    const bool xv = (ldx & 3) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0;
    const bool wv = (cout & 3) == 0 && (reinterpret_cast<uintptr_t>(wt) & 15) == 0;
    const bool yv = (ldy & 3) == 0 && (coff & 3) == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0;
    f32x4 rx[4], rw[2];
    auto fetch = [&](int c0) {
        const int nc = min(kMK, cin - c0);
#pragma unroll
        for (int e4 = 0; e4 < 4; ++e4) {
            const int i = tid + 512 * e4, r = i / (kMK / 4), c = (i % (kMK / 4)) * 4;
            const size_t p = p0 + r;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (p < np && c < nc) {
                const float* src = x + p * (size_t)ldx + c0 + c;
                if (xv && c + 4 <= nc) v = *reinterpret_cast<const f32x4*>(src);
                else
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = c + e < nc ? src[e] : 0.0f;
            }
            rx[e4] = v;
        }
#pragma unroll
        for (int e4 = 0; e4 < 2; ++e4) {
            const int i = tid + 512 * e4, m = (i % (kMC / 4)) * 4, c = i / (kMC / 4);
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (c < nc && ch0 + m < cout) {
                const float* src = wt + (size_t)(c0 + c) * cout + ch0 + m;
                if (wv && ch0 + m + 4 <= cout) v = *reinterpret_cast<const f32x4*>(src);
                else
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = ch0 + m + e < cout ? src[e] : 0.0f;
            }
            rw[e4] = v;
        }
    };
    f32x4 acc[8][2];
#pragma unroll
    for (int mt = 0; mt < 8; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    fetch(0);
    for (int c0 = 0; c0 < cin; c0 += kMK) {
        const int kend = (min(kMK, cin - c0) + 3) & ~3;
        __syncthreads();
#pragma unroll
        for (int e4 = 0; e4 < 4; ++e4) {
            const int i = tid + 512 * e4;
            *reinterpret_cast<f32x4*>(xs + (i / (kMK / 4)) * kMKP + (i % (kMK / 4)) * 4) = rx[e4];
        }
#pragma unroll
        for (int e4 = 0; e4 < 2; ++e4) {
            const int i = tid + 512 * e4, m = (i % (kMC / 4)) * 4, c = i / (kMC / 4);
            float* dst = ws + c * kMWP + (m >> 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) dst[((m & 15) + e) * 8] = rw[e4][e];
        }
        __syncthreads();
        if (c0 + kMK < cin) fetch(c0 + kMK);
        const float* xr = xs + (wave * 32 + n) * kMKP + q;
        const float* wr = ws + q * kMWP + n * 8;
        for (int kk = 0; kk < kend; kk += 4) {
            const float b0 = xr[kk], b1 = xr[16 * kMKP + kk];
            const f32x4 alo = *reinterpret_cast<const f32x4*>(wr + kk * kMWP), ahi = *reinterpret_cast<const f32x4*>(wr + kk * kMWP + 4);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                acc[mt][0] = mfma16x16x4(alo[mt], b0, acc[mt][0]);
                acc[mt][1] = mfma16x16x4(alo[mt], b1, acc[mt][1]);
            }
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                acc[4 + mt][0] = mfma16x16x4(ahi[mt], b0, acc[4 + mt][0]);
                acc[4 + mt][1] = mfma16x16x4(ahi[mt], b1, acc[4 + mt][1]);
            }
        }
    }
    // the lane index again, from mbcnt: nothing lane-dependent has to stay live across the K loop for the epilogue (at the 128
    // registers of two workgroups per CU that one value was spilled: 8 bytes of scratch per lane for a single reload)
    const int lane_e = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)), n_e = lane_e & 15, q_e = lane_e >> 4;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const size_t p = p0 + wave * 32 + nt * 16 + n_e;
        if (p >= np) continue;
        float* yr = y + p * (size_t)ldy + coff;
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) {
            const int ch = ch0 + mt * 16 + 4 * q_e;
            if (ch >= cout) continue;
            f32x4 v;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float t = acc[mt][nt][i] + (ch + i < cout ? bias[ch + i] : 0.0f);
                if (act == 1) t = t > 0.0f ? t : expm1f(t);
                v[i] = t;
            }
            if (yv && ch + 4 <= cout) *reinterpret_cast<f32x4*>(yr + ch) = v;
            else
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (ch + i < cout) yr[ch + i] = v[i];
        }
    }
}

__device__ __forceinline__ float sigm(float x) { return 1.0f / (1.0f + expf(-x)); }

// One (block, direction) of one recurrent layer; thread u owns hidden unit u.  gi (B, L, 2, G, H) holds W_ih x + b_ih for both
// directions; whh_t = W_hh transposed, [k][G * H].  PyTorch cells:
//   GRU  (r, z, n):    r = s(gi_r + W_hr h + b_hr), z = s(gi_z + W_hz h + b_hz), n = tanh(gi_n + r * (W_hn h + b_hn)), h' = (1 - z) n + z h
//   LSTM (i, f, g, o): c' = s(f) c + s(i) tanh(g), h' = s(o) tanh(c')      (pre-activations gi_* + W_h* h + b_h*)
//   RNN:               h' = tanh(gi + W_hh h + b_hh)
// One workgroup = NB blocks x one direction, thread = hidden unit: every recurrent weight a thread fetches is used for its NB
// blocks (r03: NB = 1, i.e. the whole W_hh - 160 KB for an LSTM - re-read from L2 per block and step; r04: up to 8 blocks share it).
// The per-(block, unit) FMA chain is the same for every NB (k ascending), so results do not depend on the batch or on NB.
template <int NB>
__global__ void gen_rnn_kernel(int cell, const float* __restrict__ gi, const float* __restrict__ whh_t0, const float* __restrict__ whh_t1,
                               const float* __restrict__ bhh0, const float* __restrict__ bhh1, float* __restrict__ y, int H, int L, int B) {
    extern __shared__ float hs[];                      // h_{t-1}: [NB][H]
    const int u = threadIdx.x, dir = blockIdx.y;
    const size_t b0 = (size_t)blockIdx.x * NB;
    const int G = cell == 0 ? 3 : (cell == 1 ? 4 : 1);
    const float* whh = dir ? whh_t1 : whh_t0;
    const float* bhh = dir ? bhh1 : bhh0;
    float c[NB], hprev[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) { c[j] = 0.0f; hprev[j] = 0.0f; if (u < H) hs[j * H + u] = 0.0f; }
    __syncthreads();
    for (int s = 0; s < L; ++s) {
        const int t = dir ? L - 1 - s : s;
        float hn[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) hn[j] = 0.0f;
        if (u < H) {
            float a[4][NB];
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int j = 0; j < NB; ++j) a[g][j] = 0.0f;
            for (int k = 0; k < H; ++k) {
                const float* w = whh + (size_t)k * G * H + u;
                float wg[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) wg[g] = g < G ? w[g * H] : 0.0f;
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    const float hk = hs[j * H + k];
#pragma unroll
                    for (int g = 0; g < 4; ++g) a[g][j] = fmaf(wg[g], hk, a[g][j]);
                }
            }
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const size_t b = b0 + j;
                if (b >= (size_t)B) continue;
                const float* gp = gi + ((b * L + t) * 2 + dir) * (size_t)(G * H) + u;
                if (cell == 0) {
                    const float r = sigm(gp[0] + (a[0][j] + bhh[u]));
                    const float z = sigm(gp[H] + (a[1][j] + bhh[H + u]));
                    const float n = tanhf(gp[2 * H] + r * (a[2][j] + bhh[2 * H + u]));
                    hn[j] = (1.0f - z) * n + z * hprev[j];
                } else if (cell == 1) {
                    const float ig = sigm(gp[0] + (a[0][j] + bhh[u]));
                    const float fg = sigm(gp[H] + (a[1][j] + bhh[H + u]));
                    const float gg = tanhf(gp[2 * H] + (a[2][j] + bhh[2 * H + u]));
                    const float og = sigm(gp[3 * H] + (a[3][j] + bhh[3 * H + u]));
                    c[j] = fg * c[j] + ig * gg;
                    hn[j] = og * tanhf(c[j]);
                } else {
                    hn[j] = tanhf(gp[0] + (a[0][j] + bhh[u]));
                }
                y[(b * L + t) * (size_t)(2 * H) + dir * H + u] = hn[j];
            }
        }
        __syncthreads();
        if (u < H) {
#pragma unroll
            for (int j = 0; j < NB; ++j) { hs[j * H + u] = hn[j]; hprev[j] = hn[j]; }
        }
        __syncthreads();
    }
}

// The same recurrence on the fp32 matrix cores for H <= 128, H a multiple of 4 (r04; the vector-ALU kernel above took 287 of the LSTM decoder's 857 ms):
// workgroup = 16 blocks x one direction, wave w owns hidden units [16w, 16w + 16) of every gate and keeps ITS rows of W_hh in
// registers for the whole sequence (G * ceil(H / 4) operand registers: 100 for the LSTM at H = 100 - the matrix that does not fit in
// LDS fits in the register files of 7 waves).  Per step: a = W_hh h_{t-1} as G accumulator tiles (M = 16 units, N = 16 blocks,
// K = H in steps of 4; v_mfma_f32_16x16x4_f32, fp32 operands - no range limits), h_{t-1} read as B operands from LDS (rows of
// 4 KS + {0, 4} floats: 16 blocks x 4 k fall into 64 banks), gates on the lane's 4 units x 1 block, h_t to the other LDS buffer
// (one barrier per step) and to y.  sigmoid / tanh through v_exp_f32 + v_rcp_f32 as in turboae_gru.hip.
__device__ __forceinline__ float sigm_fast(float x) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896341f * x)); }
__device__ __forceinline__ float tanh_fast(float x) { return fmaf(-2.0f, __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.88539008177792681f * x)), 1.0f); }

// XK > 0 (layer inputs of <= 4 XK channels, i.e. every first layer: 2 + num_iter_ft <= 8 decoder inputs, 1 encoder input): the input
// projection W_ih x_t + b_ih is XK more MFMAs per gate and step on operands fetched from x itself, so GI (2 G H floats per position
// written by a projection launch and read back here: 2 x 5.2 GB per LSTM layer at 16 384 blocks) never exists.
template <int G, int KS, int NT, int XK>
__global__ __launch_bounds__(512) void gen_rnn_mfma_kernel(const float* __restrict__ gi, const float* __restrict__ xin, int ldin, int cin,
                                                           const float* __restrict__ wih_t, const float* __restrict__ bih,
                                                           const float* __restrict__ whh_t0, const float* __restrict__ whh_t1,
                                                           const float* __restrict__ bhh0, const float* __restrict__ bhh1, float* __restrict__ y, int H, int L, int B) {
    constexpr int HP = 4 * KS + ((4 * KS) % 8 == 4 ? 0 : 4), NBLK = 16 * NT, XKR = XK > 0 ? XK : 1;
    extern __shared__ float hs[];                      // h_{t-1} / h_t: [2][NBLK][HP]
    const int tid = threadIdx.x, lane = tid & 63, n = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int dir = blockIdx.y, b0 = blockIdx.x * NBLK, nb = min(NBLK, B - b0), GH = G * H;
    bool valid[NT];
    size_t bn[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        valid[nt] = n + 16 * nt < nb;
        bn[nt] = (size_t)b0 + (valid[nt] ? n + 16 * nt : nb - 1);
    }
    const int u0 = wave * 16, ul = u0 + 4 * q;         // first of this lane's 4 units in the accumulator tiles
    const float* whh = dir ? whh_t1 : whh_t0;
    const float* bhh = dir ? bhh1 : bhh0;
    float w[G][KS], wx[G][XKR];
#pragma unroll
    for (int g = 0; g < G; ++g) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int u = u0 + n, k = 4 * ks + q;
            w[g][ks] = (u < H && k < H) ? whh[(size_t)k * GH + g * H + u] : 0.0f;
        }
#pragma unroll
        for (int j = 0; j < XKR; ++j) {
            const int u = u0 + n, k = 4 * j + q;
            wx[g][j] = (XK > 0 && u < H && k < cin) ? wih_t[(size_t)k * (2 * GH) + dir * GH + g * H + u] : 0.0f;
        }
    }
    f32x4 bias[G], biasx[G];
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            bias[g][i] = ul + i < H ? bhh[g * H + ul + i] : 0.0f;
            biasx[g][i] = (XK > 0 && ul + i < H) ? bih[dir * GH + g * H + ul + i] : 0.0f;
        }
    for (int i = tid; i < 2 * NBLK * HP; i += (int)blockDim.x) hs[i] = 0.0f;
    f32x4 hprev[NT], c[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) hprev[nt] = c[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    // GI (or x) of step s + 1 is fetched while step s computes: unconditional loads from clamped addresses (a predicated load
    // compiles to a branch with s_waitcnt vmcnt(0) behind it - four serialised HBM latencies per step in the first cut of this
    // kernel, 4.7 instead of 3.4 ms per launch); lanes of units >= H compute on whatever they got and never store.
    const int ulc = min(ul, H - 4);                    // H is a multiple of 4 here (the launcher sends other widths to gen_rnn_kernel)
    f32x4 gnext[NT][G];
    float xnext[NT][XKR];
    auto fetch = [&](int s) {
        const int t = dir ? L - 1 - s : s;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            if constexpr (XK > 0) {
                const float* xp = xin + (bn[nt] * L + t) * (size_t)ldin;
#pragma unroll
                for (int j = 0; j < XKR; ++j) xnext[nt][j] = xp[min(4 * j + q, cin - 1)];
            } else {
                const float* gp = gi + ((bn[nt] * L + t) * 2 + dir) * (size_t)GH + ulc;
#pragma unroll
                for (int g = 0; g < G; ++g) gnext[nt][g] = *reinterpret_cast<const f32x4*>(gp + g * H);
            }
        }
    };
    fetch(0);
    for (int s = 0; s < L; ++s) {
        const int t = dir ? L - 1 - s : s;
        f32x4 gv[NT][G];
        float xb[NT][XKR];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
            for (int g = 0; g < G; ++g) gv[nt][g] = gnext[nt][g];
#pragma unroll
            for (int j = 0; j < XKR; ++j) xb[nt][j] = 4 * j + q < cin ? xnext[nt][j] : 0.0f;
        }
        fetch(min(s + 1, L - 1));
        if constexpr (XK > 0) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    gv[nt][g] = biasx[g];
#pragma unroll
                    for (int j = 0; j < XKR; ++j) gv[nt][g] = mfma16x16x4(wx[g][j], xb[nt][j], gv[nt][g]);
                }
        }
        const float* hc = hs + (s & 1) * NBLK * HP + n * HP + q;
        f32x4 acc[NT][G];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int g = 0; g < G; ++g) acc[nt][g] = bias[g];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const float hb = hc[nt * 16 * HP + 4 * ks];
#pragma unroll
                for (int g = 0; g < G; ++g) acc[nt][g] = mfma16x16x4(w[g][ks], hb, acc[nt][g]);
            }
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            f32x4 hn;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if constexpr (G == 3) {
                    const float r = sigm_fast(gv[nt][0][i] + acc[nt][0][i]);
                    const float z = sigm_fast(gv[nt][1][i] + acc[nt][1][i]);
                    const float nn = tanh_fast(fmaf(r, acc[nt][2][i], gv[nt][2][i]));
                    hn[i] = fmaf(z, hprev[nt][i] - nn, nn);
                } else if constexpr (G == 4) {
                    const float ig = sigm_fast(gv[nt][0][i] + acc[nt][0][i]);
                    const float fg = sigm_fast(gv[nt][1][i] + acc[nt][1][i]);
                    const float gg = tanh_fast(gv[nt][2][i] + acc[nt][2][i]);
                    const float og = sigm_fast(gv[nt][3][i] + acc[nt][3][i]);
                    c[nt][i] = fmaf(fg, c[nt][i], ig * gg);
                    hn[i] = og * tanh_fast(c[nt][i]);
                } else {
                    hn[i] = tanh_fast(gv[nt][0][i] + acc[nt][0][i]);
                }
            }
            hprev[nt] = hn;
            float* hnext = hs + ((s + 1) & 1) * NBLK * HP + (n + 16 * nt) * HP + ul;
            float* yp = y + (bn[nt] * L + t) * (size_t)(2 * H) + dir * H + ul;
            if (ul < H) {
                *reinterpret_cast<f32x4*>(hnext) = hn;
                if (valid[nt]) *reinterpret_cast<f32x4*>(yp) = hn;
            }
        }
        __syncthreads();
    }
}

struct RnnArgs { const float *gi, *xin; int ldin, cin; const float *wih_t, *bih, *w0, *w1, *b0, *b1; float* y; int H, L, B; };

// 16 blocks per workgroup.  32 (NT = 2: two position tiles share the register-resident W_hh and the per-step barrier) measured the
// same - 4.65 vs 4.73 ms per launch of the LSTM decoder at 16 384 blocks, profiles/r04_gen_rnn_nt_ab.txt: the step is bound by the
// fp32 MFMA issue, not by the barrier - and is not instantiated.
template <int G, int XK>
static hipError_t launch_rnn_mfma_x(const RnnArgs& a, hipStream_t st) {
    const dim3 grid((a.B + 15) / 16, 2), block(64 * ((a.H + 15) / 16));
    const int ks = (a.H + 3) / 4;
#define TAE_RNN_LAUNCH(KS) hipLaunchKernelGGL((gen_rnn_mfma_kernel<G, KS, 1, XK>), grid, block, 2 * 16 * (4 * KS + ((4 * KS) % 8 == 4 ? 0 : 4)) * sizeof(float), st, \
                                              a.gi, a.xin, a.ldin, a.cin, a.wih_t, a.bih, a.w0, a.w1, a.b0, a.b1, a.y, a.H, a.L, a.B)
    if (ks <= 8) TAE_RNN_LAUNCH(8);
    else if (ks <= 16) TAE_RNN_LAUNCH(16);
    else if (ks <= 25) TAE_RNN_LAUNCH(25);
    else TAE_RNN_LAUNCH(32);
#undef TAE_RNN_LAUNCH
    return hipGetLastError();
}

template <int G>
static hipError_t launch_rnn_mfma(const RnnArgs& a, bool fused, hipStream_t st) {
    return fused ? launch_rnn_mfma_x<G, 2>(a, st) : launch_rnn_mfma_x<G, 0>(a, st);
}

// decoder stack inputs: XA = [r_sys, r_par1, prior = 0...], XB = [r_sys_int, r_par2, 0...] (decoders.py:87-93,221-227), W = 2 + F wide
__global__ void gen_prep_dec_kernel(const float* __restrict__ rx, const int32_t* __restrict__ perm, float* __restrict__ XA, float* __restrict__ XB,
                                    size_t B, int L, int W) {
    const size_t n = B * L;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t b = i / L;
        const int t = (int)(i - b * L);
        const float* r = rx + i * 3;
        float* a = XA + i * W;
        float* c = XB + i * W;
        a[0] = r[0]; a[1] = r[1];
        c[0] = rx[(b * L + perm[t]) * 3]; c[1] = r[2];
        for (int f = 2; f < W; ++f) { a[f] = 0.0f; c[f] = 0.0f; }
    }
}

// encoder stack input (B, L, 1): CNN encoders see 2u - 1 (encoders.py:362), the RNN encoder the raw bits (encoders.py:283);
// the third stack the interleaved sequence (encoders.py:369 / :289)
__global__ void gen_prep_enc_kernel(const float* __restrict__ u, const int32_t* __restrict__ perm, float* __restrict__ X, size_t B, int L,
                                    int interleaved, int bipolar) {
    const size_t n = B * L;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t b = i / L;
        const int t = (int)(i - b * L);
        const float v = u[b * L + (interleaved ? perm[t] : t)];
        X[i] = bipolar ? 2.0f * v - 1.0f : v;
    }
}

__global__ void gen_copy_rows_kernel(const float* __restrict__ src, int lds_, float* __restrict__ dst, int ldd, int w, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n * w; i += (size_t)gridDim.x * blockDim.x) {
        const size_t r = i / w;
        const int c = (int)(i - r * w);
        dst[r * ldd + c] = src[r * lds_ + c];
    }
}

// o (B * L, nout) = Linear output (bias included).  Decoder: x_plr = act(o) [dec_act on the RNN decoder, decoders.py:103,115; none
// on DEC_LargeCNN] - extrinsic input (decoders.py:105-106,117-118,235-236,246-247), scattered to the other panel at the
// (de)interleaved row (interleave after dec1: row inv[t]; deinterleave after dec2: row p[t]); last half-iteration:
// sigmoid(deinterleave(act(o))) (decoders.py:143-147,262-267).  Encoder (enc_stack >= 0): x_tx[., s] = enc_act(o).
__global__ void gen_head_kernel(const float* __restrict__ o, int nout, const float* __restrict__ xcur, float* __restrict__ xnext, int W,
                                float* __restrict__ xdec, const int32_t* __restrict__ ptab, size_t B, int L, int F, int extrinsic, int last,
                                int act, int enc_stack, float* __restrict__ xtx, float* __restrict__ tap) {
    const size_t n = B * L;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t b = i / L;
        const int t = (int)(i - b * L);
        const float* v = o + i * nout;
        if (enc_stack >= 0) {
            const float x = v[0];
            xtx[i * 3 + enc_stack] = act == 0 ? (x > 0.0f ? x : expm1f(x)) : act_apply(x, act);
        } else if (!last) {
            const float* xc = xcur + i * W + 2;
            float* xn = xnext + (b * L + ptab[t]) * W + 2;
            for (int f = 0; f < F; ++f) {
                xn[f] = act_apply(v[f], act) - (extrinsic ? xc[f] : 0.0f);
                if (tap) tap[i * F + f] = xn[f];      // tae_decode_taps: [b][position in this stack's order][f]
            }
        } else {
            xdec[b * L + ptab[t]] = sigm(act_apply(v[0], act));
        }
    }
}

// per-workgroup fp64 (sum, sum of squares) of x_tx for the power constraint (encoders.py:107-108), fixed-order tree
__global__ __launch_bounds__(256) void gen_stats_kernel(const float* __restrict__ x, size_t n, double* __restrict__ partials) {
    __shared__ double red[512];
    double s = 0.0, q = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const double v = (double)x[i];
        s += v;
        q += v * v;
    }
    red[threadIdx.x] = s;
    red[256 + threadIdx.x] = q;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) { red[threadIdx.x] += red[threadIdx.x + off]; red[256 + threadIdx.x] += red[256 + threadIdx.x + off]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { partials[2 * blockIdx.x] = red[0]; partials[2 * blockIdx.x + 1] = red[256]; }
}

constexpr int kStatsGrid = 256;

inline int gates_of(int cell) { return cell == 0 ? 3 : (cell == 1 ? 4 : 1); }

struct ConvL { size_t wt, bias; int cin, cout, k; };
struct RnnL { size_t wih_t, bih; size_t whh_t[2], bhh[2]; int cin; };     // wih_t / bih: both directions concatenated (2 * G * H outputs)
struct Stack {
    bool rnn = false, dense = false;
    int cell = 0, H = 0, cin0 = 0, nout = 0;
    std::vector<ConvL> conv;
    std::vector<RnnL> rl;
    ConvL lin{};
};

}  // namespace

struct GenericEngine {
    tae_config cfg{};
    std::vector<Stack> enc, dec;
    float* d_w = nullptr;          // every tensor in kernel order (conv / W_ih / W_hh transposed, biases, Linear transposed)
    int cap = 0;
    float *d_xa = nullptr, *d_xb = nullptr, *d_p0 = nullptr, *d_p1 = nullptr, *d_gi = nullptr, *d_o = nullptr;
    double* d_partials = nullptr;
    size_t wide = 0, gi_w = 0;
    bool enc_dense = false, dec_dense = false;
};

// ---- configuration envelope -------------------------------------------------------------------------------------------
bool generic_dec_dense(const tae_config* c) {
    // the reference builds DenseSameShapeConv1d decoders whenever the ENCODER is not the plain CNN (decoders.py:173-176)
    return c->dec_type == 0 && (c->dense != 0 || c->enc_type == 1);
}

bool generic_needed(const tae_config* c) {
    // TAE_FORCE_GENERIC=1 (testing knob): run ANY configuration on the generic kernels - a third, independent implementation to hold
    // against the two MFMA arithmetics (tests/test_gpu_generic.py).  Read per call on purpose (the tests flip it between handles); a
    // handle's own choice is made once, in tae_create, and tae_create re-checks the blob size under the value it sees
    if (const char* e = tae::debug_knob("TAE_FORCE_GENERIC")) if (e[0] == '1') return true;
    const bool big_k = c->enc_kernel_size > 9 || c->dec_kernel_size > 9;
    const bool mid_k = c->enc_kernel_size > 5 || c->dec_kernel_size > 5;
    // channel widths: CNN stacks up to 124 on the MFMA kernels of both arithmetics (instantiated for 32 / 64 / 100 / 124 - the widths whose
    // unpadded LDS rows are conflict-free, U = 4 mod 8; narrower ones run embedded), the recurrent kernels up to 100
    const int cnn_max = 124;
    const int enc_max = c->enc_type == 1 ? 100 : cnn_max, dec_max = c->dec_type == 1 ? 100 : cnn_max;
    if (big_k || c->enc_num_unit > enc_max || c->dec_num_unit > dec_max || c->num_iter_ft > 6) return true;
    // LSTM / vanilla-RNN encoder cells: unit-split f16x2 kernels since r06 (2 layers, checked below); in fp32 here
    if (c->enc_type == 1 && c->enc_rnn != 0 && c->precision == TAE_PREC_F32) return true;
    // LSTM / vanilla-RNN decoder: unit-split f16x2 kernels (turboae_rnn_u.hip, r05) behind the CNN encoder or (r06) the 2-layer GRU
    // encoder; in fp32 here
    if (c->dec_type == 1 && c->dec_rnn != 0 && c->precision == TAE_PREC_F32) return true;
    if (c->enc_type == 1 && (c->enc_num_layer != 2 || c->dec_type != 1)) return true;
    if (c->dense && c->dec_type == 1) return true;
    auto inst = [](int u) { return u == 32 || u == 64 || u == 100; };
    if (c->dense && (!inst(c->enc_num_unit) || !inst(c->dec_num_unit))) return true;     // the f16x2 dense kernels do not embed narrower widths
    if (c->precision == TAE_PREC_F32 && (c->dense || mid_k)) return true;
    return false;
}

const char* generic_check(const tae_config* c) {
    for (int ks : {c->enc_kernel_size, c->dec_kernel_size})
        if (ks < 1 || ks > 63 || (ks & 1) == 0) return "kernel_size must be odd and in 1..63 (SameShapeConv1d pads with kernel_size // 2: an even size changes the length)";
    if (c->enc_num_unit < 1 || c->enc_num_unit > 1024 || c->dec_num_unit < 1 || c->dec_num_unit > 1024) return "enc_num_unit / dec_num_unit must be in 1..1024";
    if (c->num_iter_ft < 1 || c->num_iter_ft > 64) return "num_iter_ft must be in 1..64";
    if (c->enc_rnn < 0 || c->enc_rnn > 2 || c->dec_rnn < 0 || c->dec_rnn > 2) return "enc_rnn / dec_rnn must be 0 (gru), 1 (lstm) or 2 (rnn)";
    if (c->dense && c->enc_type == 1) return "dense = 1 names the dense CNN encoder (enc_type must be 0)";
    return nullptr;
}

static size_t stack_floats(bool rnn, int cell, int H, int cin0, int n_layer, int k, bool dense, int nout) {
    size_t n = 0;
    if (rnn) {
        const size_t G = gates_of(cell);
        for (int l = 0; l < n_layer; ++l) {
            const size_t cin = l == 0 ? cin0 : 2 * H;
            n += 2 * (G * H * cin + G * H * (size_t)H + 2 * G * H);
        }
        return n + (size_t)nout * 2 * H + nout;
    }
    for (int l = 0; l < n_layer; ++l) {
        const size_t cin = l == 0 ? cin0 : (dense ? cin0 + (size_t)l * H : H);
        n += (size_t)H * cin * k + H;
    }
    return n + (size_t)nout * H + nout;
}

size_t generic_num_weights(const tae_config* c) {
    const int F = c->num_iter_ft;
    size_t n = 3 * stack_floats(c->enc_type == 1, c->enc_rnn, c->enc_num_unit, 1, c->enc_num_layer, c->enc_kernel_size, c->dense != 0, 1);
    for (int it = 0; it < c->num_iteration; ++it)
        for (int half = 0; half < 2; ++half) {
            const int nout = (half == 1 && it == c->num_iteration - 1) ? 1 : F;
            n += stack_floats(c->dec_type == 1, c->dec_rnn, c->dec_num_unit, 2 + F, c->dec_type == 1 ? 2 : c->dec_num_layer, c->dec_kernel_size,
                              generic_dec_dense(c), nout);
        }
    return n;
}

// ---- weights: canonical blob (PyTorch layouts, turboae_amd/weights.py) -> kernel order ------------------------------------
static void add_conv(std::vector<float>& out, const float*& src, int cout, int cin, int k, ConvL* L) {
    L->cin = cin; L->cout = cout; L->k = k;
    L->wt = out.size();
    out.resize(out.size() + (size_t)cin * k * cout);
    float* t = out.data() + L->wt;
    for (int co = 0; co < cout; ++co)
        for (int ci = 0; ci < cin; ++ci)
            for (int j = 0; j < k; ++j) t[((size_t)ci * k + j) * cout + co] = src[((size_t)co * cin + ci) * k + j];
    src += (size_t)cout * cin * k;
    L->bias = out.size();
    out.insert(out.end(), src, src + cout);
    src += cout;
}

static Stack build_stack(std::vector<float>& out, const float*& src, bool rnn, int cell, int H, int cin0, int n_layer, int k, bool dense, int nout) {
    Stack S;
    S.rnn = rnn; S.dense = dense; S.cell = cell; S.H = H; S.cin0 = cin0; S.nout = nout;
    if (rnn) {
        const int G = gates_of(cell), GH = G * H;
        for (int l = 0; l < n_layer; ++l) {
            RnnL R;
            R.cin = l == 0 ? cin0 : 2 * H;
            const float* wih[2]; const float* whh[2]; const float* bih[2]; const float* bhh[2];
            for (int d = 0; d < 2; ++d) {          // weight_ih (GH, cin) | weight_hh (GH, H) | bias_ih | bias_hh, forward then _reverse
                wih[d] = src; src += (size_t)GH * R.cin;
                whh[d] = src; src += (size_t)GH * H;
                bih[d] = src; src += GH;
                bhh[d] = src; src += GH;
            }
            R.wih_t = out.size();                  // [ci][dir * GH + r]: one k = 1 "convolution" yields both directions' projections
            out.resize(out.size() + (size_t)R.cin * 2 * GH);
            for (int d = 0; d < 2; ++d)
                for (int r = 0; r < GH; ++r)
                    for (int ci = 0; ci < R.cin; ++ci) out[R.wih_t + (size_t)ci * 2 * GH + d * GH + r] = wih[d][(size_t)r * R.cin + ci];
            R.bih = out.size();
            for (int d = 0; d < 2; ++d) out.insert(out.end(), bih[d], bih[d] + GH);
            for (int d = 0; d < 2; ++d) {
                R.whh_t[d] = out.size();
                out.resize(out.size() + (size_t)H * GH);
                for (int r = 0; r < GH; ++r)
                    for (int kk = 0; kk < H; ++kk) out[R.whh_t[d] + (size_t)kk * GH + r] = whh[d][(size_t)r * H + kk];
                R.bhh[d] = out.size();
                out.insert(out.end(), bhh[d], bhh[d] + GH);
            }
            S.rl.push_back(R);
        }
        add_conv(out, src, nout, 2 * H, 1, &S.lin);
        return S;
    }
    for (int l = 0; l < n_layer; ++l) {
        ConvL C;
        add_conv(out, src, H, l == 0 ? cin0 : (dense ? cin0 + l * H : H), k, &C);
        S.conv.push_back(C);
    }
    add_conv(out, src, nout, H, 1, &S.lin);
    return S;
}

#define GEN_HIP(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e__ = (expr);                                                                        \
        if (e__ != hipSuccess) return fail_msg(TAE_EHIP, (std::string(#expr) + ": " + hipGetErrorString(e__)).c_str()); \
    } while (0)

void generic_destroy(GenericEngine* g) {
    if (!g) return;
    (void)hipFree(g->d_w); (void)hipFree(g->d_xa); (void)hipFree(g->d_xb); (void)hipFree(g->d_p0); (void)hipFree(g->d_p1);
    (void)hipFree(g->d_gi); (void)hipFree(g->d_o); (void)hipFree(g->d_partials);
    delete g;
}

int generic_create(const tae_config* c, const float* weights, size_t n_weights, GenericEngine** out) {
    *out = nullptr;
    if (const char* msg = generic_check(c)) return fail_msg(TAE_EINVAL, msg);
    if (n_weights != generic_num_weights(c)) return fail_msg(TAE_EINVAL, "internal: generic weight count mismatch");
    GenericEngine* g = new GenericEngine();
    g->cfg = *c;
    g->enc_dense = c->dense != 0;
    g->dec_dense = generic_dec_dense(c);
    const int F = c->num_iter_ft;
    std::vector<float> w;
    w.reserve(n_weights + 1024);
    const float* src = weights;
    for (int s = 0; s < 3; ++s)
        g->enc.push_back(build_stack(w, src, c->enc_type == 1, c->enc_rnn, c->enc_num_unit, 1, c->enc_num_layer, c->enc_kernel_size, g->enc_dense, 1));
    for (int it = 0; it < c->num_iteration; ++it)
        for (int half = 0; half < 2; ++half) {
            const int nout = (half == 1 && it == c->num_iteration - 1) ? 1 : F;
            g->dec.push_back(build_stack(w, src, c->dec_type == 1, c->dec_rnn, c->dec_num_unit, 2 + F, c->dec_type == 1 ? 2 : c->dec_num_layer,
                                         c->dec_kernel_size, g->dec_dense, nout));
        }
    if ((size_t)(src - weights) != n_weights) { delete g; return fail_msg(TAE_EINVAL, "internal: generic weight walk mismatch"); }
    // widest activation row / projection row any stack needs
    auto widths = [&](const Stack& S, int n_layer) {
        if (S.rnn) {
            g->wide = std::max(g->wide, (size_t)2 * S.H);
            g->gi_w = std::max(g->gi_w, (size_t)2 * gates_of(S.cell) * S.H);
        } else {
            g->wide = std::max(g->wide, S.dense ? (size_t)S.cin0 + (size_t)n_layer * S.H : (size_t)S.H);
        }
    };
    for (const Stack& S : g->enc) widths(S, c->enc_num_layer);
    for (const Stack& S : g->dec) widths(S, c->dec_num_layer);
    hipError_t e = hipMalloc(&g->d_w, w.size() * sizeof(float));
    if (e == hipSuccess) e = hipMemcpy(g->d_w, w.data(), w.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc(&g->d_partials, kStatsGrid * 2 * sizeof(double));
    if (e != hipSuccess) { generic_destroy(g); return fail_msg(TAE_EHIP, hipGetErrorString(e)); }
    *out = g;
    return TAE_OK;
}

int generic_reserve(GenericEngine* g, int32_t B) {
    if (B <= g->cap) return TAE_OK;
    (void)hipFree(g->d_xa); (void)hipFree(g->d_xb); (void)hipFree(g->d_p0); (void)hipFree(g->d_p1); (void)hipFree(g->d_gi); (void)hipFree(g->d_o);
    g->d_xa = g->d_xb = g->d_p0 = g->d_p1 = g->d_gi = g->d_o = nullptr;
    g->cap = 0;
    const size_t np = (size_t)B * g->cfg.block_len, W = 2 + (size_t)g->cfg.num_iter_ft;
    GEN_HIP(hipMalloc(&g->d_xa, np * W * sizeof(float)));
    GEN_HIP(hipMalloc(&g->d_xb, np * W * sizeof(float)));
    GEN_HIP(hipMalloc(&g->d_p0, np * g->wide * sizeof(float)));
    GEN_HIP(hipMalloc(&g->d_p1, np * g->wide * sizeof(float)));
    if (g->gi_w) GEN_HIP(hipMalloc(&g->d_gi, np * g->gi_w * sizeof(float)));
    GEN_HIP(hipMalloc(&g->d_o, np * W * sizeof(float)));
    g->cap = B;
    return TAE_OK;
}

static hipError_t conv(const GenericEngine* g, const ConvL& C, const float* x, int ldx, float* y, int ldy, int coff, int act, int B, hipStream_t st) {
    const int L = g->cfg.block_len;
    static const bool valu = [] { const char* e = tae::debug_knob("TAE_GEN_CONV"); return e && !strcmp(e, "valu"); }();     // experiments: the r03 vector-ALU kernel
    if (!valu) {
        const size_t np = (size_t)B * L;
        const size_t lds = ((size_t)(kMP + C.k - 1) * kMKP + (size_t)std::min(C.k, kMJ) * kMK * kMWP) * sizeof(float);
        if (lds > 65536) {             // per device and cheap next to a launch of this size: no process-wide cache of "already set"
            const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gen_conv_mfma_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
        }
        const unsigned ctiles = (C.cout + kMC - 1) / kMC, ntiles = (unsigned)((np + kMP - 1) / kMP) * ctiles;
        const dim3 grid((ntiles + 7) / 8 * 8);
        static const int xcd = [] { const char* e = tae::debug_knob("TAE_GEN_XCD"); return !(e && e[0] == '0'); }();     // experiments: 0 = tiles in launch order
        static const bool no_proj = [] { const char* e = tae::debug_knob("TAE_GEN_PROJ"); return e && e[0] == '0'; }();     // experiments: k = 1 on the general kernel
        if (C.k == 1 && !no_proj) {
            hipLaunchKernelGGL(gen_proj_mfma_kernel, grid, dim3(512), 0, st, x, ldx, C.cin, g->d_w + C.wt, g->d_w + C.bias, y, ldy, coff, C.cout, np, act, ctiles, ntiles, xcd);
            return hipGetLastError();
        }
        hipLaunchKernelGGL(gen_conv_mfma_kernel, grid, dim3(512), lds, st, x, ldx, C.cin, g->d_w + C.wt, g->d_w + C.bias, y, ldy, coff, C.cout, C.k, L, np, act, ctiles, ntiles, xcd);
        return hipGetLastError();
    }
    const size_t lds = (size_t)(kConvPos + C.k - 1) * kConvCi * sizeof(float);
    // the block index rides in grid.y, which HIP caps at 65535: larger batches go out in slices
    for (int b0 = 0; b0 < B; b0 += 65535) {
        const int nb = B - b0 < 65535 ? B - b0 : 65535;
        const dim3 grid((L + kConvPos - 1) / kConvPos, nb, (C.cout + kConvCh - 1) / kConvCh);
        hipLaunchKernelGGL(gen_conv_kernel, grid, dim3(256), lds, st, x + (size_t)b0 * L * ldx, ldx, C.cin, g->d_w + C.wt, g->d_w + C.bias,
                           y + (size_t)b0 * L * ldy, ldy, coff, C.cout, C.k, L, act);
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

// one stack: x (B, L, cin0) -> o (B, L, nout) = Linear(stack(x)); returns through g->d_o
static int run_stack(GenericEngine* g, const Stack& S, const float* x, int B, hipStream_t st) {
    const int L = g->cfg.block_len;
    const size_t np = (size_t)B * L;
    const float* feat = nullptr;
    int ldf = 0;
    if (S.rnn) {
        const int G = gates_of(S.cell), GH = G * S.H;
        if (S.H > 1024) return fail_msg(TAE_EINVAL, "recurrent units above 1024 are not supported");
        const float* in = x;
        int ldin = S.cin0;
        float* bufs[2] = {g->d_p0, g->d_p1};
        for (size_t l = 0; l < S.rl.size(); ++l) {
            const RnnL& R = S.rl[l];
            static const bool rnn_valu = [] { const char* e = tae::debug_knob("TAE_GEN_RNN"); return e && !strcmp(e, "valu"); }();     // experiments: the vector-ALU kernel
            static const bool no_fuse = [] { const char* e = tae::debug_knob("TAE_GEN_RNN_FUSE"); return e && e[0] == '0'; }();         // experiments: GI through HBM for every layer
            const bool mfma = S.H <= 128 && (S.H & 3) == 0 && !rnn_valu, fused = mfma && R.cin <= 8 && !no_fuse;
            float* y = bufs[l & 1];
            if (!fused) {
                ConvL P{R.wih_t, R.bih, R.cin, 2 * GH, 1};
                GEN_HIP(conv(g, P, in, ldin, g->d_gi, 2 * GH, 0, 0, B, st));
            }
            if (mfma) {
                const RnnArgs a{g->d_gi, in, ldin, R.cin, g->d_w + R.wih_t, g->d_w + R.bih, g->d_w + R.whh_t[0], g->d_w + R.whh_t[1],
                                g->d_w + R.bhh[0], g->d_w + R.bhh[1], y, S.H, L, B};
                GEN_HIP(S.cell == 0 ? launch_rnn_mfma<3>(a, fused, st) : S.cell == 1 ? launch_rnn_mfma<4>(a, fused, st) : launch_rnn_mfma<1>(a, fused, st));
                in = y;
                ldin = 2 * S.H;
                continue;
            }
            const int threads = (S.H + 63) / 64 * 64;
            // blocks per workgroup: share the recurrent weights where the batch still fills the chip with workgroups
            static const int nb_env = [] { const char* e = tae::debug_knob("TAE_GEN_RNN_NB"); return e ? atoi(e) : 0; }();     // experiments: read once
            const int nb = nb_env ? nb_env : (B >= 4096 ? 8 : (B >= 1024 ? 4 : 1));
            if (nb == 8)
                hipLaunchKernelGGL(gen_rnn_kernel<8>, dim3((B + 7) / 8, 2), dim3(threads), 8 * S.H * sizeof(float), st, S.cell, g->d_gi, g->d_w + R.whh_t[0],
                                   g->d_w + R.whh_t[1], g->d_w + R.bhh[0], g->d_w + R.bhh[1], y, S.H, L, B);
            else if (nb == 4)
                hipLaunchKernelGGL(gen_rnn_kernel<4>, dim3((B + 3) / 4, 2), dim3(threads), 4 * S.H * sizeof(float), st, S.cell, g->d_gi, g->d_w + R.whh_t[0],
                                   g->d_w + R.whh_t[1], g->d_w + R.bhh[0], g->d_w + R.bhh[1], y, S.H, L, B);
            else
                hipLaunchKernelGGL(gen_rnn_kernel<1>, dim3(B, 2), dim3(threads), S.H * sizeof(float), st, S.cell, g->d_gi, g->d_w + R.whh_t[0],
                                   g->d_w + R.whh_t[1], g->d_w + R.bhh[0], g->d_w + R.bhh[1], y, S.H, L, B);
            GEN_HIP(hipGetLastError());
            in = y;
            ldin = 2 * S.H;
        }
        feat = in;
        ldf = 2 * S.H;
    } else if (S.dense) {
        // layer l convolves cat(inputs, out_0 .. out_{l-1}) = the first cin0 + l * H channels of one wide row (cnn_utils.py:59-62)
        const int wtot = S.cin0 + (int)S.conv.size() * S.H;
        const int grid = (int)std::min<size_t>((np * S.cin0 + 255) / 256, 4096);
        hipLaunchKernelGGL(gen_copy_rows_kernel, dim3(grid), dim3(256), 0, st, x, S.cin0, g->d_p0, wtot, S.cin0, np);
        GEN_HIP(hipGetLastError());
        for (size_t l = 0; l < S.conv.size(); ++l) GEN_HIP(conv(g, S.conv[l], g->d_p0, wtot, g->d_p0, wtot, S.cin0 + (int)l * S.H, 1, B, st));
        feat = g->d_p0 + S.cin0 + (S.conv.size() - 1) * (size_t)S.H;
        ldf = wtot;
    } else {
        float* bufs[2] = {g->d_p0, g->d_p1};
        const float* in = x;
        int ldin = S.cin0;
        for (size_t l = 0; l < S.conv.size(); ++l) {
            float* y = bufs[l & 1];
            GEN_HIP(conv(g, S.conv[l], in, ldin, y, S.H, 0, 1, B, st));
            in = y;
            ldin = S.H;
        }
        feat = in;
        ldf = S.H;
    }
    GEN_HIP(conv(g, S.lin, feat, ldf, g->d_o, S.nout, 0, 0, B, st));
    return TAE_OK;
}

int generic_encode(GenericEngine* g, const float* u, float* xtx, double* stats, const int32_t* perm, int32_t B, hipStream_t st) {
    if (B > g->cap) return fail_msg(TAE_ESTATE, "batch exceeds reserved workspace");
    const int L = g->cfg.block_len;
    const size_t np = (size_t)B * L;
    const int grid = (int)std::min<size_t>((np + 255) / 256, 4096);
    for (int s = 0; s < 3; ++s) {
        hipLaunchKernelGGL(gen_prep_enc_kernel, dim3(grid), dim3(256), 0, st, u, perm, g->d_xa, (size_t)B, L, s == 2 ? 1 : 0, g->cfg.enc_type == 1 ? 0 : 1);
        GEN_HIP(hipGetLastError());
        const int rc = run_stack(g, g->enc[s], g->d_xa, B, st);
        if (rc != TAE_OK) return rc;
        hipLaunchKernelGGL(gen_head_kernel, dim3(grid), dim3(256), 0, st, g->d_o, 1, (const float*)nullptr, (float*)nullptr, 0, (float*)nullptr,
                           perm, (size_t)B, L, 1, 0, 0, g->cfg.enc_act, s, xtx, (float*)nullptr);
        GEN_HIP(hipGetLastError());
    }
    hipLaunchKernelGGL(gen_stats_kernel, dim3(kStatsGrid), dim3(256), 0, st, xtx, np * 3, g->d_partials);
    GEN_HIP(hipGetLastError());
    GEN_HIP(launch_reduce_partials(g->d_partials, kStatsGrid, (double)np * 3.0, stats, st));
    return TAE_OK;
}

int generic_decode(GenericEngine* g, const float* rx, float* xdec, const int32_t* perm, const int32_t* inv, int32_t B, hipStream_t st, float* tap_out) {
    if (B > g->cap) return fail_msg(TAE_ESTATE, "batch exceeds reserved workspace");
    const int L = g->cfg.block_len, F = g->cfg.num_iter_ft, W = 2 + F;
    const size_t np = (size_t)B * L;
    const int grid = (int)std::min<size_t>((np + 255) / 256, 4096);
    hipLaunchKernelGGL(gen_prep_dec_kernel, dim3(grid), dim3(256), 0, st, rx, perm, g->d_xa, g->d_xb, (size_t)B, L, W);
    GEN_HIP(hipGetLastError());
    const int act = g->cfg.dec_type == 1 ? g->cfg.dec_act : TAE_ACT_LINEAR;      // DEC_LargeCNN has no dec_act
    const int ns = 2 * g->cfg.num_iteration;
    for (int s = 0; s < ns; ++s) {
        const bool odd = (s & 1) != 0, last = s == ns - 1;
        float* xin = odd ? g->d_xb : g->d_xa;
        const int rc = run_stack(g, g->dec[s], xin, B, st);
        if (rc != TAE_OK) return rc;
        hipLaunchKernelGGL(gen_head_kernel, dim3(grid), dim3(256), 0, st, g->d_o, g->dec[s].nout, xin, odd ? g->d_xa : g->d_xb, W, xdec,
                           odd ? perm : inv, (size_t)B, L, F, g->cfg.extrinsic, last ? 1 : 0, act, -1, (float*)nullptr,
                           (tap_out && !last) ? tap_out + (size_t)s * np * F : (float*)nullptr);
        GEN_HIP(hipGetLastError());
    }
    return TAE_OK;
}

}  // namespace tae
