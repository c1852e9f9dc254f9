// DeepTurbo GRU decoder (BASELINE configs[4]): DEC_LargeRNN.forward (decoders.py:84-149) with
// dec_rnn = 'gru' (get_args.py:80), dec_act = 'linear' (get_args.py:101), dropout = 0 at eval:
// the same turbo iteration as DEC_LargeCNN with every conv stack replaced by
// torch.nn.GRU(2+F, H, num_layers=2, bidirectional=True, batch_first=True) + Linear(2H -> F | 1).
//
// PyTorch GRU cell (gate order r, z, n; the arithmetic lives in ATen, third-party to the reference):
//   r = sigmoid(W_ir x + b_ir + W_hr h + b_hr)      z = sigmoid(W_iz x + b_iz + W_hz h + b_hz)
//   n = tanh(W_in x + b_in + r * (W_hn h + b_hn))   h' = (1 - z) * n + z * h          h_0 = 0
//
// MI355X mapping (all fp32, v_mfma_f32_16x16x4_f32):
//   gru_prep  : received (B,L,3) -> XA / XB input panels in HBM (B,L,8), as in the CNN kernels
//   gru_rec   : one WAVE = 16 blocks x one direction of one layer, strictly sequential over t; the gate
//               pre-activations of a step are a [304 x 100] x [100 x 16] MFMA product whose A fragments
//               (W_hh) stay in LDS for the whole launch and whose B operand (h_t) never leaves the
//               registers (see the kernel).  Layer 0 adds its K = 2+F input projection as two more k-steps;
//               layer 1 reads the projections computed by gru_proj.
//   gru_proj  : layer-1 input projections for all positions, both directions, as an MFMA GEMM
//               GI = W_ih1 * Y0 + b (K = 2H) on the CNN kernels' K-loop (weights streamed from L2).
//   gru_head  : Linear(2H -> F|1) + extrinsic subtraction + (de)interleave scatter into the other
//               panel, or sigmoid + deinterleave for the last half-iteration (decoders.py:145-147).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "turboae_internal.hpp"
#include "turboae_device.hpp"

namespace tae {

using u32x4v = __attribute__((ext_vector_type(4))) uint32_t;

constexpr int kGruH = 100;          // hidden units per direction (dec_num_unit)
constexpr int kXWg = 8;             // floats per row of the XA / XB panels

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// received (B,L,3) -> XA = [r_sys, r_par1, 0...], XB = [r_sys_int, r_par2, 0...]  (decoders.py:87-93)
__global__ void gru_prep_kernel(const float* __restrict__ rx, const int32_t* __restrict__ perm, float* __restrict__ XA,
                                float* __restrict__ XB, int B, int L) {
    const size_t n = (size_t)B * L;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t b = i / L;
        const int t = (int)(i - b * L);
        const float* r = rx + i * 3;
        f32x4 z = {0.f, 0.f, 0.f, 0.f};
        f32x4 a = {r[0], r[1], 0.f, 0.f};
        f32x4 c = {rx[(b * L + perm[t]) * 3], r[2], 0.f, 0.f};
        reinterpret_cast<f32x4*>(XA + i * kXWg)[0] = a;
        reinterpret_cast<f32x4*>(XA + i * kXWg)[1] = z;
        reinterpret_cast<f32x4*>(XB + i * kXWg)[0] = c;
        reinterpret_cast<f32x4*>(XB + i * kXWg)[1] = z;
    }
}

// ENC_interRNN input panel (encoders.py:283-292): column 0 = the bit (raw 0/1 - no 2u - 1 in this encoder), the third
// branch sees the interleaved bits; columns 1..7 zero
__global__ void gru_prep_enc_kernel(const float* __restrict__ u, const int32_t* __restrict__ perm, float* __restrict__ X, int B, int L, int interleaved) {
    const size_t n = (size_t)B * L;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t b = i / L;
        const int t = (int)(i - b * L);
        const float v = u[b * L + (interleaved ? perm[t] : t)];
        reinterpret_cast<f32x4*>(X + i * kXWg)[0] = f32x4{v, 0.f, 0.f, 0.f};
        reinterpret_cast<f32x4*>(X + i * kXWg)[1] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
}

// ---- recurrent kernel -------------------------------------------------------------------------------
// Gate rows of one direction are arranged in kRT = 19 MFMA row tiles: tile 3*ut + g (ut < 6, g = r, z, n)
// holds gate g of units 16*ut .. 16*ut + 15, and the remainder tile 18 holds, in row 4*qq + i, gate i of
// unit 96 + qq (i = 3: padding; layer 0 parks the n-gate INPUT projection there).  One wave owns 16
// blocks (the N dimension) for the whole sequence.  In the 16x16x4 D layout lane (n, q) then holds
// r, z and n of units 16*ut + 4*q + i (i = register index) of block n - the gate arithmetic is purely
// per-lane - and the new h values sit in exactly the B-operand layout of the NEXT step if k-step
// s = 4*ut + i is defined to contract over the units {16*ut + 4*kq + i : kq = 0..3} (the contraction
// order is free; W_hh is packed to match).  So h never leaves the registers, waves never synchronise,
// and only the A fragments (W_hh, 126 KB per direction) are read from LDS, where they stay for the launch.
constexpr int kRT = 19;                          // gate-row tiles per direction
constexpr int kKP = 13;                          // k-step pairs: 24 k-steps over units 0..95 + 1 over units 96..99
constexpr int kRecFragF = kRT * kKP * 128;       // floats of recurrent A fragments per direction
constexpr int kXFragF = kRT * 128;               // layer 0: A fragments of the K = 8 input projection
constexpr int kBias0F = 25 * 16;                 // layer 0: accumulator-init rows (19 tiles + 6 n-input tiles)
constexpr int kBias1F = 7 * 16;                  // layer 1: b_hn rows (6 unit tiles + remainder)
constexpr int kGiRowF = 2 * kRT * 16;            // floats per position of the projection buffer GI

__device__ __forceinline__ float sigm_fast(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896341f * x));
}
// 1 - 2 / (1 + e^{2x}): absolute error ~1e-7 everywhere (saturates cleanly at +-1)
__device__ __forceinline__ float tanh_fast(float x) {
    return fmaf(-2.0f, __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.88539008177792681f * x)), 1.0f);
}

using lds_f2 = const f32x2 __attribute__((address_space(3)));
using lds_f4 = const f32x4 __attribute__((address_space(3)));

// {ND LDS reads spread through NM MFMAs}
template <int ND, int NM>
__device__ __forceinline__ void spread_ds() {
    constexpr int G = NM / ND;
#pragma unroll
    for (int i = 0; i < ND; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, G, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x008, NM - ND * G, 0);
}

// fragment f = tile * 13 + pair lives at byte f * 512: two bases keep every offset inside the 16-bit DS immediate
template <int T0, int NT>
__device__ __forceinline__ void rec_load(f32x2 (&a)[NT], lds_cptr frag0, lds_cptr frag1, int kp) {
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        const int f = (T0 + i) * kKP + kp;
        a[i] = *reinterpret_cast<lds_f2*>((f < 128 ? frag0 : frag1) + (f & 127) * 512);
    }
}

template <int T0, int NT, bool BOTH>
__device__ __forceinline__ void rec_mma(f32x4 (&acc)[kRT], const f32x2 (&a)[NT], float bx, float by) {
#pragma unroll
    for (int i = 0; i < NT; ++i) acc[T0 + i] = mfma16x16x4(a[i].x, bx, acc[T0 + i]);
    if (BOTH) {
#pragma unroll
        for (int i = 0; i < NT; ++i) acc[T0 + i] = mfma16x16x4(a[i].y, by, acc[T0 + i]);
    }
}

template <bool LAYER0>
__global__ __launch_bounds__(512) void gru_rec_kernel(GruRecParams P) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, n = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // wave-uniform: the buffer resources below live in SGPRs
    const int dir = blockIdx.y, L = P.L;
    {
        const f32x4* src = reinterpret_cast<const f32x4*>(P.w + (size_t)dir * P.w_dir_stride);
        constexpr int NV = (kRecFragF + (LAYER0 ? kXFragF + kBias0F : kBias1F)) / 4;
        for (int i = tid; i < NV; i += (int)blockDim.x) reinterpret_cast<f32x4*>(smem)[i] = src[i];
    }
    __syncthreads();
    const int b0 = (blockIdx.x * (int)(blockDim.x >> 6) + wave) * 16;
    if (b0 >= P.B) return;                                  // no barrier below: waves are independent
    const int nb = min(16, P.B - b0);
    const bool valid = n < nb;
    const int nc = valid ? n : nb - 1;                      // tail lanes recompute the last block, stores are dropped
    const lds_cptr frag = (lds_cptr)smem + lane * 8;
    lds_cptr frag1 = frag + 128 * 512;
    asm volatile("" : "+v"(frag1));                         // keep the second base in its own register
    const lds_cptr xfrag = (lds_cptr)smem + kRecFragF * 4 + lane * 8;
    const lds_cptr bias = (lds_cptr)smem + (kRecFragF + (LAYER0 ? kXFragF : 0)) * 4 + q * 16;

    // buffer resources over this wave's 16 blocks: wave-uniform SGPR offset = position, lane VGPR offset = block
    const __amdgpu_buffer_rsrc_t rs_in = LAYER0
        ? __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(P.x + (size_t)b0 * L * kXWg), 0, nb * L * kXWg * 4, 0x00020000)
        : __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(P.gi + (size_t)b0 * L * kGiRowF), 0, nb * L * kGiRowF * 4, 0x00020000);
    const uint32_t v_in = LAYER0 ? (uint32_t)(nc * L * kXWg * 4 + q * 4) : (uint32_t)(nc * L * kGiRowF * 4 + q * 16);
    const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(P.y + (size_t)b0 * L * 2 * kGruH, 0, nb * L * 2 * kGruH * 4, 0x00020000);
    const uint32_t v_y = valid ? (uint32_t)(n * L * 2 * kGruH * 4 + dir * kGruH * 4 + q * 16) : 0x80000000u;
    const uint32_t v_yr = valid ? (uint32_t)(n * L * 2 * kGruH * 4 + dir * kGruH * 4 + 96 * 4 + q * 4) : 0x80000000u;

    f32x4 h[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) h[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float hr = 0.0f;
    float xa = 0.f, xb = 0.f;
    if (LAYER0) {
        const uint32_t so = (uint32_t)(dir ? L - 1 : 0) * kXWg * 4;
        xa = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_in, v_in, so, 0));
        xb = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_in, v_in + 16, so, 0));
    }
#pragma unroll 1
    for (int s = 0; s < L; ++s) {
        const int t = dir ? L - 1 - s : s;
        f32x4 acc[kRT], ani[6], g[kRT];
        float xna = 0.f, xnb = 0.f;
        if (LAYER0) {
            // next step's inputs (clamped at the end), this step's accumulator init = biases
            const int tn = dir ? (t > 0 ? t - 1 : 0) : (t + 1 < L ? t + 1 : t);
            const uint32_t so = (uint32_t)tn * kXWg * 4;
            xna = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_in, v_in, so, 0));
            xnb = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_in, v_in + 16, so, 0));
#pragma unroll
            for (int T = 0; T < kRT; ++T) acc[T] = *reinterpret_cast<lds_f4*>(bias + T * 64);
#pragma unroll
            for (int u = 0; u < 6; ++u) ani[u] = *reinterpret_cast<lds_f4*>(bias + (kRT + u) * 64);
            // K = 8 input projection: k-step x <- panel columns 0..3, k-step y <- columns 4..7
#pragma unroll
            for (int T = 0; T < kRT; ++T) {
                const f32x2 a = *reinterpret_cast<lds_f2*>(xfrag + T * 512);
                if (T < 18 && T % 3 == 2) {
                    ani[T / 3] = mfma16x16x4(a.x, xa, ani[T / 3]);
                    ani[T / 3] = mfma16x16x4(a.y, xb, ani[T / 3]);
                } else {
                    acc[T] = mfma16x16x4(a.x, xa, acc[T]);
                    acc[T] = mfma16x16x4(a.y, xb, acc[T]);
                }
            }
        } else {
            // this step's input projections: in flight during the MFMAs, consumed by the gate arithmetic
            const uint32_t so = (uint32_t)t * kGiRowF * 4 + (uint32_t)dir * (kRT * 64);
#pragma unroll
            for (int T = 0; T < kRT; ++T)
                g[T] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_in, v_in + T * 64, so, 0));
#pragma unroll
            for (int T = 0; T < kRT; ++T)
                acc[T] = (T == 18 || T % 3 == 2) ? *reinterpret_cast<lds_f4*>(bias + (T == 18 ? 6 : T / 3) * 64) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        // W_hh * h: 13 k-step pairs x 19 tiles, A fragments double-buffered in two tile groups
        f32x2 fa[10], fb[9];
        rec_load<0, 10>(fa, frag, frag1, 0);
#pragma unroll
        for (int kp = 0; kp < kKP; ++kp) {
            const float bx = kp < 12 ? h[kp / 2][2 * (kp & 1)] : hr;
            const float by = kp < 12 ? h[kp / 2][2 * (kp & 1) + 1] : 0.0f;
            rec_load<10, 9>(fb, frag, frag1, kp);
            if (kp < 12) rec_mma<0, 10, true>(acc, fa, bx, by); else rec_mma<0, 10, false>(acc, fa, bx, by);
            if (kp < 12) spread_ds<9, 20>(); else spread_ds<9, 10>();
            if (kp + 1 < kKP) rec_load<0, 10>(fa, frag, frag1, kp + 1);
            if (kp < 12) rec_mma<10, 9, true>(acc, fb, bx, by); else rec_mma<10, 9, false>(acc, fb, bx, by);
            if (kp + 1 < kKP) { if (kp < 12) spread_ds<10, 18>(); else spread_ds<9, 9>(); }
        }
        // gates (PyTorch order r, z, n) and state update, all in this lane's registers
#pragma unroll
        for (int u = 0; u < 6; ++u) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float r = sigm_fast(LAYER0 ? acc[3 * u][i] : acc[3 * u][i] + g[3 * u][i]);
                const float z = sigm_fast(LAYER0 ? acc[3 * u + 1][i] : acc[3 * u + 1][i] + g[3 * u + 1][i]);
                const float nn = tanh_fast(fmaf(r, acc[3 * u + 2][i], LAYER0 ? ani[u][i] : g[3 * u + 2][i]));
                h[u][i] = fmaf(z, h[u][i] - nn, nn);
            }
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v, h[u]), rs_y, v_y + u * 64, (uint32_t)t * (2 * kGruH * 4), 0);
        }
        {
            const float r = sigm_fast(LAYER0 ? acc[18][0] : acc[18][0] + g[18][0]);
            const float z = sigm_fast(LAYER0 ? acc[18][1] : acc[18][1] + g[18][1]);
            const float nn = tanh_fast(fmaf(r, acc[18][2], LAYER0 ? acc[18][3] : g[18][2]));
            hr = fmaf(z, hr - nn, nn);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, hr), rs_y, v_yr, (uint32_t)t * (2 * kGruH * 4), 0);
        }
        if (LAYER0) { xa = xna; xb = xnb; }
    }
}

// ---- layer-1 input projections as an MFMA GEMM ---------------------------------------------------------
// GI[p][dir][tile][16] = bias + W_ih1 (2 x 304 rows in the recurrent kernel's tile order) * Y0[p][0..199].
// Workgroup = 4 waves, 64 positions staged in LDS (rows of 200 floats, 51 KB, so that 2-3 workgroups share a
// CU and one's panel load overlaps the others' MFMAs); wave (g, hh) owns position tiles 2g, 2g+1 and row
// tiles [0, 10) or [10, 19), one direction per pass; the K loop is the CNN kernels' conv_accumulate
// (weights streamed from L2 in A-fragment order, B fragments = one ds_read_b64 per position tile).
constexpr int kProjPos = 64;
constexpr int kProjThreads = 256;
constexpr int kProjLds = kProjPos * 2 * kGruH * 4 + 256;

template <int C0, int NC>
__device__ __forceinline__ void proj_half(const GruProjParams& P, const char* smem, int g, int lane, size_t p0) {
    const int n = lane & 15, kq = lane >> 4;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(P.w), 0, 2 * 25 * kRT * 512 + 2 * kRT * 64, 0x00020000);
    const uint32_t voff = (uint32_t)lane * 16u;
    const uint32_t baddr[2] = {(uint32_t)(((g * 2 + 0) * 16 + n) * (2 * kGruH * 4) + 8 * kq),
                               (uint32_t)(((g * 2 + 1) * 16 + n) * (2 * kGruH * 4) + 8 * kq)};
    const float* bias = P.w + 2 * 25 * kRT * 128;
#pragma unroll 1
    for (int dir = 0; dir < 2; ++dir) {
        const uint32_t soff = (uint32_t)dir * (25 * kRT * 512);
        Ops<NC, 2> o0;
        load_w<kRT, C0, NC, 2>(o0, rsrc, voff, soff);
        f32x4 acc[2][NC], accS[2];
#pragma unroll
        for (int ct = 0; ct < NC; ++ct) {
            const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + dir * (kRT * 16) + (C0 + ct) * 16 + 4 * kq);
            acc[0][ct] = bv;
            acc[1][ct] = bv;
        }
        conv_accumulate<kRT, C0, NC, 2, 25, 0>(acc, accS, o0, rsrc, voff, soff, 0u, smem, baddr, 0u);
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const size_t pos = p0 + (g * 2 + p) * 16 + n;
            if (pos < P.npos && !(TAE_X & 32)) {
                float* dst = P.gi + pos * kGiRowF + dir * (kRT * 16) + C0 * 16 + 4 * kq;
#pragma unroll
                for (int ct = 0; ct < NC; ++ct) *reinterpret_cast<f32x4*>(dst + ct * 16) = acc[p][ct];
            }
        }
    }
}

__global__ __launch_bounds__(kProjThreads) void gru_proj_kernel(GruProjParams P) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t p0 = (size_t)blockIdx.x * kProjPos;
    const int np = (int)min((size_t)kProjPos, P.npos - p0);
    {
        const f32x4* src = reinterpret_cast<const f32x4*>(P.yin + p0 * 2 * kGruH);
        const int nv = np * (2 * kGruH / 4);
        for (int i = tid; i < kProjLds / 16; i += kProjThreads)
            reinterpret_cast<f32x4*>(smem)[i] = i < nv ? src[i] : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    __syncthreads();
    // waves 0, 1 / 2, 3 of every workgroup land on the same SIMD pairs: alternate which pair gets the larger half
    if (((wave >> 1) ^ (int)(blockIdx.x & 1)) == 0) proj_half<0, 10>(P, smem, wave & 1, lane, p0);
    else proj_half<10, 9>(P, smem, wave & 1, lane, p0);
}

// Linear(2H -> F|1) + dec_act (linear) + extrinsic + (de)interleave (decoders.py:104-147).
// HBM-bound (one pass over Y1): a wave takes 16 positions at a time, its B operand straight from global memory
// (lane (n, kq) loads Y1[pos n][16c + 4kq .. +3] = its share of 4 k-steps), the <= 8 output rows of the Linear
// as A fragments held in registers, four accumulation chains (k-step mod 4) summed at the end.
constexpr int kHeadWaves = 4;
__global__ __launch_bounds__(64 * kHeadWaves) void gru_head_kernel(GruHeadParams P) {
    constexpr int K = 2 * kGruH;
    const int lane = threadIdx.x & 63, n = lane & 15, kq = lane >> 4;
    float a[50];
#pragma unroll
    for (int c = 0; c < 12; ++c)
#pragma unroll
        for (int j = 0; j < 4; ++j) a[4 * c + j] = n < P.nout ? P.w[n * K + 16 * c + 4 * kq + j] : 0.0f;
    a[48] = n < P.nout ? P.w[n * K + 192 + 2 * kq] : 0.0f;
    a[49] = n < P.nout ? P.w[n * K + 192 + 2 * kq + 1] : 0.0f;
    f32x4 bias;
#pragma unroll
    for (int i = 0; i < 4; ++i) bias[i] = (4 * kq + i) < P.nout ? P.b[4 * kq + i] : 0.0f;
    const size_t ntile = (P.npos + 15) / 16;
    double esum = 0.0, esq = 0.0;
    for (size_t tile = (size_t)blockIdx.x * kHeadWaves + (threadIdx.x >> 6); tile < ntile; tile += (size_t)gridDim.x * kHeadWaves) {
        const size_t posy = tile * 16 + n;      // row of Y1
        const size_t pc = posy < P.npos ? posy : P.npos - 1;
        const float* y = P.y + pc * K + 4 * kq;
        f32x4 v[12];
#pragma unroll
        for (int c = 0; c < 12; ++c) v[c] = *reinterpret_cast<const f32x4*>(y + 16 * c);
        const f32x2 vl = *reinterpret_cast<const f32x2*>(P.y + pc * K + 192 + 2 * kq);
        f32x4 acc[4] = {bias, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int c = 0; c < 12; ++c)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = mfma16x16x4(a[4 * c + j], v[c][j], acc[j]);
        acc[0] = mfma16x16x4(a[48], vl.x, acc[0]);
        acc[1] = mfma16x16x4(a[49], vl.y, acc[1]);
        const f32x4 o = (acc[0] + acc[1]) + (acc[2] + acc[3]);       // rows 4*kq + i = outputs f of position n
        bool live = posy < P.npos && kq < 2;
        size_t b = 0;
        int t = 0;
        if (P.grouped) {            // f16x2 path: pos' = ((b / 16) * L + t) * 16 + b % 16
            const size_t row = posy >> 4;
            const size_t grp = row / P.L;
            t = (int)(row - grp * P.L);
            b = grp * 16 + (posy & 15);
            live = live && b < (size_t)P.B;
        } else {
            b = posy / P.L;
            t = (int)(posy - b * P.L);
        }
        if (!live) continue;
        const size_t pos = b * P.L + t;      // (block, t) order of the X panels
        if (P.enc_stack >= 0) {
            if (kq == 0) {               // ENC_interRNN: x = enc_act(Linear(2H -> 1)) (encoders.py:284,287,292)
                float v = o[0];
                v = P.act == 0 ? (v > 0.0f ? v : expm1f(v)) : act_apply(v, P.act);
                P.xtx[pos * 3 + P.enc_stack] = v;
                esum += (double)v;
                esq += (double)v * (double)v;
            }
        } else if (!P.last) {
            const float* xc = P.xcur + pos * kXWg + 2;
            float* xn = P.xnext + (b * P.L + P.ptab[t]) * kXWg + 2;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int f = 4 * kq + i;
                if (f < P.F) {
                    xn[f] = act_apply(o[i], P.act) - (P.extrinsic ? xc[f] : 0.0f);      // dec_act, then extrinsic (decoders.py:103-106)
                    if (P.tap) P.tap[pos * P.F + f] = xn[f];
                }
            }
        } else if (kq == 0) {
            P.xdec[b * P.L + P.ptab[t]] = sigmoidf_(act_apply(o[0], P.act));      // sigmoid(deinterleave(dec_act(x_plr))), decoders.py:143-147
        }
    }
    if (P.enc_stack >= 0) {
        // per-workgroup partial sums for power_constraint (encoders.py:107-108), fixed-order tree
        __shared__ double red[2 * 64 * kHeadWaves];
        const int tid = threadIdx.x;
        red[tid] = esum;
        red[64 * kHeadWaves + tid] = esq;
        __syncthreads();
        for (int off = 32 * kHeadWaves; off > 0; off >>= 1) {
            if (tid < off) { red[tid] += red[tid + off]; red[64 * kHeadWaves + tid] += red[64 * kHeadWaves + tid + off]; }
            __syncthreads();
        }
        if (tid == 0) { P.partials[2 * blockIdx.x] = red[0]; P.partials[2 * blockIdx.x + 1] = red[64 * kHeadWaves]; }
    }
}

hipError_t launch_gru_prep(const float* rx, const int32_t* perm, float* XA, float* XB, int B, int L, hipStream_t st) {
    const size_t n = (size_t)B * L;
    const int grid = (int)std::min<size_t>((n + 255) / 256, 4096);
    hipLaunchKernelGGL(gru_prep_kernel, dim3(grid), dim3(256), 0, st, rx, perm, XA, XB, B, L);
    return hipGetLastError();
}

int gru_rec_lds_bytes(bool layer0) { return (kRecFragF + (layer0 ? kXFragF + kBias0F : kBias1F)) * 4; }

hipError_t launch_gru_rec(bool layer0, const GruRecParams& P, hipStream_t st) {
    const int lds = gru_rec_lds_bytes(layer0);
    // 8 waves (128 blocks) per workgroup when that still fills the 256 CUs (one workgroup per CU: W_hh takes
    // most of the LDS); fewer waves for small batches - a step costs the same whether a SIMD carries 1 or 2 waves' MFMAs
    int nw = 8;
    while (nw > 1 && 2 * ((P.B + 16 * nw - 1) / (16 * nw)) < 256) nw >>= 1;
    const dim3 grid((P.B + 16 * nw - 1) / (16 * nw), 2);
    const void* fn = layer0 ? reinterpret_cast<const void*>(gru_rec_kernel<true>) : reinterpret_cast<const void*>(gru_rec_kernel<false>);
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return e;
    if (layer0) hipLaunchKernelGGL(gru_rec_kernel<true>, grid, dim3(64 * nw), lds, st, P);
    else hipLaunchKernelGGL(gru_rec_kernel<false>, grid, dim3(64 * nw), lds, st, P);
    return hipGetLastError();
}

hipError_t launch_gru_proj(const GruProjParams& P, hipStream_t st) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gru_proj_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kProjLds);
    if (e != hipSuccess) return e;
    const dim3 grid((unsigned)((P.npos + kProjPos - 1) / kProjPos));
    hipLaunchKernelGGL(gru_proj_kernel, grid, dim3(kProjThreads), kProjLds, st, P);
    return hipGetLastError();
}

int gru_head_grid(size_t npos) {
    const size_t ntile = (npos + 15) / 16;
    return (int)std::min<size_t>((ntile + kHeadWaves - 1) / kHeadWaves, 256 * 8);
}

hipError_t launch_gru_head(const GruHeadParams& P, hipStream_t st) {
    hipLaunchKernelGGL(gru_head_kernel, dim3(gru_head_grid(P.npos)), dim3(64 * kHeadWaves), 0, st, P);
    return hipGetLastError();
}

hipError_t launch_gru_prep_enc(const float* u, const int32_t* perm, float* X, int B, int L, int interleaved, hipStream_t st) {
    const size_t n = (size_t)B * L;
    const int grid = (int)std::min<size_t>((n + 255) / 256, 4096);
    hipLaunchKernelGGL(gru_prep_enc_kernel, dim3(grid), dim3(256), 0, st, u, perm, X, B, L, interleaved);
    return hipGetLastError();
}

}  // namespace tae
