// DeepTurbo GRU decoder (BASELINE configs[4]): DEC_LargeRNN.forward (decoders.py:84-149) with
// dec_rnn = 'gru' (get_args.py:80), dec_act = 'linear' (get_args.py:101), dropout = 0 at eval:
// the same turbo iteration as DEC_LargeCNN with every conv stack replaced by
// torch.nn.GRU(2+F, H, num_layers=2, bidirectional=True, batch_first=True) + Linear(2H -> F | 1).
//
// PyTorch GRU cell (gate order r, z, n; the arithmetic lives in ATen, third-party to the reference):
//   r = sigmoid(W_ir x + b_ir + W_hr h + b_hr)      z = sigmoid(W_iz x + b_iz + W_hz h + b_hz)
//   n = tanh(W_in x + b_in + r * (W_hn h + b_hn))   h' = (1 - z) * n + z * h          h_0 = 0
//
// First (correctness-first) MI355X mapping - a parity configuration, not the bench line:
//   gru_prep  : received (B,L,3) -> XA / XB input panels in HBM (B,L,8), as in the CNN kernels
//   gru_rec   : one workgroup = NBK blocks x ONE direction of one layer, strictly sequential over t.
//               Thread j (< 3H) owns gate row j: its W_hh row stays in registers for all L steps, h_t
//               is broadcast from LDS, gate pre-activations are exchanged through LDS, 2 barriers/step.
//               Layer 0 computes its input projection (K = 2+F) on the fly from the panel staged in
//               LDS; layer 1 reads the precomputed projections.
//   gru_proj  : layer-1 input projections for all positions, both directions: GI = Y0 * W_ih1^T + b_ih1
//               (non-sequential half of the FLOPs); thread j owns a W_ih row (2H registers).
//   gru_head  : Linear(2H -> F|1) + extrinsic subtraction + (de)interleave scatter into the other
//               panel, or sigmoid + deinterleave for the last half-iteration (decoders.py:145-147).
// All fp32, VALU FMA chains in k order.  The MFMA formulation (gates as a [3H x NBK] GEMM per step
// with W_hh fragments resident in LDS) is the planned follow-up; see DESIGN.md.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "turboae_internal.hpp"

namespace tae {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;

// two independent fp32 FMAs in one v_pk_fma_f32: dot products run as two interleaved chains (even / odd k)
// that are added at the end; the weight pair (w[k], w[k+1]) is a natural register pair (a broadcast {w, w}
// operand would be materialised per weight and double the register footprint)
__device__ __forceinline__ f32x2 pk_fma(f32x2 w, f32x2 x, f32x2 acc) { return __builtin_elementwise_fma(w, x, acc); }

constexpr int kGruH = 100;          // hidden units per direction (dec_num_unit)
constexpr int kGruRows = 3 * kGruH; // gate rows per direction
constexpr int kGruThreads = 300;    // one thread per gate row (5 waves, the last one partially filled): no per-row predicates,
                                    // which would let the compiler sink the FMAs below the LDS reads and spill all of h
constexpr int kGruNBK = 8;          // blocks per workgroup in gru_rec
constexpr int kGruPT = 32;          // positions per workgroup in gru_proj
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

// One direction of one GRU layer for kGruNBK blocks, sequential over the block length.
//   LAYER0: gi = W_ih x_t + b_ih computed on the fly from X (B,L,8) (only the first 2+F columns are non-zero);
//   else  : gi read from GI (B,L,2,3H).
// Y (B,L,2H): h_t is written to columns [dir*H, dir*H + H).
template <bool LAYER0>
__global__ __launch_bounds__(kGruThreads) void gru_rec_kernel(GruRecParams P) {
    constexpr int H = kGruH, NBK = kGruNBK;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* hbuf = reinterpret_cast<float*>(smem);            // [NBK][H]
    float* gates = hbuf + NBK * H;                           // [NBK][5][H]: r, z, n_i, n_h, (unused)
    float* xs = gates + NBK * 5 * H;                         // LAYER0: [NBK][L][8]
    const int tid = threadIdx.x, j = tid;
    const int dir = blockIdx.y;
    const int L = P.L;
    const int b0 = blockIdx.x * NBK;
    const int nblk = min(NBK, P.B - b0);
    const int jj = j;

    // this thread's recurrent weights stay in registers for the whole sequence
    float wh[H];
    {
        const float* w = P.w_hh + ((size_t)dir * kGruRows + jj) * H;
#pragma unroll
        for (int k = 0; k < H; k += 4) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(w + k);
            wh[k] = v.x; wh[k + 1] = v.y; wh[k + 2] = v.z; wh[k + 3] = v.w;
        }
    }
    const float bh = P.b_hh[dir * kGruRows + jj];
    const float bi = P.b_ih[dir * kGruRows + jj];
    float wi[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) wi[c] = (LAYER0 && c < P.cin) ? P.w_ih[((size_t)dir * kGruRows + jj) * P.cin + c] : 0.0f;

    for (int i = tid; i < NBK * H; i += kGruThreads) hbuf[i] = 0.0f;     // h_0 = 0; layout hbuf[b][k]
    if (LAYER0) {
        const float* X = P.x + (size_t)b0 * L * kXWg;
        for (int i = tid; i < nblk * L * 2; i += kGruThreads)
            reinterpret_cast<f32x4*>(xs)[i] = reinterpret_cast<const f32x4*>(X)[i];
    }
    __syncthreads();

    const int gate = jj / H, u = jj - gate * H;       // gate 0 = r, 1 = z, 2 = n
    const float* GI = P.gi + (size_t)b0 * L * 2 * kGruRows + (size_t)dir * kGruRows + jj;
    float* Y = P.y + (size_t)b0 * L * 2 * H + (size_t)dir * H;
    for (int s = 0; s < L; ++s) {
        const int t = dir == 0 ? s : L - 1 - s;
        float gi[NBK];
#pragma unroll
        for (int b = 0; b < NBK; ++b) {
            if (LAYER0) {
                const f32x4 x0 = *reinterpret_cast<const f32x4*>(xs + (b * L + t) * kXWg);
                const f32x4 x1 = *reinterpret_cast<const f32x4*>(xs + (b * L + t) * kXWg + 4);
                float a = bi;
                a = fmaf(wi[0], x0.x, a); a = fmaf(wi[1], x0.y, a); a = fmaf(wi[2], x0.z, a); a = fmaf(wi[3], x0.w, a);
                a = fmaf(wi[4], x1.x, a); a = fmaf(wi[5], x1.y, a); a = fmaf(wi[6], x1.z, a); a = fmaf(wi[7], x1.w, a);
                gi[b] = a;
            } else {
                gi[b] = (b < nblk) ? GI[((size_t)b * L + t) * 2 * kGruRows] : 0.0f;
            }
        }
        // gh[b] = b_hh[j] + W_hh[j,:] . h_t[b,:]: h broadcast from LDS, two k per v_pk_fma_f32
        f32x2 gh2[NBK];
#pragma unroll
        for (int b = 0; b < NBK; ++b) gh2[b] = f32x2{bh, 0.0f};
#pragma unroll
        for (int k = 0; k < H; k += 4) {
#pragma unroll
            for (int b = 0; b < NBK; ++b) {
                const f32x4 hv = *reinterpret_cast<const f32x4*>(hbuf + b * H + k);
                gh2[b] = pk_fma(f32x2{wh[k], wh[k + 1]}, f32x2{hv.x, hv.y}, gh2[b]);
                gh2[b] = pk_fma(f32x2{wh[k + 2], wh[k + 3]}, f32x2{hv.z, hv.w}, gh2[b]);
            }
            // keep the scheduler from hoisting all the broadcast reads above the FMAs (it spills otherwise)
            __builtin_amdgcn_sched_barrier(0);
        }
        // branch-free stores (a per-thread `if (gate < 2)` lets the compiler sink the FMA chains below all
        // the LDS reads, which then spill): slot = gate holds sigmoid(pre) for r/z and gi for n; the n rows
        // park gh in slot 3, the r/z rows in the unused slot 4
#pragma unroll
        for (int b = 0; b < NBK; ++b) {
            const float ghb = gh2[b].x + gh2[b].y;
            const float sg = sigmoidf_(gi[b] + ghb);
            gates[(b * 5 + gate) * H + u] = gate < 2 ? sg : gi[b];
            gates[(b * 5 + (gate < 2 ? 4 : 3)) * H + u] = ghb;
        }
        __syncthreads();
        for (int e = tid; e < nblk * H; e += kGruThreads) {
            const int b = e / H, uu = e - b * H;
            const float r = gates[(b * 5 + 0) * H + uu], z = gates[(b * 5 + 1) * H + uu];
            const float n = tanhf(gates[(b * 5 + 2) * H + uu] + r * gates[(b * 5 + 3) * H + uu]);
            const float hn = (1.0f - z) * n + z * hbuf[b * H + uu];
            hbuf[b * H + uu] = hn;
            Y[((size_t)b * L + t) * 2 * H + uu] = hn;
        }
        __syncthreads();
    }
}

// GI[p][dir][j] = b_ih[dir][j] + sum_k W_ih[dir][j][k] * Yin[p][k], K = 2H (layer-1 input projections).
// Thread j owns W_ih row j (2H registers); the position tile is staged in LDS and broadcast.
__global__ __launch_bounds__(kGruThreads, 2) void gru_proj_kernel(GruProjParams P) {
    constexpr int K = 2 * kGruH, PT = kGruPT;
    __shared__ __attribute__((aligned(16))) float ys[PT * K];
    const int tid = threadIdx.x;
    const int dir = blockIdx.y;
    const int jj = tid;
    float w[K];
    {
        const float* wp = P.w_ih + ((size_t)dir * kGruRows + jj) * K;
#pragma unroll
        for (int k = 0; k < K; k += 4) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(wp + k);
            w[k] = v.x; w[k + 1] = v.y; w[k + 2] = v.z; w[k + 3] = v.w;
        }
    }
    const float bias = P.b_ih[dir * kGruRows + jj];
    const size_t p0 = (size_t)blockIdx.x * PT;
    const int np = (int)min((size_t)PT, P.npos - p0);
    for (int i = tid; i < np * K / 4; i += kGruThreads)
        reinterpret_cast<f32x4*>(ys)[i] = reinterpret_cast<const f32x4*>(P.yin + p0 * K)[i];
    __syncthreads();
#pragma unroll 1
    for (int p = 0; p < np; p += 2) {
        f32x2 a0 = {bias, 0.0f}, a1 = {bias, 0.0f};
        const float* y0 = ys + p * K;
        const float* y1 = ys + (p + 1 < np ? p + 1 : p) * K;
#pragma unroll
        for (int k = 0; k < K; k += 4) {
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(y0 + k);
            const f32x4 v1 = *reinterpret_cast<const f32x4*>(y1 + k);
            a0 = pk_fma(f32x2{w[k], w[k + 1]}, f32x2{v0.x, v0.y}, a0);
            a1 = pk_fma(f32x2{w[k], w[k + 1]}, f32x2{v1.x, v1.y}, a1);
            a0 = pk_fma(f32x2{w[k + 2], w[k + 3]}, f32x2{v0.z, v0.w}, a0);
            a1 = pk_fma(f32x2{w[k + 2], w[k + 3]}, f32x2{v1.z, v1.w}, a1);
        }
        P.gi[((p0 + p) * 2 + dir) * kGruRows + jj] = a0.x + a0.y;
        if (p + 1 < np) P.gi[((p0 + p + 1) * 2 + dir) * kGruRows + jj] = a1.x + a1.y;
    }
}

// Linear(2H -> F|1) + dec_act (linear) + extrinsic + (de)interleave (decoders.py:104-147)
__global__ __launch_bounds__(256) void gru_head_kernel(GruHeadParams P) {
    constexpr int K = 2 * kGruH, KP = K + 1, NP = 64;       // padded LDS rows: conflict-free column walks
    __shared__ float ys[NP * KP];
    __shared__ float wl[8 * K];
    __shared__ float bl[8];
    const int tid = threadIdx.x;
    const size_t p0 = (size_t)blockIdx.x * NP;
    const int np = (int)min((size_t)NP, P.npos - p0);
    for (int i = tid; i < 8 * K; i += 256) wl[i] = (i / K) < P.nout ? P.w[i] : 0.0f;
    if (tid < 8) bl[tid] = tid < P.nout ? P.b[tid] : 0.0f;
    for (int i = tid; i < np * K; i += 256) {
        const int p = i / K, k = i - p * K;
        ys[p * KP + k] = P.y[p0 * K + i];
    }
    __syncthreads();
    const int p = tid & (NP - 1), fq = tid >> 6;         // thread (position, f in {fq, fq + 4})
    if (p >= np) return;
    float a0 = bl[fq], a1 = bl[fq + 4];
    for (int k = 0; k < K; ++k) {
        const float yv = ys[p * KP + k];
        a0 = fmaf(wl[fq * K + k], yv, a0);
        a1 = fmaf(wl[(fq + 4) * K + k], yv, a1);
    }
    const size_t pos = p0 + p;
    const size_t b = pos / P.L;
    const int t = (int)(pos - b * P.L);
    if (!P.last) {
        const float* xc = P.xcur + pos * kXWg;
        float* xn = P.xnext + (b * P.L + P.ptab[t]) * kXWg;
        if (fq < P.F) xn[2 + fq] = a0 - (P.extrinsic ? xc[2 + fq] : 0.0f);
        if (fq + 4 < P.F) xn[2 + fq + 4] = a1 - (P.extrinsic ? xc[2 + fq + 4] : 0.0f);
    } else if (fq == 0) {
        P.xdec[b * P.L + P.ptab[t]] = sigmoidf_(a0);      // sigmoid(deinterleave(x_plr)), decoders.py:145-147
    }
}

hipError_t launch_gru_prep(const float* rx, const int32_t* perm, float* XA, float* XB, int B, int L, hipStream_t st) {
    const size_t n = (size_t)B * L;
    const int grid = (int)std::min<size_t>((n + 255) / 256, 4096);
    hipLaunchKernelGGL(gru_prep_kernel, dim3(grid), dim3(256), 0, st, rx, perm, XA, XB, B, L);
    return hipGetLastError();
}

int gru_rec_lds_bytes(int L, bool layer0) {
    size_t b = (size_t)kGruNBK * kGruH * 4 + (size_t)kGruNBK * 5 * kGruH * 4;
    if (layer0) b += (size_t)kGruNBK * L * kXWg * 4;
    return (int)((b + 15) & ~(size_t)15);
}

hipError_t launch_gru_rec(bool layer0, const GruRecParams& P, hipStream_t st) {
    const int lds = gru_rec_lds_bytes(P.L, layer0);
    const dim3 grid((P.B + kGruNBK - 1) / kGruNBK, 2);
    if (layer0) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gru_rec_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(gru_rec_kernel<true>, grid, dim3(kGruThreads), lds, st, P);
    } else {
        hipLaunchKernelGGL(gru_rec_kernel<false>, grid, dim3(kGruThreads), lds, st, P);
    }
    return hipGetLastError();
}

hipError_t launch_gru_proj(const GruProjParams& P, hipStream_t st) {
    const dim3 grid((unsigned)((P.npos + kGruPT - 1) / kGruPT), 2);
    hipLaunchKernelGGL(gru_proj_kernel, grid, dim3(kGruThreads), 0, st, P);
    return hipGetLastError();
}

hipError_t launch_gru_head(const GruHeadParams& P, hipStream_t st) {
    const unsigned grid = (unsigned)((P.npos + 63) / 64);
    hipLaunchKernelGGL(gru_head_kernel, dim3(grid), dim3(256), 0, st, P);
    return hipGetLastError();
}

int gru_max_rec_block_len() { return (160 * 1024 - kGruNBK * 6 * kGruH * 4) / (kGruNBK * kXWg * 4); }

}  // namespace tae
