// TurboAE whole-block kernels in the f16x2 representation (see turboae_device.hpp, "fp16-split
// contraction"): the same workgroup organisation as turboae_kernels.hip - nb whole blocks per workgroup,
// activations resident in LDS for every stack, 8 waves = 4 position groups x 2 channel halves, weights
// streamed from L2 in A-fragment order, in-place panel update, Linear head fused on the accumulators -
// but every fp32 operand of the convolutions is carried as two fp16 halves and the contraction runs on
// v_mfma_f32_16x16x32_f16 (3 products per 32 k) instead of v_mfma_f32_16x16x4_f32 (8 per 32 k).
//
// LDS planes (bytes per row): ACT_HI / ACT_LO (rows+1) x U halves; XA_HI / XA_LO / XB_HI / XB_LO
// (rows+4) x 8 halves; PERM, INV int32[L]; HS head-combine scratch.  Same total as the fp32 layout.
// Because the planes are position-major and unpadded, the im2col row of position t (5 taps x U channels)
// is the 5*U contiguous halves starting at row t-2 of each plane: a B fragment (8 consecutive k) is two
// ds_read_b64 per plane.
//
// Packed stack (bytes), written by turboae_api.hip::pack_stack_h:
//   per layer: A fragments [slab][channel tile][hi | lo][lane][8 halves] | bias * 2^S [CP] fp32 | 2^-S x 4 fp32
//   then Linear weights [8][CP] fp32 | bias [8] fp32
// where 2^S is the layer's power-of-two weight scale (max |w| * 2^S in [2^13, 2^14)).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "turboae_internal.hpp"
#include "turboae_device.hpp"

namespace tae {

template <int U>
struct GeoH {
    static constexpr int CT = (U + 15) / 16;
    static constexpr int CP = CT * 16;
    static constexpr int NSL_MID = (5 * U + 31) / 32;   // 32-k slabs of a U->U layer
    static constexpr int NSL_L0 = 2;                    // first layer: 5 taps x 8 padded inputs = 40 k
    static constexpr uint32_t SLB = CT * 2048u;         // bytes of A fragments per slab
    static constexpr uint32_t MIDB = NSL_MID * SLB;
    static constexpr uint32_t L0B = NSL_L0 * SLB;
    static constexpr uint32_t TAILB = CP * 4u + 16u;    // bias + scale block after the fragments
};

constexpr int kXRowB = 16;        // bytes per row of an X plane (8 halves)
constexpr int kXSlack = 3;        // extra rows of the X planes: the first layer's second slab over-reads up to row + 5

// one stack-input panel = two fp16 planes
struct XPlane {
    char* h;
    char* l;
    __device__ __forceinline__ float read(int row, int c) const {
        return (float)reinterpret_cast<const _Float16*>(h)[row * 8 + c] + (float)reinterpret_cast<const _Float16*>(l)[row * 8 + c];
    }
    __device__ __forceinline__ void write(int row, int c, float v) const {
        v = __builtin_amdgcn_fmed3f(v, -kH2Limit, kH2Limit);
        const _Float16 hi = (_Float16)v;
        reinterpret_cast<_Float16*>(h)[row * 8 + c] = hi;
        reinterpret_cast<_Float16*>(l)[row * 8 + c] = (_Float16)(v - (float)hi);
    }
};

struct PanelsH {
    char* AH;
    char* AL;
    XPlane XA, XB;
    int* PERM;
    int* INV;
    float* HS;
};

template <int U>
__device__ __forceinline__ PanelsH carve_h(char* smem, int rows, int L) {
    PanelsH pn;
    const size_t ab = (size_t)(rows + 1) * U * 2, xb = (size_t)(rows + 1 + kXSlack) * kXRowB;
    pn.AH = smem;
    pn.AL = pn.AH + ab;
    pn.XA.h = pn.AL + ab;
    pn.XA.l = pn.XA.h + xb;
    pn.XB.h = pn.XA.l + xb;
    pn.XB.l = pn.XB.h + xb;
    pn.PERM = reinterpret_cast<int*>(pn.XB.l + xb);
    pn.INV = pn.PERM + L;
    pn.HS = reinterpret_cast<float*>(smem + (((reinterpret_cast<char*>(pn.INV + L) - smem) + 15) & ~15));
    return pn;
}

template <int U, int C0, int NC>
struct WeightStreamH {
    __amdgpu_buffer_rsrc_t rsrc;
    uint32_t voff;
    OpsHA<NC> a;
    __device__ __forceinline__ void init(const void* wpack, uint32_t bytes, int lane) {
        rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(wpack), 0, (int)bytes, 0x00020000);
        voff = (uint32_t)lane * 16u;
    }
    __device__ __forceinline__ void prefetch(uint32_t soff) { load_wh<GeoH<U>::CT, C0, NC>(a, rsrc, voff, soff); }
};

// One SameShapeConv1d stack (cnn_utils.py:36-46) + Linear head; same contract as run_stack in
// turboae_kernels.hip.  `vmax` collects max |activation| before the fp16-range clamp (overflow report).
template <int U, int PT, int C0, int NC, class Epi>
__device__ __forceinline__ void run_stack_h(const char* __restrict__ wpack, uint32_t soff, uint32_t snext, int n_layer, char* smem,
                                            const PanelsH& pn, const XPlane& xin, const TileCtx<PT>& tc, int g, int lane,
                                            WeightStreamH<U, C0, NC>& ws, float& vmax, Epi epi) {
    using G = GeoH<U>;
    constexpr int CTT = G::CT;
    const int q = lane >> 4;
    f32x4 acc[PT][NC];
    uint32_t lo = soff;
    float inv_scale = 1.0f;
    for (int l = 0; l < n_layer; ++l) {
        const bool first = (l == 0);
        const uint32_t fragb = first ? G::L0B : G::MIDB;
        const float* bias = reinterpret_cast<const float*>(wpack + lo + fragb);
        inv_scale = bias[G::CP];
        {
            f32x4 b4[NC];
#pragma unroll
            for (int i = 0; i < NC; ++i) b4[i] = *reinterpret_cast<const f32x4*>(bias + (C0 + i) * 16 + 4 * q);
#pragma unroll
            for (int p = 0; p < PT; ++p)
#pragma unroll
                for (int i = 0; i < NC; ++i) acc[p][i] = b4[i];
        }
        uint32_t bh[PT], bl[PT];
        const uint32_t stride = first ? (uint32_t)kXRowB : (uint32_t)(U * 2);
        const uint32_t ph = (uint32_t)((first ? xin.h : pn.AH) - smem), pl = (uint32_t)((first ? xin.l : pn.AL) - smem);
#pragma unroll
        for (int p = 0; p < PT; ++p) {
            const uint32_t o = (uint32_t)(tc.row[p] - 2) * stride + 16u * q;
            bh[p] = ph + o;
            bl[p] = pl + o;
        }
        if (first) conv_accumulate_h<CTT, C0, NC, PT, G::NSL_L0>(acc, ws.a, ws.rsrc, ws.voff, lo, smem, bh, bl);
        else conv_accumulate_h<CTT, C0, NC, PT, G::NSL_MID>(acc, ws.a, ws.rsrc, ws.voff, lo, smem, bh, bl);
        lo += fragb + G::TAILB;
        {
            const uint32_t nxt = (l + 1 < n_layer) ? lo : snext;
            if (nxt != 0xffffffffu) ws.prefetch(nxt);
        }
        if (l + 1 < n_layer) {
            if (!first && !(TAE_X & 1)) __syncthreads();
#pragma unroll
            for (int p = 0; p < PT; ++p) {
#pragma unroll
                for (int i = 0; i < NC; ++i) {
                    f32x4 v = acc[p][i] * inv_scale;
                    if (!(TAE_X & 4)) { v.x = elu1(v.x); v.y = elu1(v.y); v.z = elu1(v.z); v.w = elu1(v.w); }
                    // ELU output is >= -1: only the upper side can leave the fp16 range
                    vmax = fmaxf(fmaxf(vmax, fmaxf(v.x, v.y)), fmaxf(v.z, v.w));
                    v.x = fminf(v.x, kH2Limit); v.y = fminf(v.y, kH2Limit); v.z = fminf(v.z, kH2Limit); v.w = fminf(v.w, kH2Limit);
                    h4 hi, lw;
                    split4(v, hi, lw);
                    const int ch = (C0 + i) * 16 + 4 * q;
                    if (TAE_X & 2) asm volatile("" :: "v"(hi), "v"(lw));
                    else if (tc.valid[p] && ch < U) {
                        *reinterpret_cast<h4*>(pn.AH + (size_t)(tc.row[p] * U + ch) * 2) = hi;
                        *reinterpret_cast<h4*>(pn.AL + (size_t)(tc.row[p] * U + ch) * 2) = lw;
                    }
                }
            }
            if (!(TAE_X & 1)) __syncthreads();
        }
    }
    // ---- Linear head on the accumulators of the last conv layer, fp32 vector ALU (as in run_stack)
    const float* wl = reinterpret_cast<const float*>(wpack + lo);
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
            f32x4 v = acc[p][i] * inv_scale;
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
    const int n = lane & 15;
    float* HS = pn.HS;
    if constexpr (C0 != 0) {
#pragma unroll
        for (int p = 0; p < PT; ++p)
            *reinterpret_cast<float2*>(HS + ((g * PT + p) * 16 + n) * 8 + 2 * q) = float2{k2[p][0], k2[p][1]};
    }
    __syncthreads();
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

__device__ __forceinline__ void report_range(float vmax, uint32_t* flags) {
    if (flags != nullptr && !(vmax <= kH2Limit)) atomicOr(flags, 1u);     // also catches NaN
}

// =============================================================================================
// Decoder: DEC_LargeCNN.forward (decoders.py:206-269)
template <int U, int PT, int C0, int NC>
__device__ __forceinline__ void dec_body_h(const FusedParams& P, char* smem, const PanelsH& pn, const TileCtx<PT>& tc, int g, int lane, int blk0) {
    const int L = P.L;
    const int n_stack = 2 * P.n_iter;
    const int F = P.F;
    const bool extrinsic = P.extrinsic != 0;
    float* xdec = P.out + (size_t)blk0 * L;
    const char* wpack = reinterpret_cast<const char*>(P.wpack);
    WeightStreamH<U, C0, NC> ws;
    ws.init(wpack, P.wpack_bytes, lane);
    ws.prefetch(0);
    const uint32_t sstride = P.stack_stride;       // bytes in this representation
    float vmax = 0.0f;
    for (int s = 0; s < n_stack; ++s) {
        const XPlane Xin = (s & 1) ? pn.XB : pn.XA;
        const XPlane Xout = (s & 1) ? pn.XA : pn.XB;
        const int* ptab = (s & 1) ? pn.PERM : pn.INV;
        if (s + 1 < n_stack) {
            run_stack_h<U, PT, C0, NC>(wpack, s * sstride, (s + 1) * sstride, P.n_layer, smem, pn, Xin, tc, g, lane, ws, vmax,
                                       [&](int p, int f, float v) {
                if (f < F) {
                    if (extrinsic) v -= Xin.read(tc.row[p], 2 + f);            // decoders.py:235-236,246-247
                    vmax = fmaxf(vmax, fabsf(v));
                    Xout.write(tc.rowbase[p] + ptab[tc.t[p]], 2 + f, v);       // interleave / deinterleave (decoders.py:238,249)
                }
            });
        } else {
            run_stack_h<U, PT, C0, NC>(wpack, s * sstride, 0xffffffffu, P.n_layer, smem, pn, Xin, tc, g, lane, ws, vmax,
                                       [&](int p, int f, float v) {
                if (f == 0) xdec[tc.blk[p] * L + ptab[tc.t[p]]] = 1.0f / (1.0f + expf(-v));   // decoders.py:262-267
            });
        }
    }
    report_range(vmax, P.flags);
}

template <int U, int PT>
__global__ __launch_bounds__(kThreads, 2) void dec_kernel_h(FusedParams P) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = wave & (kGroups - 1), h = wave / kGroups;
    const int L = P.L, nb = P.nb;
    const int rows = nb * (L + 2) + 2;
    const PanelsH pn = carve_h<U>(smem, rows, L);
    const int blk0 = blockIdx.x * nb;
    const int nblk = min(nb, P.B - blk0);
    const int npos = nblk * L;

    zero_lds(smem, P.lds_bytes, tid);
    __syncthreads();
    for (int i = tid; i < L; i += kThreads) { pn.PERM[i] = P.perm[i]; pn.INV[i] = P.inv[i]; }
    __syncthreads();
    // r_sys, r_par1 -> XA ch 0,1 (natural order); r_sys_int, r_par2 -> XB ch 0,1 (decoders.py:221-224)
    const float* rx = P.in + (size_t)blk0 * L * 3;
    float vmax = 0.0f;
    for (int m = tid; m < npos; m += kThreads) {
        const int b = m / L, t = m - b * L;
        const int row = b * (L + 2) + 2 + t;
        const float* r = rx + (size_t)m * 3;
        const float r0 = r[0], r1 = r[1], r2 = r[2], ri = rx[((size_t)b * L + pn.PERM[t]) * 3 + 0];
        vmax = fmaxf(fmaxf(vmax, fmaxf(fabsf(r0), fabsf(r1))), fmaxf(fabsf(r2), fabsf(ri)));
        pn.XA.write(row, 0, r0);
        pn.XA.write(row, 1, r1);
        pn.XB.write(row, 0, ri);
        pn.XB.write(row, 1, r2);
    }
    report_range(vmax, P.flags);
    __syncthreads();

    TileCtx<PT> tc;
    make_tiles<PT>(tc, g, lane, L, npos);
    const bool upper = __builtin_amdgcn_readfirstlane(h) != 0;
    if (!upper) dec_body_h<U, PT, 0, Split<U>::CTA>(P, smem, pn, tc, g, lane, blk0);
    else dec_body_h<U, PT, Split<U>::CTA, Split<U>::CTB>(P, smem, pn, tc, g, lane, blk0);
}

// =============================================================================================
// Encoder before power normalisation: ENC_interCNN.forward (encoders.py:362-373)
template <int U, int PT, int C0, int NC>
__device__ __forceinline__ void enc_body_h(const FusedParams& P, char* smem, const PanelsH& pn, const TileCtx<PT>& tc, int g, int lane,
                                           int blk0, double& sum, double& sumsq) {
    const int L = P.L;
    float* xtx = P.out + (size_t)blk0 * L * 3;
    const bool act_elu = P.act == 0;
    const char* wpack = reinterpret_cast<const char*>(P.wpack);
    WeightStreamH<U, C0, NC> ws;
    ws.init(wpack, P.wpack_bytes, lane);
    ws.prefetch(0);
    const uint32_t sstride = P.stack_stride;
    float vmax = 0.0f;
    for (int s = 0; s < 3; ++s) {
        const XPlane Xin = (s == 2) ? pn.XB : pn.XA;
        run_stack_h<U, PT, C0, NC>(wpack, s * sstride, s < 2 ? (s + 1) * sstride : 0xffffffffu, P.n_layer, smem, pn, Xin, tc, g, lane, ws, vmax,
                                   [&](int p, int f, float v) {
            if (f == 0) {
                if (act_elu) v = elu1(v);                                  // enc_act (encoders.py:364)
                xtx[(size_t)(tc.blk[p] * L + tc.t[p]) * 3 + s] = v;        // x_p2 stays in interleaved order (encoders.py:371-373)
                sum += (double)v;
                sumsq += (double)v * (double)v;
            }
        });
    }
    report_range(vmax, P.flags);
}

template <int U, int PT>
__global__ __launch_bounds__(kThreads, 2) void enc_kernel_h(FusedParams P) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = wave & (kGroups - 1), h = wave / kGroups;
    const int L = P.L, nb = P.nb;
    const int rows = nb * (L + 2) + 2;
    const PanelsH pn = carve_h<U>(smem, rows, L);
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
        pn.XA.write(row, 0, 2.0f * u[m] - 1.0f);
        pn.XB.write(row, 0, 2.0f * u[b * L + pn.PERM[t]] - 1.0f);
    }
    __syncthreads();

    TileCtx<PT> tc;
    make_tiles<PT>(tc, g, lane, L, npos);
    double sum = 0.0, sumsq = 0.0;
    const bool upper = __builtin_amdgcn_readfirstlane(h) != 0;
    if (!upper) enc_body_h<U, PT, 0, Split<U>::CTA>(P, smem, pn, tc, g, lane, blk0, sum, sumsq);
    else enc_body_h<U, PT, Split<U>::CTA, Split<U>::CTB>(P, smem, pn, tc, g, lane, blk0, sum, sumsq);
    block_reduce_stats(smem, tid, sum, sumsq, P.partials);
}

// ---------------------------------------------------------------------------------------------
template <int U>
static hipError_t launch_fused_h_u(bool decoder, const FusedParams& P, int grid, hipStream_t st) {
    constexpr int PT = 5;
    auto kd = dec_kernel_h<U, PT>;
    auto ke = enc_kernel_h<U, PT>;
    const void* fn = decoder ? reinterpret_cast<const void*>(kd) : reinterpret_cast<const void*>(ke);
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, P.lds_bytes);
    if (e != hipSuccess) return e;
    if (decoder) hipLaunchKernelGGL(kd, dim3(grid), dim3(kThreads), P.lds_bytes, st, P);
    else hipLaunchKernelGGL(ke, dim3(grid), dim3(kThreads), P.lds_bytes, st, P);
    return hipGetLastError();
}

hipError_t launch_fused_h(int U, bool decoder, const FusedParams& P, int grid, hipStream_t st) {
    switch (U) {
        case 100: return launch_fused_h_u<100>(decoder, P, grid, st);
        case 64: return launch_fused_h_u<64>(decoder, P, grid, st);
        case 32: return launch_fused_h_u<32>(decoder, P, grid, st);
        default: return hipErrorInvalidValue;
    }
}

int fused_lds_bytes_h(int U, int L, int nb) {
    const int rows = nb * (L + 2) + 2;
    size_t b = 2 * (size_t)(rows + 1) * U * 2 + 4 * (size_t)(rows + 1 + kXSlack) * kXRowB + 2 * (size_t)L * 4;
    b = (b + 15) & ~(size_t)15;
    b += (size_t)kHeadSlots * 8 * 4;
    if (b < 2 * kThreads * sizeof(double)) b = 2 * kThreads * sizeof(double);
    return (int)b;
}

}  // namespace tae
