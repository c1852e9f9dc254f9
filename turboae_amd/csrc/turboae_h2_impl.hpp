#pragma once
// (included by turboae_h2.hip and turboae_h2_w.hip, which instantiate the kernels for different channel widths - two translation
// units so that `make -j` compiles them in parallel: one file took 4.5 minutes)
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
// Packed stack (bytes), written by turboae_api_create.hip::pack_stack_h:
//   per layer: A fragments [slab][channel tile][hi | lo][lane][8 halves] | bias * 2^S [CP] fp32 | 2^-S x 4 fp32
//   then Linear weights [8][CP] fp32 | bias [8] fp32
// where 2^S is the layer's power-of-two weight scale (max |w| * 2^S in [2^13, 2^14)).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "turboae_internal.hpp"
#include "turboae_device.hpp"
#include <type_traits>

#ifndef TAE_PAIR_REVERSE
#define TAE_PAIR_REVERSE 1
#endif

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
    static constexpr uint32_t TAILB = CP * 4u + 32u;    // bias + scale block (8 floats) after the fragments
    static constexpr bool TAIL20 = (U == 100);          // 5 taps only; the whole-block kernels end every layer in a two-MFMA tail slab (conv_accumulate_h; the host packs it so)
};

// The same for a kernel size known at run time (5, 7, 9): the plain conv stacks take it from the launch parameters; the
// dense stacks stay on the 5-tap constants above.
struct TapGeo {
    int pad, nsl_mid, nsl_l0;
    uint32_t midb, l0b;
};
template <int U>
__device__ __forceinline__ TapGeo tap_geo(int taps) {
    TapGeo t;
    t.pad = taps >> 1;
    t.nsl_mid = (taps * U + 31) / 32;
    t.nsl_l0 = (taps * 8 + 31) / 32;       // first layer: taps x 8 padded inputs
    t.midb = (uint32_t)t.nsl_mid * GeoH<U>::SLB;
    t.l0b = (uint32_t)t.nsl_l0 * GeoH<U>::SLB;
    return t;
}

constexpr int kXRowB = 16;        // bytes per row of an X plane (8 halves)
constexpr int kRangeHeaderB = 32, kRangeLayerB = 32;   // PanelsH::RNG: header + one row of per-wave maxima per (stack, layer) the workgroup runs
constexpr int kXSlack = 3;        // extra rows of the X planes: the first layer's last slab over-reads up to 4 * nsl_l0 - 2 * pad - 2 <= 2 rows past the panel

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

// Encoder stacks have ONE input channel, so the first layer's K = taps: instead of one 8-wide k chunk per tap (taps x 8 padded
// inputs = 2..3 slabs, 5/64 real at 5 taps) the taps are folded into one 32-k slab: chunk q of an output position reads input row
// (t - pad + q), and channel a of that row holds the input 4a positions further on, X[row][a] = x[row + 4a], so k = 8q + a is tap
// q + 4a (pack_stack_h packs the weights to match).  Called by the thread that stages input position t (panel row `row`) with value
// v: scatters v into channel a of the row 4a positions back.  `t` = index inside the block (rows before t = -pad belong to the
// previous block's tail and are never written), `row_min` = first row of the panel this workgroup may write.
__device__ __forceinline__ void fold_enc_input(const XPlane& X, int row, int t, int pad, int row_min, float v) {
#pragma unroll
    for (int a = 1; a <= 2; ++a)
        if (4 * a <= 2 * pad && t - 4 * a >= -pad && row - 4 * a >= row_min) X.write(row - 4 * a, a, v);
}

struct PanelsH {
    int dump;        // write-only row after the slack row (index rows + 1): where padding lanes / padding channels store
    uint32_t panel_bytes;   // dense stacks: bytes between consecutive layer panels (hi plane | lo plane); AH / AL = panel 0
    char* AH;
    char* AL;
    XPlane XA, XB;
    int* PERM;
    int* INV;
    int* ROWT;       // [kHeadSlots] panel row of every position slot of the workgroup
    float* HS;
    uint32_t* RNG;   // range bookkeeping: [0..1] the launch's flag word, [2..3] its calibration array (device pointers, parked here so that
                     // they cost no scalar registers across the K loops), [4] the staged stack inputs' maximum (float bits), then
                     // [8 + 8 * layer + wave]: every wave's max |scaled ELU output| of (stack, layer) `layer`, checked once at the end
};

template <int U>
__device__ __forceinline__ PanelsH carve_h(char* smem, int rows, int L) {
    PanelsH pn;
    pn.dump = rows + 1;
    const size_t ab = (size_t)(rows + 2) * U * 2, xb = (size_t)(rows + 1 + kXSlack) * kXRowB;
    pn.AH = smem;
    pn.AL = pn.AH + ab;
    pn.XA.h = pn.AL + ab;
    pn.XA.l = pn.XA.h + xb;
    pn.XB.h = pn.XA.l + xb;
    pn.XB.l = pn.XB.h + xb;
    pn.PERM = reinterpret_cast<int*>(pn.XB.l + xb);
    pn.INV = pn.PERM + L;
    pn.ROWT = pn.INV + L;
    pn.HS = reinterpret_cast<float*>(smem + (((reinterpret_cast<char*>(pn.ROWT + kHeadSlots) - smem) + 15) & ~15));
    pn.RNG = reinterpret_cast<uint32_t*>(pn.HS + kHeadSlots * 8);
    return pn;
}

// Per-lane view of the position tiles a wave owns, kept small (the accumulators, two A-fragment sets and the
// B ring leave few registers): everything else is recomputed where it is needed (once per stack).
template <int PT>
struct TileH {
    static constexpr int kTiles = PT;
    const int* rowtab;  // LDS table [PT][16] of this wave's group: panel row of tile p's position n (padding lanes: row 2, real
                        // finite data).  Read on demand (twice per layer) instead of living in - and being spilled from - VGPRs.
    uint32_t valid;     // bit p: in-block position held in this workgroup's panel (its activations are written back)
    uint32_t center;    // bit p: position whose stack output this workgroup owns (== valid for whole blocks)
    int m0;             // workgroup-relative position of tile 0 (tile p: m0 + 16 p)
    int L;
    int pad;            // zero rows in front of every block (kernel size / 2)
    __device__ __forceinline__ int row(int p) const {
        const int* t = rowtab;
        asm volatile("" : "+v"(t));          // keep the load where it is used (it is loop-invariant: hoisted, it would be spilled)
        return t[p * 16];
    }
    __device__ __forceinline__ bool ok(int p) const { return (valid >> p) & 1u; }
    __device__ __forceinline__ bool own(int p) const { return (center >> p) & 1u; }
    // block / index-in-block of tile p's position: recomputed where they are used (the head epilogues, once per stack) - computed
    // once and kept, the ten values are loop-invariant across the stack loop and were what the register allocator spilled
    __device__ __forceinline__ int mpos(int p) const { int m = m0 + 16 * p; asm volatile("" : "+v"(m)); return m; }
    __device__ __forceinline__ int blk(int p) const { return mpos(p) / L; }
    __device__ __forceinline__ int t(int p) const { const int m = mpos(p); return m - (m / L) * L; }
    __device__ __forceinline__ int rowbase(int p) const { return blk(p) * (L + pad) + pad; }
};

// Padding lanes (positions past the workgroup's blocks) compute on row 2 and store to the write-only dump row, so
// the epilogue needs no per-tile branches.
// `gt0` = first position tile of the wave's group; a group walks PT <= 5 tiles.

// Even deal of the workgroup's ceil(npos / 16) position tiles over the kGroups position groups (wave-uniform).
struct GroupSpan { int gt0, live; };
__device__ __forceinline__ GroupSpan group_span(int npos, int g) {
    const int ntile = (npos + 15) / 16;
    const int base = ntile / kGroups, rem = ntile - base * kGroups;
    GroupSpan s;
    s.gt0 = __builtin_amdgcn_readfirstlane(g * base + min(g, rem));
    s.live = __builtin_amdgcn_readfirstlane(base + (g < rem ? 1 : 0));
    return s;
}

// Calls f(integral_constant<PT'>) with PT' = clamp(live, 1, PTMAX): one code path per tile count (a group without
// any tile still walks one all-padding tile: it has to meet the others at the per-layer barriers).
template <int PTMAX, class F>
__device__ __forceinline__ void dispatch_tiles(int live, F&& f) {
    static_assert(PTMAX == 5, "tile-count dispatch is written for 5 tiles per group at most");
    if (live >= 5) f(std::integral_constant<int, 5>{});
    else if (live == 4) f(std::integral_constant<int, 4>{});
    else if (live == 3) f(std::integral_constant<int, 3>{});
    else if (live == 2) f(std::integral_constant<int, 2>{});
    else f(std::integral_constant<int, 1>{});
}

template <int PT>
__device__ __forceinline__ void make_tiles_h(TileH<PT>& tc, int* rowtab, int gt0, int lane, int L, int npos, int pad) {
    const int n = lane & 15;
    tc.valid = 0u;
    tc.m0 = gt0 * 16 + n;
    tc.L = L;
    tc.pad = pad;
    tc.rowtab = rowtab + gt0 * 16 + n;
#pragma unroll
    for (int p = 0; p < PT; ++p) {
        const int m = tc.m0 + 16 * p;
        const bool v = m < npos;
        const int b = (v ? m : 0) / L;
        if (lane < 16) rowtab[(gt0 + p) * 16 + n] = v ? b * (L + pad) + pad + (m - b * L) : pad;    // both channel halves write the same values
        tc.valid |= (v ? 1u : 0u) << p;
    }
    tc.center = tc.valid;
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

// ---- range bookkeeping of the fp16-split representation ------------------------------------------------------------
// hi + lo carries a value to 2^-22 relative only while its lo half is a NORMAL fp16 number, i.e. for |x| >= 2^-3; below that
// the pair has an absolute floor of 2^-25, and above 65504 it overflows.  fp32 has neither limit, so every panel is stored
// SCALED: layer l writes ELU(v) * 2^A_l with a per-layer exponent the host calibrates (turboae_api_create.hip::calibrate_range) so
// that the layer's largest activation lands in [2^10, 2^11): 2^5 of headroom above, and every value down to 2^-13 of the
// maximum keeps the full 2^-22 (the floor is then 2^-35 of the maximum - far below the fp32 accumulation noise of the dot
// products that consume it).  The next layer's accumulators carry 2^(S + A_l) (its bias is pre-scaled, its 2^-(S + A_l) comes
// from the packed tail), so the scale costs no instruction of its own: ELU(a k) c = med3(a (k c), expm1(a k) c, 0) with c folded
// into the last operation of either expm1 branch.
// Every scale is a power of two: results do not depend on A_l except where a value meets the floor or the ceiling.
// At run time each layer's workgroup-wide maximum is checked against both ends (flags bit 0: above 65504, results invalid;
// bit 1: below the threshold the host packs beside the scales - 2^3, i.e. the data sits >= 2^7 under the calibration maximum
// and the pair is no longer fp32-grade).

struct EluScale {
    float inv;     // 2^-(S + A_in)           : accumulator -> value
    float k1;      // 2^-(S + A_in) * 2^A_out : accumulator -> scaled value
    float k2;      // 2^-(S + A_in) * log2(e) : accumulator -> exp2 argument
    float c;       // 2^A_out
};

struct RangeH {
    uint32_t* row;     // PanelsH::RNG row of this stack's layer 0 (8 words per layer: one per wave)
};

__device__ __forceinline__ void range_park(uint32_t* rng, uint32_t* flags, uint32_t* cal) {     // thread 0, before the first barrier
    reinterpret_cast<uint32_t**>(rng)[0] = flags;
    reinterpret_cast<uint32_t**>(rng)[1] = cal;
}
__device__ __forceinline__ uint32_t* range_flags(const uint32_t* rng) { return reinterpret_cast<uint32_t* const*>(rng)[0]; }
__device__ __forceinline__ uint32_t* range_cal(const uint32_t* rng) { return reinterpret_cast<uint32_t* const*>(rng)[1]; }
__device__ __forceinline__ uint32_t* range_rows(uint32_t* rng) { return rng + 8; }

__device__ __forceinline__ void lds_max_bits(uint32_t* slot, float v) {
    using lds_u32 = uint32_t __attribute__((address_space(3)));
    __hip_atomic_fetch_max(reinterpret_cast<lds_u32*>((uint32_t)(uintptr_t)slot), __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// Maximum of a non-negative float over the 64 lanes as a wave-uniform bit pattern: four DPP steps inside the rows of 16, then the
// four row results through scalar registers.  (Non-negative floats order like their bit patterns.)  A dozen instructions per
// layer and wave - against the r04 first cut, an LDS atomic per lane plus a check by thread 0 behind the layer barrier: +8.6 %
// decoder time (tools/ab_libs.sh, TAE_RANGE_BOOK builds): every layer's barrier waited for wave 0's dependent LDS / global reads.
__device__ __forceinline__ uint32_t wave_max_bits(float x) {
    int v = (int)__float_as_uint(x);
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, false));     // quad_perm [1, 0, 3, 2]
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, false));     // quad_perm [2, 3, 0, 1]
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, false));    // row_half_mirror
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, false));    // row_mirror
    const int a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16);
    const int c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
    return (uint32_t)max(max(a, b), max(c, d));
}
// after a layer's panel writes: park this wave's maximum of the SCALED values; nothing reads it before range_finish.
// TAE_WAVE_REDUCE 1: reduce inside the rows of 16 with DPP, then the four row leaders meet in the wave's slot with an LDS max
// (4 lanes, no scalar round trip); 0: the full reduction to a scalar and one plain store.
#ifndef TAE_WAVE_REDUCE
#define TAE_WAVE_REDUCE 1
#endif
__device__ __forceinline__ void range_note_layer(const RangeH& rg, int l, float vmax) {
#if TAE_WAVE_REDUCE
    int v = (int)__float_as_uint(vmax);
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, false));
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, false));
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, false));
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, false));
    if ((threadIdx.x & 15) == 0) {
        using lds_u32 = uint32_t __attribute__((address_space(3)));
        __hip_atomic_fetch_max(reinterpret_cast<lds_u32*>((uint32_t)(uintptr_t)(rg.row + l * 8 + (threadIdx.x >> 6))), (uint32_t)v,
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
#else
    const uint32_t m = wave_max_bits(vmax);
    if ((threadIdx.x & 63) == 0) rg.row[l * 8 + (threadIdx.x >> 6)] = m;
#endif
}
// End of the kernel, after a barrier: thread i checks (stack, layer) row i against the layer's packed tail (2^-(S + A_in) |
// 2^A_out | low-side threshold | high-side threshold) - `tail_of(i)` returns it, or nullptr for a row that holds no panel.
template <class TailOf>
__device__ __forceinline__ void range_finish(uint32_t* rng, int n_rows, int cal_base, TailOf tail_of) {
    // strided: 2 * num_iteration * dec_num_layer rows may exceed the workgroup's 512 threads (60 iterations x 5 layers: ADVICE r04 -
    // rows past 512 used to stay unchecked and uncalibrated)
    for (int i = threadIdx.x; i < n_rows; i += kThreads) {
        const float* tail = tail_of(i);
        if (tail == nullptr) continue;
        const uint32_t* r = range_rows(rng) + i * 8;
        uint32_t mb = 0u;
#pragma unroll
        for (int w = 0; w < kWaves; ++w) mb = max(mb, r[w]);
        const float m = __uint_as_float(mb);
        const uint32_t f = (!(m <= tail[3]) ? 1u : 0u) | ((m < tail[2]) ? 2u : 0u);
        uint32_t* flags = range_flags(rng);
        if (f != 0u && flags != nullptr) atomicOr(flags, f);
        uint32_t* cal = range_cal(rng);
        if (cal != nullptr) atomicMax(cal + cal_base + i, __float_as_uint(m / tail[1]));
    }
}

// Layer epilogue for 4 accumulator values: ELU(acc * 2^-S') * 2^A, running max of |.| (range report), split into fp16
// halves.  Scalar fp32 ops on purpose: the packed forms (v_pk_mul_f32 / v_pk_add_f32) measured 6 % slower
// here.  No clamp: an out-of-range activation turns into inf / NaN halves and `vmax` reports it
// (tae_range_status).
// TAE_ELU_MODE (A/B builds): 0 = exp2 only (r03: absolute error 3e-8 * c), 1 = both expm1 branches per value (elu1's own
// arithmetic: +9 vector-ALU instructions per value, measured +7 % decoder time - the epilogues of the two waves of a SIMD
// coincide behind the layer barrier, so their vector-ALU time is exposed), 2 (default) = the host picks per LAYER from the
// layer's calibrated maximum M (tail[4]; calibrate_range in turboae_api_create.hip):
//   kind 1, M <= 2^-5: x (1 + x/2 + x^2/6 + x^3/24) for both signs of x, no exp2 - truncation x^4/120 <= 1.3e-7 relative up to
//     |x| = 1/16; the layer's high-side threshold is lowered to |x| = 1/8 (2e-6) so that data which outgrows the polynomial
//     raises TAE_RANGE_HIGH instead of losing accuracy silently;
//   kind 2, 2^-5 < M < 1/4: both branches per value (exact to fp32 rounding at any magnitude);
//   kind 0, M >= 1/4: exp2 - 1, absolute error 3e-8 <= 2^-23 of the layer's largest value (one fp32 ulp of it) - what every O(1)
//     network (the trained ones: layer maxima 0.5 .. 16) has always run, at the r03 cost.
#ifndef TAE_ELU_MODE
#define TAE_ELU_MODE 2
#endif
#ifndef TAE_RANGE_BOOK
#define TAE_RANGE_BOOK 1     // A/B builds: 0 = no maxima, no range checks of the panels; 2 = the per-value maximum only (what r03 carried)
#endif
__device__ __forceinline__ float elu_scaled_exp(float a, const EluScale& s) {
    const float e = __builtin_fmaf(__builtin_amdgcn_exp2f(a * s.k2), s.c, -s.c);
    return __builtin_amdgcn_fmed3f(a * s.k1, e, 0.0f);
}
__device__ __forceinline__ float elu_scaled_poly(float a, const EluScale& s) {
    const float x = a * s.inv, xs = a * s.k1;
    float p = __builtin_fmaf(x, 1.0f / 24.0f, 1.0f / 6.0f);
    p = __builtin_fmaf(x, p, 0.5f);
    p = __builtin_fmaf(x, p, 1.0f);
    return __builtin_amdgcn_fmed3f(xs, xs * p, 0.0f);          // x > 0: xs p > xs > 0 -> xs; x < 0: xs < xs p < 0 -> xs p
}
__device__ __forceinline__ float elu_scaled_both(float a, const EluScale& s) {      // elu1(a * inv) * c (turboae_device.hpp), the scale folded in
    const float x = a * s.inv, xs = a * s.k1;
    const float small = xs * expm1_poly(x);
    const float big = __builtin_fmaf(__builtin_amdgcn_exp2f(a * s.k2), s.c, -s.c);
    return __builtin_amdgcn_fmed3f(xs, x > kExpm1Switch ? small : big, 0.0f);
}
template <int KIND>      // the same three branches on an unscaled value (the last layer of a stack: its ELU feeds the Linear head in fp32)
__device__ __forceinline__ float elu_kind(float x) {
    if constexpr (KIND == 0) return __builtin_amdgcn_fmed3f(x, __builtin_amdgcn_exp2f(x * 1.44269504088896341f) - 1.0f, 0.0f);
    else if constexpr (KIND == 1) {
        float p = __builtin_fmaf(x, 1.0f / 24.0f, 1.0f / 6.0f);
        p = __builtin_fmaf(x, p, 0.5f);
        p = __builtin_fmaf(x, p, 1.0f);
        return __builtin_amdgcn_fmed3f(x, x * p, 0.0f);
    } else return elu1(x);
}
template <int KIND>      // 0: exp2 branch, 1: polynomial branch, 2: both per value
__device__ __forceinline__ void elu_split4(f32x4 a, const EluScale& s, float& vmax, h4& hi, h4& lo) {
    f32x4 v;
    if (TAE_X & 4) v = a * s.k1;
    else if constexpr (KIND == 0) { v.x = elu_scaled_exp(a.x, s); v.y = elu_scaled_exp(a.y, s); v.z = elu_scaled_exp(a.z, s); v.w = elu_scaled_exp(a.w, s); }
    else if constexpr (KIND == 1) { v.x = elu_scaled_poly(a.x, s); v.y = elu_scaled_poly(a.y, s); v.z = elu_scaled_poly(a.z, s); v.w = elu_scaled_poly(a.w, s); }
    else {
        // the rare branch (layer maxima in (2^-5, 1)): one value at a time - interleaved four deep, its temporaries were what
        // pushed three registers of the decoder into scratch
        v.x = elu_scaled_both(a.x, s); __builtin_amdgcn_sched_barrier(0);
        v.y = elu_scaled_both(a.y, s); __builtin_amdgcn_sched_barrier(0);
        v.z = elu_scaled_both(a.z, s); __builtin_amdgcn_sched_barrier(0);
        v.w = elu_scaled_both(a.w, s); __builtin_amdgcn_sched_barrier(0);
    }
    // two v_max3_f32 with |.| modifiers per four values (written as a tree the compiler spent 3.2 instructions on them)
    vmax = fmaxf(fmaxf(vmax, fabsf(v.x)), fabsf(v.y));
    vmax = fmaxf(fmaxf(vmax, fabsf(v.z)), fabsf(v.w));
    split4(v, hi, lo);
}
// Packed tail (CP bias values, then 2^-(S + A_in) | 2^A_out | low | high | ELU kind) of (stack, layer) row i of a plain conv network
// (a stack's last layer feeds the Linear head unscaled: 2^A_out = 1, no low-side check).  Layout as pack_stack_h writes it.
template <int U>
__device__ __forceinline__ const float* plain_tail(const float* wpack, uint32_t stack_stride, int n_layer, int taps, int i) {
    using G = GeoH<U>;
    const int s = i / n_layer, l = i - s * n_layer;
    const TapGeo tg = tap_geo<U>(taps);
    const uint32_t off = (uint32_t)s * stack_stride + tg.l0b + (uint32_t)l * (tg.midb + G::TAILB);
    return reinterpret_cast<const float*>(reinterpret_cast<const char*>(wpack) + off) + G::CP;
}
template <int U>
__device__ __forceinline__ const float* dense_tail(const float* wpack, uint32_t soff, int n_layer, int l) {
    using G = GeoH<U>;
    if (l + 1 >= n_layer) return nullptr;
    uint32_t off = soff;
    for (int k = 0; k < l; ++k) off += G::L0B + (uint32_t)k * G::MIDB + G::TAILB;
    off += G::L0B + (uint32_t)l * G::MIDB;
    return reinterpret_cast<const float*>(reinterpret_cast<const char*>(wpack) + off) + G::CP;
}

// Timing experiment TAE_X & 1024 (results stay correct): shader-clock stamps of workgroup 0 at the phase boundaries of every conv layer of
// the stack that ran last (bias loaded | K loop issued | panel free | epilogue issued | panel written), printed per wave by dec_kernel_h.
constexpr int kStampBase = 155 * 1024;             // 158 720: behind the U = 100 panels (158 240 B), inside the 160 KB the experiment launch asks for
__device__ __forceinline__ void stamp_h(char* smem, int lane, int layer, int i) {
    if constexpr ((TAE_X & 1024) != 0) {
        if (blockIdx.x == 0 && layer < 5) {
            __builtin_amdgcn_sched_barrier(0);
            const uint32_t t = (uint32_t)__builtin_readcyclecounter();
            if (lane == 0) reinterpret_cast<uint32_t*>(smem + kStampBase)[((threadIdx.x >> 6) * 5 + layer) * 8 + i] = t;
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// One SameShapeConv1d stack (cnn_utils.py:36-46) + Linear head; same contract as run_stack in
// turboae_kernels.hip, except that `g` is the first position TILE of the wave's group (not the group index).  `rg`: range
// bookkeeping of the layer panels (above; TRACK = 0 compiles it out, 1 = the panels, 2 = also the maximum of the last layer's ELU
// output, which only calibration launches need - in the production decoder it cost nine spilled registers); the stack inputs are `xin` as the caller scaled them
// (the first layer's packed 2^-(S + A_x) undoes it).
// `l0_slabs` > 0: the first layer walks that many K slabs instead of the tap_geo count (encoder stacks: C_in = 1 folds all taps
// into ONE slab, see fold_enc_input).
template <int U, int PT, int C0, int NC, int TRACK = 1, bool HEAD2 = (TRACK == 2), int PROD = 3, bool T20 = GeoH<U>::TAIL20, class Epi>
__device__ __forceinline__ void run_stack_h(const char* __restrict__ wpack, uint32_t soff, uint32_t snext, int n_layer, char* smem,
                                            const PanelsH& pn, const XPlane& xin, const TileH<PT>& tc, int g, int lane,
                                            WeightStreamH<U, C0, NC>& ws, const RangeH& rg, Epi epi, int l0_slabs = 0) {
    using G = GeoH<U>;
    constexpr int CTT = G::CT;
    const int q = lane >> 4;
    const int dump_row = pn.dump;
    const TapGeo tg = tap_geo<U>(G::TAIL20 ? 5 : tc.pad * 2 + 1);      // the 100-wide instantiations run 5 taps only (conv_kernel_width on the host): constants
    f32x4 acc[PT][NC];
    uint32_t lo = soff;
    float inv_scale = 1.0f;
    for (int l = 0; l < n_layer; ++l) {
        const bool first = (l == 0);
        const uint32_t fragb = first ? tg.l0b : tg.midb;
        const float* bias = reinterpret_cast<const float*>(wpack + lo + fragb);
        // the layer's scales (tail: 2^-(S + A_in) | 2^A_out | low | high | ELU kind) are fetched HERE, a K loop ahead of the epilogue
        // that needs them: fetched after the loop (r04 first cut, to spare three scalar registers) every layer's epilogue began with
        // an exposed scalar-load latency - 60 of them per decoder workgroup, +2 % (tools/ab_abi.sh against the r03 library)
        stamp_h(smem, lane, l, 0);
        inv_scale = bias[G::CP];
        const float out_scale_l = bias[G::CP + 1];
        const int kind_l = (int)bias[G::CP + 4];
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
            const uint32_t o = (uint32_t)(tc.row(p) - tg.pad) * stride + 16u * q;
            bh[p] = ph + o;
            bl[p] = pl + o;
        }
        // T20 (the whole-block 100-wide kernels; the host packs their layers so, LayoutH::tail20): every layer ends in a two-MFMA tail
        // slab (5 taps only - other kernel sizes at these widths run on the 124-wide instantiation): first layer 1 full slab + tail (the
        // encoder's folded input: its tail fragments are zero), U -> U layers 15 + tail
        conv_accumulate_h<CTT, C0, NC, PT, 0, false, PROD, T20>(acc, ws.a, ws.rsrc, ws.voff, lo, smem, bh, bl,
                                              T20 ? (first ? 1 : tg.nsl_mid - 1) : (first ? (l0_slabs > 0 ? l0_slabs : tg.nsl_l0) : tg.nsl_mid));
        stamp_h(smem, lane, l, 1);
        lo += fragb + G::TAILB;
        {
            const uint32_t nxt = (l + 1 < n_layer) ? lo : snext;
            if (nxt != 0xffffffffu) ws.prefetch(nxt);
        }
        if (l + 1 < n_layer) {
            if (!first && !(TAE_X & 1)) __syncthreads();
            stamp_h(smem, lane, l, 2);
            int wrow[PT];
#pragma unroll
            for (int p = 0; p < PT; ++p) wrow[p] = tc.row(p);
            const float out_scale = out_scale_l;
            const EluScale es{inv_scale, inv_scale * out_scale, inv_scale * 1.44269504088896341f, out_scale};
            float vmax = 0.0f;
            auto write_panel = [&](auto kind) {
                constexpr int KIND = decltype(kind)::value;
#pragma unroll
                for (int p = 0; p < PT; ++p) {
#pragma unroll
                    for (int i = 0; i < NC; ++i) {
                        h4 hi, lw;
                        elu_split4<KIND>(acc[p][i], es, vmax, hi, lw);
                        // channels >= U exist only in the last channel tile (zero weights, zero bias -> ELU(0) = 0): they are
                        // steered to the dump row instead of branching
                        const int ch = (C0 + i) * 16 + 4 * q;
                        const bool inb = (((C0 + i) * 16 + 16 <= U) || (ch < U)) && tc.ok(p);
                        const int off = inb ? (wrow[p] * U + ch) * 2 : (dump_row * U + 4 * q) * 2;
                        if (TAE_X & 2) asm volatile("" :: "v"(hi), "v"(lw));
                        else {
                            *reinterpret_cast<h4*>(pn.AH + off) = hi;
                            *reinterpret_cast<h4*>(pn.AL + off) = lw;
                        }
                    }
                }
            };
#if TAE_ELU_MODE == 2
            const int kind = __builtin_amdgcn_readfirstlane(kind_l);
            if (kind == 0) write_panel(std::integral_constant<int, 0>{});
            else if (kind == 1) write_panel(std::integral_constant<int, 1>{});
            else write_panel(std::integral_constant<int, 2>{});
#elif TAE_ELU_MODE == 1
            write_panel(std::integral_constant<int, 2>{});
#else
            write_panel(std::integral_constant<int, 0>{});
#endif
            if (TAE_RANGE_BOOK == 1 && TRACK) range_note_layer(rg, l, vmax);      // !TRACK: vmax is dead, its arithmetic goes with it
            else if (TAE_RANGE_BOOK == 2) asm volatile("" :: "v"(vmax));      // A/B: the per-value maximum alone (as r03 carried it), no per-layer reduction
            stamp_h(smem, lane, l, 3);
            if (!(TAE_X & 1)) __syncthreads();
            stamp_h(smem, lane, l, 4);
        }
    }
    // ---- Linear head on the accumulators of the last conv layer, fp32 vector ALU (as in run_stack)
    const float* wl = reinterpret_cast<const float*>(wpack + lo);
    // ELU of the last layer with the branch the host picked for it (its outputs are a fifth of all ELU evaluations: both expm1
    // branches on every one of them cost the decoder 1.4 %), fused with the Linear FMAs so that every accumulator dies as it is
    // used (ELU in place first, FMAs after: 12 spilled registers).  TRACK == 2 (calibration launches): their maximum goes to the
    // stack's last range row.
    float part[PT][8];
#pragma unroll
    for (int p = 0; p < PT; ++p)
#pragma unroll
        for (int f = 0; f < 8; ++f) part[p][f] = 0.0f;
    // calibration: max |ELU output| of the last layer from its pre-activations, in a loop of its own (threaded through the head's
    // FMAs a running maximum spilled 60 registers): ELU(x) = x above 0, 1 - exp(-|x|) below
    float hmax = 0.0f;
    if constexpr (TRACK == 2) {
        float hneg = 0.0f;
#pragma unroll
        for (int p = 0; p < PT; ++p)
#pragma unroll
            for (int i = 0; i < NC; ++i) {
                hmax = fmaxf(fmaxf(hmax, fmaxf(acc[p][i].x, acc[p][i].y)), fmaxf(acc[p][i].z, acc[p][i].w));
                hneg = fmaxf(fmaxf(hneg, fmaxf(-acc[p][i].x, -acc[p][i].y)), fmaxf(-acc[p][i].z, -acc[p][i].w));
            }
        hmax = fmaxf(hmax * inv_scale, 1.0f - __expf(-hneg * inv_scale));
    }
    // Which expm1 the last layer gets is a COMPILE-TIME property of the instantiation (a run-time choice - two copies of the head, or
    // a wave-uniform branch per tile - spilled 10 / 48 registers): HEAD2 instantiations evaluate both branches per value - the "full"
    // one (TRACK == 2: calibration launches, tap export) and, since r05, a production twin of the plain kernels for networks one of
    // whose last layers stays below 1/4 (the host decides, FusedParams::head2; it carries none of the calibration bookkeeping, so a
    // full-size launch does not run the spilling instantiation); the plain instantiations keep r03's exp2 - 1 (3e-8 absolute against a
    // layer maximum >= 1/4).
    constexpr int HEAD_KIND = (TAE_ELU_MODE == 0) ? 0 : ((TAE_ELU_MODE == 1 || HEAD2) ? 2 : 0);
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        f32x4 w4[8];
#pragma unroll
        for (int f = 0; f < 8; ++f) w4[f] = *reinterpret_cast<const f32x4*>(wl + f * G::CP + (C0 + i) * 16 + 4 * q);
#pragma unroll
        for (int p = 0; p < PT; ++p) {
            f32x4 v = acc[p][i] * inv_scale;
            v.x = elu_kind<HEAD_KIND>(v.x); v.y = elu_kind<HEAD_KIND>(v.y); v.z = elu_kind<HEAD_KIND>(v.z); v.w = elu_kind<HEAD_KIND>(v.w);
#pragma unroll
            for (int f = 0; f < 8; ++f) {
                float s = part[p][f];
                s = fmaf(v.x, w4[f].x, s); s = fmaf(v.y, w4[f].y, s);
                s = fmaf(v.z, w4[f].z, s); s = fmaf(v.w, w4[f].w, s);
                part[p][f] = s;
            }
        }
    }
    if constexpr (TAE_RANGE_BOOK == 1 && TRACK == 2) range_note_layer(rg, n_layer - 1, hmax);
    const bool hi32 = (q & 2) != 0, hi16 = (q & 1) != 0;
    float k2[PT][2];
#pragma unroll
    for (int p = 0; p < PT; ++p) butterfly8(part[p], hi32, hi16, k2[p]);
    const int n = lane & 15;
    float* HS = pn.HS;
    if constexpr (C0 != 0) {
#pragma unroll
        for (int p = 0; p < PT; ++p)
            *reinterpret_cast<float2*>(HS + ((g + p) * 16 + n) * 8 + 2 * q) = float2{k2[p][0], k2[p][1]};
    }
    __syncthreads();
    if constexpr (C0 == 0) {
        const float* lb = wl + 8 * G::CP;
        const float bq0 = lb[2 * q], bq1 = lb[2 * q + 1];
#pragma unroll
        for (int p = 0; p < PT; ++p) {
            float2 other = float2{0.0f, 0.0f};
            if constexpr (NC < CTT) other = *reinterpret_cast<const float2*>(HS + ((g + p) * 16 + n) * 8 + 2 * q);
            if (tc.own(p)) {
                epi(p, 2 * q, (k2[p][0] + other.x) + bq0);
                epi(p, 2 * q + 1, (k2[p][1] + other.y) + bq1);
            }
        }
    }
    __syncthreads();
}

// DenseSameShapeConv1d stack (cnn_utils.py:49-82) + Linear head: layer l convolves cat(inputs, out_0 .. out_{l-1}).  The
// contraction over that concatenation is the sum of one first-layer-style K loop over the input planes and l
// mid-layer-style K loops over the panels of the earlier outputs, all into the same accumulators; layer l's ELU output
// goes to its own panel l (never in place: one barrier per layer).  Packed layer l: input-part fragments (2 slabs) |
// l x panel-part fragments (NSL_MID slabs each) | bias * 2^S | 2^-S.  `active` = this wave's position group has live
// tiles (small dense panels fill one group; the others only keep the barriers company).
template <int U, int PT, int C0, int NC, class Epi>
__device__ __forceinline__ void run_stack_h_dense(const char* __restrict__ wpack, uint32_t soff, int n_layer, char* smem,
                                                  const PanelsH& pn, const XPlane& xin, const TileH<PT>& tc, int g, int lane,
                                                  WeightStreamH<U, C0, NC>& ws, const RangeH& rg, bool active, Epi epi) {
    using G = GeoH<U>;
    constexpr int CTT = G::CT;
    const int q = lane >> 4;
    const int dump_row = pn.dump;
    f32x4 acc[PT][NC];
    uint32_t lo = soff;
    float inv_scale = 1.0f;
    for (int l = 0; l < n_layer; ++l) {
        const uint32_t tail = lo + G::L0B + (uint32_t)l * G::MIDB;
        const float* tailf = reinterpret_cast<const float*>(wpack + tail) + G::CP;     // 2^-(S + A) | 2^A | low-side threshold (one exponent per dense stack)
        if (active) {
            const float* bias = reinterpret_cast<const float*>(wpack + tail);
            inv_scale = bias[G::CP];
            f32x4 b4[NC];
#pragma unroll
            for (int i = 0; i < NC; ++i) b4[i] = *reinterpret_cast<const f32x4*>(bias + (C0 + i) * 16 + 4 * q);
#pragma unroll
            for (int p = 0; p < PT; ++p)
#pragma unroll
                for (int i = 0; i < NC; ++i) acc[p][i] = b4[i];
            uint32_t bh[PT], bl[PT];
            {   // the stack inputs (2 + F or 1 channels, 8 halves per row)
                const uint32_t ph = (uint32_t)(xin.h - smem), pl = (uint32_t)(xin.l - smem);
#pragma unroll
                for (int p = 0; p < PT; ++p) {
                    const uint32_t o = (uint32_t)(tc.row(p) - 2) * kXRowB + 16u * q;
                    bh[p] = ph + o;
                    bl[p] = pl + o;
                }
                ws.prefetch(lo);
                conv_accumulate_h<CTT, C0, NC, PT, G::NSL_L0>(acc, ws.a, ws.rsrc, ws.voff, lo, smem, bh, bl);
            }
            for (int k = 0; k < l; ++k) {          // the outputs of layers 0 .. l-1
                const uint32_t ph = (uint32_t)(pn.AH - smem) + (uint32_t)k * pn.panel_bytes, pl = ph + (uint32_t)(pn.AL - pn.AH);
#pragma unroll
                for (int p = 0; p < PT; ++p) {
                    const uint32_t o = (uint32_t)(tc.row(p) - 2) * (uint32_t)(U * 2) + 16u * q;
                    bh[p] = ph + o;
                    bl[p] = pl + o;
                }
                const uint32_t wo = lo + G::L0B + (uint32_t)k * G::MIDB;
                ws.prefetch(wo);
                conv_accumulate_h<CTT, C0, NC, PT, G::NSL_MID>(acc, ws.a, ws.rsrc, ws.voff, wo, smem, bh, bl);
            }
        }
        lo = tail + G::TAILB;
        if (l + 1 < n_layer) {
            if (active) {
                char* PH = pn.AH + (size_t)l * pn.panel_bytes;
                char* PL = PH + (pn.AL - pn.AH);
                const float out_scale = tailf[1];
                const EluScale es{inv_scale, inv_scale * out_scale, inv_scale * 1.44269504088896341f, out_scale};
                float vmax = 0.0f;
#pragma unroll
                for (int p = 0; p < PT; ++p) {
                    const int wrow = tc.row(p);
#pragma unroll
                    for (int i = 0; i < NC; ++i) {
                        h4 hi, lw;
                        elu_split4<2>(acc[p][i], es, vmax, hi, lw);       // both expm1 branches per value: dense stacks are a parity path, not a bench line
                        const int ch = (C0 + i) * 16 + 4 * q;
                        const bool inb = (((C0 + i) * 16 + 16 <= U) || (ch < U)) && tc.ok(p);
                        const int off = inb ? (wrow * U + ch) * 2 : (dump_row * U + 4 * q) * 2;
                        *reinterpret_cast<h4*>(PH + off) = hi;
                        *reinterpret_cast<h4*>(PL + off) = lw;
                    }
                }
                range_note_layer(rg, l, vmax);
            }
            __syncthreads();
        }
    }
    // ---- Linear head (as run_stack_h)
    const float* wl = reinterpret_cast<const float*>(wpack + lo);
    float k2[PT][2];
    if (active) {
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
#pragma unroll
        for (int p = 0; p < PT; ++p) butterfly8(part[p], hi32, hi16, k2[p]);
    }
    const int n = lane & 15;
    float* HS = pn.HS;
    if constexpr (C0 != 0) {
        if (active) {
#pragma unroll
            for (int p = 0; p < PT; ++p)
                *reinterpret_cast<float2*>(HS + ((g + p) * 16 + n) * 8 + 2 * q) = float2{k2[p][0], k2[p][1]};
        }
    }
    __syncthreads();
    if constexpr (C0 == 0) {
        if (active) {
            const float* lb = wl + 8 * G::CP;
            const float bq0 = lb[2 * q], bq1 = lb[2 * q + 1];
#pragma unroll
            for (int p = 0; p < PT; ++p) {
                float2 other = float2{0.0f, 0.0f};
                if constexpr (NC < CTT) other = *reinterpret_cast<const float2*>(HS + ((g + p) * 16 + n) * 8 + 2 * q);
                if (tc.own(p)) {
                    epi(p, 2 * q, (k2[p][0] + other.x) + bq0);
                    epi(p, 2 * q + 1, (k2[p][1] + other.y) + bq1);
                }
            }
        }
    }
    __syncthreads();
}

// Range tracking outside the K loops (stack inputs, head outputs): max |x| where a NaN counts as +inf - fmaxf alone would
// drop it.  Inside the layers (elu_split4) a NaN can only descend from a NaN input (tracked here) or from an activation
// that overflowed one layer earlier (inf halves; already reported through vmax by then).
__device__ __forceinline__ float track_abs(float vmax, float x) {
    return (x == x) ? fmaxf(vmax, fabsf(x)) : __builtin_inff();
}
__device__ __forceinline__ void report_range(float vmax, uint32_t* flags) {
    if (flags != nullptr && !(vmax <= kH2Limit)) atomicOr(flags, 1u);
}
// stack inputs (received values, extrinsic values): `xmax` = this lane's max of the SCALED values it wrote; calibration launches
// also collect the unscaled maximum in cal[0]
__device__ __forceinline__ void report_range_x(float xmax, float x_inv, uint32_t* flags, uint32_t* cal) {
    report_range(xmax, flags);
    if (cal != nullptr) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) xmax = fmaxf(xmax, __shfl_xor(xmax, off));
        if ((threadIdx.x & 63) == 0) atomicMax(cal, __float_as_uint(xmax * x_inv));
    }
}
// after the staging loop of a decoder workgroup: its largest staged input against the low end of the window (thread 0, after a barrier)
__device__ __forceinline__ void range_check_inputs(uint32_t* slot, float low, uint32_t* flags) {
    const float m = __uint_as_float(*slot);
    *slot = 0u;
    if (m < low && flags != nullptr) atomicOr(flags, 2u);
}

// =============================================================================================
// Decoder: DEC_LargeCNN.forward (decoders.py:206-269)
template <int U, int PT, int C0, int NC, bool TAPS, bool HEAD2, int PROD = 3>
__device__ __forceinline__ void dec_body_h(const FusedParams& P, char* smem, const PanelsH& pn, const TileH<PT>& tc, int g, int lane, int blk0) {
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
    const float xs = P.x_scale, xinv = P.x_inv;    // the X planes hold value * 2^A_x (one exponent for the whole decoder)
    float xmax = 0.0f;
    for (int s = 0; s < n_stack; ++s) {
        const XPlane Xin = (s & 1) ? pn.XB : pn.XA;
        const XPlane Xout = (s & 1) ? pn.XA : pn.XB;
        const int* ptab = (s & 1) ? pn.PERM : pn.INV;
        const RangeH rg{range_rows(pn.RNG) + s * P.n_layer * 8};
        if (s + 1 < n_stack) {
            run_stack_h<U, PT, C0, NC, TAPS ? 2 : 1, HEAD2, PROD>(wpack, s * sstride, (s + 1) * sstride, P.n_layer, smem, pn, Xin, tc, g, lane, ws, rg,
                                       [&](int p, int f, float v) {
                if (f < F) {
                    if (extrinsic) v -= Xin.read(tc.row(p), 2 + f) * xinv;     // decoders.py:235-236,246-247
                    if constexpr (TAPS) { if (P.tap_out != nullptr) P.tap_out[(((size_t)s * P.B + blk0 + tc.blk(p)) * L + tc.t(p)) * F + f] = v; }
                    const float vs = v * xs;
                    xmax = track_abs(xmax, vs);
                    Xout.write(tc.rowbase(p) + ptab[tc.t(p)], 2 + f, vs);      // interleave / deinterleave (decoders.py:238,249)
                }
            });
        } else {
            run_stack_h<U, PT, C0, NC, TAPS ? 2 : 1, HEAD2, PROD>(wpack, s * sstride, 0xffffffffu, P.n_layer, smem, pn, Xin, tc, g, lane, ws, rg,
                                       [&](int p, int f, float v) {
                if (f == 0) xdec[tc.blk(p) * L + ptab[tc.t(p)]] = 1.0f / (1.0f + expf(-v));   // decoders.py:262-267
            });
        }
    }
    report_range_x(xmax, xinv, range_flags(pn.RNG), range_cal(pn.RNG));
}

template <int U, int PT, bool TAPS = false, bool HEAD2 = TAPS, int PROD = 3>
__global__ __launch_bounds__(kThreads, 2) void dec_kernel_h(FusedParams P) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // wave w sits on SIMD w % 4; the lower channel half (4 of 7 channel tiles) of position group s runs on SIMD s, the upper half
    // (3 tiles) of group 3 - s: group_span deals the extra position tiles to the low groups, so the SIMD that carries the heavier
    // lower wave gets the lighter upper one (18 tiles: 32 / 32 / 31 / 31 tile pairs per SIMD instead of 35 / 35 / 28 / 28)
    const int h = wave / kGroups, g = TAE_PAIR_REVERSE && h ? (kGroups - 1) - (wave & (kGroups - 1)) : (wave & (kGroups - 1));
    const int L = P.L, nb = P.nb;
    const int pad = GeoH<U>::TAIL20 ? 2 : (P.taps >> 1);       // TAIL20 instantiations run 5 taps only
    const int rows = nb * (L + pad) + pad;
    const PanelsH pn = carve_h<U>(smem, rows, L);
    // workgroups [0, n_full) own nb blocks, the tail workgroups nb_tail (fewer position tiles: a cheaper last round)
    const bool tail = P.n_full >= 0 && (int)blockIdx.x >= P.n_full;
    const int blk0 = tail ? P.n_full * nb + ((int)blockIdx.x - P.n_full) * P.nb_tail : (int)blockIdx.x * nb;
    const int nblk = min(tail ? P.nb_tail : nb, P.B - blk0);
    const int npos = nblk * L;

    zero_lds(smem, P.lds_bytes, tid);
    __syncthreads();
    if (tid == 0) range_park(pn.RNG, P.flags, P.cal);
    for (int i = tid; i < L; i += kThreads) { pn.PERM[i] = P.perm[i]; pn.INV[i] = P.inv[i]; }
    __syncthreads();
    // r_sys, r_par1 -> XA ch 0,1 (natural order); r_sys_int, r_par2 -> XB ch 0,1 (decoders.py:221-224)
    const float* rx = P.in + (size_t)blk0 * L * 3;
    float vmax = 0.0f;
    const float xs = P.x_scale;
    for (int m = tid; m < npos; m += kThreads) {
        const int b = m / L, t = m - b * L;
        const int row = b * (L + pad) + pad + t;
        const float* r = rx + (size_t)m * 3;
        const float r0 = r[0] * xs, r1 = r[1] * xs, r2 = r[2] * xs, ri = rx[((size_t)b * L + pn.PERM[t]) * 3 + 0] * xs;
        vmax = track_abs(track_abs(track_abs(vmax, r0), r1), r2);       // ri is some position's r0: covered
        pn.XA.write(row, 0, r0);
        pn.XA.write(row, 1, r1);
        pn.XB.write(row, 0, ri);
        pn.XB.write(row, 1, r2);
    }
    report_range_x(vmax, P.x_inv, P.flags, P.cal ? P.cal + P.cal_r : nullptr);     // the received values have their own slot: x_low is relative to them
    lds_max_bits(pn.RNG + 4, vmax);
    __syncthreads();
    if (tid == 0) range_check_inputs(pn.RNG + 4, P.x_low, P.flags);

    // The workgroup's position tiles are dealt out evenly over the 4 position groups and a group walks only its own
    // tiles (3 blocks of 100 = 19 tiles -> 5, 5, 5, 4; 2 blocks -> 4, 3, 3, 3; 1 block -> 2, 2, 2, 1): tiles that hold no
    // block are never computed, and a small batch can be spread over more workgroups at a lower cost each (the host
    // picks blocks per workgroup per call, nb_for_batch in turboae_api_launch.hip).
    const GroupSpan gs = group_span(npos, g);
    const bool upper = __builtin_amdgcn_readfirstlane(h) != 0;
    auto run = [&](auto pt) {
        constexpr int T = decltype(pt)::value;
        TileH<T> tc;
        make_tiles_h<T>(tc, pn.ROWT, gs.gt0, lane, L, npos, pad);
        __syncthreads();
        if (!upper) dec_body_h<U, T, 0, Split<U>::CTA, TAPS, HEAD2, PROD>(P, smem, pn, tc, gs.gt0, lane, blk0);
        else dec_body_h<U, T, Split<U>::CTA, Split<U>::CTB, TAPS, HEAD2, PROD>(P, smem, pn, tc, gs.gt0, lane, blk0);
    };
    dispatch_tiles<PT>(gs.live, run);
    // every stack ends with a barrier: all rows are in place.  Row i = (stack i / n_layer, layer i % n_layer); the last layer of a stack has no panel
    if (TAE_RANGE_BOOK == 1) range_finish(pn.RNG, 2 * P.n_iter * P.n_layer, 1, [&](int i) { return plain_tail<U>(P.wpack, P.stack_stride, P.n_layer, GeoH<U>::TAIL20 ? 5 : P.taps, i); });
    if constexpr ((TAE_X & 1024) != 0) {
        __syncthreads();
        if (blockIdx.x == 0 && lane == 0) {
            const uint32_t* r = reinterpret_cast<const uint32_t*>(smem + kStampBase) + wave * 40;
            const uint32_t t0 = reinterpret_cast<const uint32_t*>(smem + kStampBase)[0];
            for (int l = 0; l < 5; ++l)
                printf("dec wave %d (group %d half %d) layer %d: top +%u  k-loop +%u  panel-free +%u  epilogue +%u  written +%u\n", wave, g, h, l,
                       r[l * 8] - t0, r[l * 8 + 1] - t0, r[l * 8 + 2] - t0, r[l * 8 + 3] - t0, r[l * 8 + 4] - t0);
        }
    }
}

// =============================================================================================
// Encoder before power normalisation: ENC_interCNN.forward (encoders.py:362-373)
template <int U, int PT, int C0, int NC, int TRACK, bool HEAD2>
__device__ __forceinline__ void enc_body_h(const FusedParams& P, char* smem, const PanelsH& pn, const TileH<PT>& tc, int g, int lane,
                                           int blk0, double& sum, double& sumsq) {
    const int L = P.L;
    float* xtx = P.out + (size_t)blk0 * L * 3;
    const int act = P.act;
    const char* wpack = reinterpret_cast<const char*>(P.wpack);
    WeightStreamH<U, C0, NC> ws;
    ws.init(wpack, P.wpack_bytes, lane);
    ws.prefetch(0);
    const uint32_t sstride = P.stack_stride;
    for (int s = 0; s < 3; ++s) {
        const XPlane Xin = (s == 2) ? pn.XB : pn.XA;
        const RangeH rg{range_rows(pn.RNG) + s * P.n_layer * 8};
        run_stack_h<U, PT, C0, NC, TRACK, HEAD2>(wpack, s * sstride, s < 2 ? (s + 1) * sstride : 0xffffffffu, P.n_layer, smem, pn, Xin, tc, g, lane, ws, rg,
                                   [&](int p, int f, float v) {
            if (f == 0) {
                v = act_apply(v, act);                                     // enc_act (encoders.py:364)
                xtx[(size_t)(tc.blk(p) * L + tc.t(p)) * 3 + s] = v;        // x_p2 stays in interleaved order (encoders.py:371-373)
                sum += (double)v;
                sumsq += (double)v * (double)v;
            }
        }, 1);
    }
}

// TRACK = 0: no range bookkeeping (2: all of it, the last layers' maxima included).  The encoder's inputs are bit patterns - the calibration batch samples exactly the
// distribution every later call draws from - so once a handle is calibrated its encoder panels cannot leave their window unless the
// weights change; the host launches the tracking instantiation only for calibration passes and for uncalibrated handles.
template <int U, int PT, int TRACK, bool HEAD2 = (TRACK == 2)>
__global__ __launch_bounds__(kThreads, 2) void enc_kernel_h(FusedParams P) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // wave w sits on SIMD w % 4; the lower channel half (4 of 7 channel tiles) of position group s runs on SIMD s, the upper half
    // (3 tiles) of group 3 - s: group_span deals the extra position tiles to the low groups, so the SIMD that carries the heavier
    // lower wave gets the lighter upper one (18 tiles: 32 / 32 / 31 / 31 tile pairs per SIMD instead of 35 / 35 / 28 / 28)
    const int h = wave / kGroups, g = TAE_PAIR_REVERSE && h ? (kGroups - 1) - (wave & (kGroups - 1)) : (wave & (kGroups - 1));
    const int L = P.L, nb = P.nb;
    const int pad = GeoH<U>::TAIL20 ? 2 : (P.taps >> 1);       // TAIL20 instantiations run 5 taps only
    const int rows = nb * (L + pad) + pad;
    const PanelsH pn = carve_h<U>(smem, rows, L);
    // workgroups [0, n_full) own nb blocks, the tail workgroups nb_tail (fewer position tiles: a cheaper last round)
    const bool tail = P.n_full >= 0 && (int)blockIdx.x >= P.n_full;
    const int blk0 = tail ? P.n_full * nb + ((int)blockIdx.x - P.n_full) * P.nb_tail : (int)blockIdx.x * nb;
    const int nblk = min(tail ? P.nb_tail : nb, P.B - blk0);
    const int npos = nblk * L;

    zero_lds(smem, P.lds_bytes, tid);
    __syncthreads();
    if (tid == 0) range_park(pn.RNG, P.flags, P.cal);
    for (int i = tid; i < L; i += kThreads) { pn.PERM[i] = P.perm[i]; pn.INV[i] = P.inv[i]; }
    __syncthreads();
    // inputs = 2u - 1 (encoders.py:362); XB holds the interleaved copy (encoders.py:369)
    const float* u = P.in + (size_t)blk0 * L;
    for (int m = tid; m < npos; m += kThreads) {
        const int b = m / L, t = m - b * L;
        const int row = b * (L + pad) + pad + t;
        const float va = 2.0f * u[m] - 1.0f, vb = 2.0f * u[b * L + pn.PERM[t]] - 1.0f;
        pn.XA.write(row, 0, va);
        pn.XB.write(row, 0, vb);
        fold_enc_input(pn.XA, row, t, pad, 0, va);
        fold_enc_input(pn.XB, row, t, pad, 0, vb);
    }
    __syncthreads();

    double sum = 0.0, sumsq = 0.0;
    const GroupSpan gs = group_span(npos, g);          // see dec_kernel_h
    const bool upper = __builtin_amdgcn_readfirstlane(h) != 0;
    auto run = [&](auto pt) {
        constexpr int T = decltype(pt)::value;
        TileH<T> tc;
        make_tiles_h<T>(tc, pn.ROWT, gs.gt0, lane, L, npos, pad);
        __syncthreads();
        if (!upper) enc_body_h<U, T, 0, Split<U>::CTA, TRACK, HEAD2>(P, smem, pn, tc, gs.gt0, lane, blk0, sum, sumsq);
        else enc_body_h<U, T, Split<U>::CTA, Split<U>::CTB, TRACK, HEAD2>(P, smem, pn, tc, gs.gt0, lane, blk0, sum, sumsq);
    };
    dispatch_tiles<PT>(gs.live, run);
    if (TAE_RANGE_BOOK == 1 && TRACK) range_finish(pn.RNG, 3 * P.n_layer, 1, [&](int i) { return plain_tail<U>(P.wpack, P.stack_stride, P.n_layer, GeoH<U>::TAIL20 ? 5 : P.taps, i); });
    block_reduce_stats(smem, tid, sum, sumsq, P.partials);
}


// =============================================================================================
// Long blocks (block_len > 320): one stack per launch over (block, segment) workgroups with halo recompute -
// the f16x2 twin of seg_kernel in turboae_kernels.hip (same segment geometry, same exchange buffers; the
// fp32 extrinsic values are split into halves when they are staged into the X planes).
template <int U, int PT, int C0, int NC, bool DENSE>
__device__ __forceinline__ void seg_body_h(const SegParams& P, char* smem, const PanelsH& pn, const TileH<PT>& tc, int g, int lane,
                                           int stack, int b, int tstart, bool active, double& sum, double& sumsq) {
    const int L = P.L;
    const char* wpack = reinterpret_cast<const char*>(P.wpack);
    WeightStreamH<U, C0, NC> ws;
    ws.init(wpack, P.wpack_bytes, lane);
    const uint32_t soff = (uint32_t)stack * P.stack_stride;
    if (!DENSE) ws.prefetch(soff);
    const XPlane X = pn.XA;
    const RangeH rg{range_rows(pn.RNG)};
    auto run = [&](auto epi) {
        if constexpr (DENSE) run_stack_h_dense<U, PT, C0, NC>(wpack, soff, P.n_layer, smem, pn, X, tc, g, lane, ws, rg, active, epi);
        else run_stack_h<U, PT, C0, NC, 1>(wpack, soff, 0xffffffffu, P.n_layer, smem, pn, X, tc, g, lane, ws, rg, epi, P.mode == 0 ? 1 : 0);     // (long blocks: the last layers keep exp2 - 1)
    };
    if (P.mode == 0) {
        const int act = P.act;
        float* xtx = P.out + (size_t)b * L * 3;
        run([&](int p, int f, float v) {
            if (f == 0) {
                v = act_apply(v, act);
                xtx[(size_t)(tstart + tc.m0 + 16 * p) * 3 + stack] = v;
                sum += (double)v;
                sumsq += (double)v * (double)v;
            }
        });
    } else if (!P.last) {
        const int F = P.F;
        const bool extrinsic = P.extrinsic != 0;
        const float xinv = 1.0f / P.x_scale[0];
        float* ecur = P.ecur + (size_t)b * L * 8;
        run([&](int p, int f, float v) {
            if (f < F) {
                if (extrinsic) v -= X.read(tc.row(p), 2 + f) * xinv;
                ecur[(size_t)(tstart + tc.m0 + 16 * p) * 8 + f] = v;      // fp32 in HBM: scaled (and range-checked) when the next launch stages it
            }
        });
    } else {
        float* xdec = P.out + (size_t)b * L;
        run([&](int p, int f, float v) {
            if (f == 0) xdec[P.perm[tstart + tc.m0 + 16 * p]] = 1.0f / (1.0f + expf(-v));    // sigmoid(deinterleave), decoders.py:267
        });
    }
}

template <int U>
__device__ __forceinline__ PanelsH carve_seg_h(char* smem, int rows, int npanel = 1) {
    PanelsH pn;
    pn.dump = rows + 1;
    const size_t ab = (size_t)(rows + 2) * U * 2, xb = (size_t)(rows + 1 + kXSlack) * kXRowB;
    pn.AH = smem;
    pn.AL = pn.AH + ab;
    pn.panel_bytes = (uint32_t)(2 * ab);
    pn.XA.h = smem + (size_t)npanel * 2 * ab;
    pn.XA.l = pn.XA.h + xb;
    pn.XB = pn.XA;
    pn.PERM = nullptr;
    pn.INV = nullptr;
    pn.ROWT = reinterpret_cast<int*>(pn.XA.l + xb);
    pn.HS = reinterpret_cast<float*>(smem + (((reinterpret_cast<char*>(pn.ROWT + kHeadSlots) - smem) + 15) & ~15));
    pn.RNG = reinterpret_cast<uint32_t*>(pn.HS + kHeadSlots * 8);
    return pn;
}

template <int U, int PT>
__global__ __launch_bounds__(kThreads, 2) void seg_kernel_h(SegParams P) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // wave w sits on SIMD w % 4; the lower channel half (4 of 7 channel tiles) of position group s runs on SIMD s, the upper half
    // (3 tiles) of group 3 - s: group_span deals the extra position tiles to the low groups, so the SIMD that carries the heavier
    // lower wave gets the lighter upper one (18 tiles: 32 / 32 / 31 / 31 tile pairs per SIMD instead of 35 / 35 / 28 / 28)
    const int h = wave / kGroups, g = TAE_PAIR_REVERSE && h ? (kGroups - 1) - (wave & (kGroups - 1)) : (wave & (kGroups - 1));
    const int pad = GeoH<U>::TAIL20 ? 2 : (P.taps >> 1);                       // 2 for the dense stacks (5 taps) and for the 5-tap-only 100-wide kernels
    const int L = P.L, H = pad * P.n_layer;
    int bid = blockIdx.x;
    int stack = P.stack;
    if (P.mode == 0) { stack = bid % 3; bid /= 3; }
    const int seg = bid % P.nseg, b = bid / P.nseg;
    // segment 0 owns T0 centre positions, the others T (the last one what is left); only positions inside the block are
    // walked (group_span deals NP): no halo in front of position 0 or behind position L - 1
    const int s0 = seg == 0 ? 0 : P.T0 + (seg - 1) * P.T;
    const int tlen = min(seg == 0 ? P.T0 : P.T, L - s0);
    const int tstart = max(s0 - H, 0) & ~3;            // same panel origin as the fp32 kernel (floored to a multiple of 4)
    const int NP = min(s0 + tlen + H, L) - tstart;
    const int rows = P.T + 2 * H + 3 + 2 * pad;
    const PanelsH pn = carve_seg_h<U>(smem, rows, P.dense ? (P.n_layer > 1 ? P.n_layer - 1 : 1) : 1);
    const bool odd = (stack & 1) != 0;

    zero_lds(smem, P.lds_bytes, tid);
    __syncthreads();
    if (tid == 0) range_park(pn.RNG, P.flags, P.cal);
    float vmax = 0.0f, emax = 0.0f;
    const float xs = P.x_scale[P.mode == 0 ? stack : 0];          // the X planes hold value * 2^A_x (per stack on this path)
    for (int m = tid; m < NP; m += kThreads) {
        const int t = tstart + m;
        if (t < 0 || t >= L) continue;
        const int row = pad + m;
        if (P.mode == 0) {
            const int src = (stack == 2) ? P.perm[t] : t;                           // encoders.py:369
            const float v = (2.0f * P.in[(size_t)b * L + src] - 1.0f) * xs;         // encoders.py:362
            pn.XA.write(row, 0, v);
            if (!P.dense) fold_enc_input(pn.XA, row, t, pad, 0, v);
        } else {
            const float* rx = P.in + (size_t)b * L * 3;
            const float r0 = (odd ? rx[(size_t)P.perm[t] * 3] : rx[(size_t)t * 3]) * xs;   // r_sys_int / r_sys
            const float r1 = rx[(size_t)t * 3 + (odd ? 2 : 1)] * xs;                       // r_par2 / r_par1
            vmax = track_abs(track_abs(vmax, r0), r1);
            pn.XA.write(row, 0, r0);
            pn.XA.write(row, 1, r1);
            if (stack > 0) {
                // dec2 reads q[p[i]] (interleave, decoders.py:238); dec1 reads q2[inv[j]] (deinterleave, :249)
                const int gi = odd ? P.perm[t] : P.inv[t];
                const float* e = P.eprev + ((size_t)b * L + gi) * 8;
                for (int f = 0; f < P.F; ++f) {
                    const float ev = e[f] * xs;
                    emax = track_abs(emax, ev);
                    pn.XA.write(row, 2 + f, ev);
                }
            }
        }
    }
    if (P.mode != 0) {
        report_range_x(vmax, 1.0f / xs, P.flags, P.cal ? P.cal + P.cal_r : nullptr);      // received values
        report_range_x(emax, 1.0f / xs, P.flags, P.cal ? P.cal + P.cal_x : nullptr);      // the previous stack's extrinsic values
        lds_max_bits(pn.RNG + 4, vmax);
        __syncthreads();
        if (tid == 0) range_check_inputs(pn.RNG + 4, P.x_low, P.flags);
    }

    // tiles of the wave's position group: panel rows [2 + m] of the segment; `center` marks the positions this workgroup owns
    auto make_tiles = [&](auto& tc, int gt0) {
        using TC = std::remove_reference_t<decltype(tc)>;
        const int n = lane & 15;
        tc.valid = 0u;
        tc.center = 0u;
        tc.m0 = gt0 * 16 + n;
        tc.L = L;
        tc.pad = pad;
        tc.rowtab = pn.ROWT + gt0 * 16 + n;
#pragma unroll
        for (int p = 0; p < TC::kTiles; ++p) {
            const int m = tc.m0 + 16 * p;
            const int t = tstart + m;
            const bool v = (m < NP) && (t >= 0) && (t < L);
            if (lane < 16) pn.ROWT[(gt0 + p) * 16 + n] = v ? pad + m : pad;
            tc.valid |= (v ? 1u : 0u) << p;
            tc.center |= ((v && t >= s0 && t < s0 + tlen) ? 1u : 0u) << p;
        }
    };
    double sum = 0.0, sumsq = 0.0;
    const bool upper = __builtin_amdgcn_readfirstlane(h) != 0;
    if (P.dense) {
        // every earlier layer's output stays resident: a dense segment is one group's PT tiles at most, walked in full
        TileH<PT> tc;
        make_tiles(tc, g * PT);
        __syncthreads();
        const bool active = __builtin_amdgcn_readfirstlane((NP + 15) / 16 - g * PT) > 0;      // the group has at least one live tile
        if (!upper) seg_body_h<U, PT, 0, Split<U>::CTA, true>(P, smem, pn, tc, g * PT, lane, stack, b, tstart, active, sum, sumsq);
        else seg_body_h<U, PT, Split<U>::CTA, Split<U>::CTB, true>(P, smem, pn, tc, g * PT, lane, stack, b, tstart, active, sum, sumsq);
    } else {
        // even deal of the segment's tiles over the position groups (L = 1000: 4 segments of 250 + 2 x 10 halo = 18 tiles -> 5, 5, 4, 4)
        const GroupSpan gs = group_span(NP, g);
        auto run = [&](auto pt) {
            constexpr int T = decltype(pt)::value;
            TileH<T> tc;
            make_tiles(tc, gs.gt0);
            __syncthreads();
            if (!upper) seg_body_h<U, T, 0, Split<U>::CTA, false>(P, smem, pn, tc, gs.gt0, lane, stack, b, tstart, true, sum, sumsq);
            else seg_body_h<U, T, Split<U>::CTA, Split<U>::CTB, false>(P, smem, pn, tc, gs.gt0, lane, stack, b, tstart, true, sum, sumsq);
        };
        dispatch_tiles<PT>(gs.live, run);
    }
    if (TAE_RANGE_BOOK == 1) {
        // one stack per workgroup: rows 0 .. n_layer - 1; calibration slots of stack `stack`
        const int nl = P.n_layer;
        if (P.dense) range_finish(pn.RNG, nl, 1 + stack * nl, [&](int i) { return dense_tail<U>(P.wpack, (uint32_t)stack * P.stack_stride, nl, i); });
        else range_finish(pn.RNG, nl, 1 + stack * nl, [&](int i) { return plain_tail<U>(P.wpack, P.stack_stride, nl, GeoH<U>::TAIL20 ? 5 : P.taps, stack * nl + i); });
    }
    if (P.mode == 0) { __syncthreads(); block_reduce_stats(smem, tid, sum, sumsq, P.partials); }
}


// launchers of one channel width (explicitly instantiated in turboae_h2.hip / turboae_h2_w.hip)
template <int U>
hipError_t launch_fused_h_u(bool decoder, const FusedParams& P, int grid, hipStream_t st) {
    constexpr int PT = 5;
    auto kd = dec_kernel_h<U, PT>;
    auto kd2 = dec_kernel_h<U, PT, false, true>;   // production twin with both expm1 branches in the heads (FusedParams::head2)
    auto kt = dec_kernel_h<U, PT, true>;       // the instantiation for tae_decode_taps (exports every stack's extrinsic outputs) and for
                                               // calibration launches (also tracks the last layers' ELU maxima)
    auto ke = enc_kernel_h<U, PT, 0>;
    auto ke2 = enc_kernel_h<U, PT, 0, true>;
    auto ket = enc_kernel_h<U, PT, 2>;
    const bool taps = decoder && (P.tap_out != nullptr || P.cal != nullptr || P.track == 2);
    const bool etrack = !decoder && (P.track != 0 || P.cal != nullptr);
    const bool h2 = P.head2 != 0;
    if constexpr (U == 100) {
        // precision = f16x1 (hi halves only, NOT fp32-grade): production decoder launches of the 100-wide kernel only
        if (decoder && !taps && P.prod == 1) {
            auto kx = dec_kernel_h<U, PT, false, false, 1>;
            hipError_t ex = hipFuncSetAttribute(reinterpret_cast<const void*>(kx), hipFuncAttributeMaxDynamicSharedMemorySize, P.lds_bytes);
            if (ex != hipSuccess) return ex;
            hipLaunchKernelGGL(kx, dim3(grid), dim3(kThreads), P.lds_bytes, st, P);
            return hipGetLastError();
        }
    } else if (P.prod == 1) {
        return hipErrorInvalidValue;
    }
    const void* fn = decoder ? (taps ? reinterpret_cast<const void*>(kt) : (h2 ? reinterpret_cast<const void*>(kd2) : reinterpret_cast<const void*>(kd)))
                             : (etrack ? reinterpret_cast<const void*>(ket) : (h2 ? reinterpret_cast<const void*>(ke2) : reinterpret_cast<const void*>(ke)));
    const int lds_launch = (TAE_X & 1024) ? 160 * 1024 : P.lds_bytes;       // experiment 1024: room for the stamps behind the panels
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds_launch);
    if (e != hipSuccess) return e;
    if (taps) hipLaunchKernelGGL(kt, dim3(grid), dim3(kThreads), lds_launch, st, P);
    else if (decoder && h2) hipLaunchKernelGGL(kd2, dim3(grid), dim3(kThreads), lds_launch, st, P);
    else if (decoder) hipLaunchKernelGGL(kd, dim3(grid), dim3(kThreads), lds_launch, st, P);
    else if (etrack) hipLaunchKernelGGL(ket, dim3(grid), dim3(kThreads), P.lds_bytes, st, P);
    else if (h2) hipLaunchKernelGGL(ke2, dim3(grid), dim3(kThreads), P.lds_bytes, st, P);
    else hipLaunchKernelGGL(ke, dim3(grid), dim3(kThreads), P.lds_bytes, st, P);
    return hipGetLastError();
}

template <int U>
hipError_t launch_seg_h_u(const SegParams& P, int grid, hipStream_t st) {
    constexpr int PT = 5;
    auto k = seg_kernel_h<U, PT>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, P.lds_bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k, dim3(grid), dim3(kThreads), P.lds_bytes, st, P);
    return hipGetLastError();
}


}  // namespace tae
