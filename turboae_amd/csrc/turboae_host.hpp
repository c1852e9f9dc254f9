// Host-side internals shared by the three translation units of the C ABI (not part of it):
//   turboae_api_create.hip  handle life cycle: configuration checks, weight packing, range calibration, tae_create / _destroy / _reserve
//   turboae_api_launch.hip  geometry (blocks per workgroup, segments) and the launch sequences of encoder / decoder (CNN, long-block, GRU)
//   turboae_api_entry.hip   error state, debug knobs, the fp32 fall-back wrapper and every other extern "C" entry point
// Everything here lives in tae::host; the kernels' own host<->device interface is turboae_internal.hpp.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <math.h>
#include <algorithm>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/turboae_hip.h"
#include "turboae_internal.hpp"

namespace tae {
namespace host {
// LSTM / RNN decoder stacks: calls below this many blocks run layer 1 as projection kernel + recurrence (the projection GEMM spreads over
// the whole chip whatever the batch), calls at or above it the fused kernel (one workgroup per 16 blocks and direction: measured r06, LSTM,
// 256 CUs: 1 000 blocks 4.8 vs 5.1 ms, 2 048 blocks 6.8 vs 5.9 ms, 16 384 blocks 53.3 vs 43.3 ms).  The forms are bit-identical.
int rnn_l1_split_below(const tae_handle* h);


int fail(int code, const std::string& msg);       // sets the calling thread's tae_last_error string, returns `code`
const char* last_error();
std::string knob_report();                        // "NAME=value;..." of the debug knobs that took effect (tae_overrides)

#define TAE_HIP(expr)                                                                              \
    do {                                                                                           \
        hipError_t e__ = (expr);                                                                   \
        if (e__ != hipSuccess) return ::tae::host::fail(TAE_EHIP, std::string(#expr) + ": " + hipGetErrorString(e__)); \
    } while (0)

// The tail of one packed layer as the host keeps it for the range calibration (calibrate_range below): the layer's accumulators carry
// 2^(S + A_in) (S: the weights' own power-of-two scale, A_in: exponent of the panel / stack inputs it reads), its ELU output is
// stored * 2^A_out.  Device tail = bias * 2^(S + A_in) [CP] | 2^-(S + A_in) | 2^A_out | low-side threshold | high-side threshold | ELU kind | 3 spare.
struct TailRef {
    uint32_t off = 0;              // byte offset of the tail inside the side's packed buffer
    int S = 0;
    std::vector<float> bias_s;     // bias * 2^S, padded to CP
};

}  // namespace host
}  // namespace tae

struct tae_handle {
    tae_config cfg;
    int device = 0;
    // per side (encoder / decoder): channel width, blocks per workgroup of the whole-block kernels (0: long-block path), LDS bytes
    int U = 0, nb = 0, lds_bytes = 0;            // encoder
    int Ud = 0, nbd = 0, lds_bytes_d = 0;        // decoder
    int ncu = 256;           // compute units of the device (workgroups resident at once: one per CU)
    bool fixed_nb = false;   // TAE_FIXED_NB=1: always nb blocks per workgroup (testing knob)
    // long-block (segmented) path, used when a whole block does not fit one workgroup (nb == 0)
    int enc_T = 0, enc_nseg = 0, enc_lds = 0, dec_T = 0, dec_nseg = 0, dec_lds = 0;
    int enc_T0 = 0, dec_T0 = 0;      // centre length of segment 0 (no left halo: up to H + 3 more than the others)
    uint32_t enc_stride = 0, dec_stride = 0;
    uint32_t enc_bytes = 0, dec_bytes = 0;
    int super = 0, super_d = 0;   // remainder channels via super-tiles (U % 16 == 4 and block_len % 4 == 0), encoder / decoder
    // f16x2 representation of the whole-block kernels (prec == 1); the fp32 packs above stay resident for the long-block path
    int prec = 0;            // 0: v_mfma_f32_16x16x4_f32 on fp32 operands; 1: 3 x v_mfma_f32_16x16x32_f16 on hi/lo halves
    int lds_bytes_h = 0, lds_bytes_hd = 0, enc_lds_h = 0, dec_lds_h = 0;
    uint32_t enc_stride_h = 0, dec_stride_h = 0, enc_bytes_h = 0, dec_bytes_h = 0;
    char* d_wenc_h = nullptr;
    char* d_wdec_h = nullptr;
    uint32_t* d_flags = nullptr;   // bit 0: an activation left the fp16 range (f16x2 kernels clamp and report)
    float* d_wenc = nullptr;
    float* d_wdec = nullptr;
    int32_t* d_perm = nullptr;
    int32_t* d_inv = nullptr;
    // workspace
    int32_t cap = 0;
    float* d_xtx = nullptr;
    float* d_rx = nullptr;
    double* d_partials = nullptr;
    double* d_stats = nullptr;
    float* d_e0 = nullptr;   // long-block path: extrinsic exchange buffers (B, L, 8)
    float* d_e1 = nullptr;
    // GRU decoder (dec_type = 1): canonical decoder weights uploaded as they are + per-chunk workspace
    float* d_wrnn = nullptr;
    char* d_wrnn_h = nullptr;   // f16x2 packing of the same (prec == 1)
    float* d_wernn = nullptr;   // GRU encoder (enc_type = 1): the three ENC_interRNN stacks, packed like the decoder's
    char* d_wernn_h = nullptr;
    // tae_eval_snr workspace (grown on demand)
    float* d_eval_u = nullptr;       // bits of one decode group (kept for the error count)
    float* d_eval_noise = nullptr;   // noise of one batch
    float* d_eval_xdec = nullptr;    // decisions of one decode group
    int64_t eval_group_blocks = 0;
    int32_t eval_batch = 0;
    bool eval_noise_x2 = false;      // d_eval_noise holds fading coefficients + noise
    double* d_rnn_partials = nullptr;   // GRU encoder: per (chunk, stack, head workgroup) partial sums
    int32_t rnn_partial_slots = 0;
    bool x1 = false;           // precision = TAE_PREC_F16X1: decoder launches take the one-product instantiation (hi halves only; not fp32-grade)
    bool rnn_l1_check = false; // debug: run rnn_proj_u beside the fused layer-1 kernel (its GI is what a -DTAE_L1F_DBG_GI build compares against)
    int rnn_l1_mode = 0;       // LSTM / RNN stacks, layer 1: 0 = by batch (below), 1 = always rnn_proj_u + rnn_rec_u (r05 pair), 2 = always rnn_l1f_u_kernel
                               // (the two forms are bit-identical: results never depend on the choice; debug knob TAE_RNN_L1=split|fused)
    bool gru_l1_split = false; // f16x2 GRU stacks: layer 1 as projection kernel + recurrence kernel (r04) instead of the fused kernel
    int gru_l0_mode = 0;         // layer 0 of the f16x2 GRU decoder stacks: 0 by batch (unit-split twin up to kGruL0UnitMaxB blocks), 1 block-split, 2 unit-split (TAE_GRU_L0, debug knob)
    int32_t rnn_chunk = 0;   // blocks per internal chunk (bounds the workspace)
    float* d_gxa = nullptr;  // (chunk, L, 8) natural-order panel
    float* d_gxb = nullptr;  // (chunk, L, 8) interleaved-order panel
    float* d_gy0 = nullptr;  // (chunk, L, 2H) layer-0 outputs
    float* d_gy1 = nullptr;  // (chunk, L, 2H) layer-1 outputs
    float* d_ggi = nullptr;  // (chunk, L, 2, 19, 16) layer-1 input projections in gate-tile order
    int enc_gates = 3;           // recurrent encoder cell (ENC_interRNN), as dec_gates
    char* d_wernn_u = nullptr;   // ... its three stacks packed for turboae_rnn_u.hip (enc_gates != 3)
    std::vector<float> rnn_u_gimul_enc;
    int dec_gates = 3;           // recurrent decoder cell: 3 GRU (turboae_gru*.hip), 4 LSTM / 1 vanilla RNN (turboae_rnn_u.hip, f16x2 only)
    char* d_wrnn_u = nullptr;    // LSTM / RNN decoder stacks in the unit-split layouts
    std::vector<float> rnn_u_gimul;   // [stack][dir]: scale the projection kernel folds into GI (the layer-1 recurrence's own 2^S)
    tae::NormOpts nopts;       // encoder-output / channel variant (tae_set_channel_opts)
    tae_noise_opts noise_opts; // generator tae_eval_snr draws from (tae_set_noise_opts; default AWGN)
    tae::GenericEngine* gen = nullptr;   // generic fp32 kernels (configurations outside the MFMA kernels' envelope)
    // ---- range calibration of the fp16-split conv kernels (calibrate_range): per-layer activation exponents
    std::vector<tae::host::TailRef> enc_tails, dec_tails;      // [stack * n_layer + l]; empty: the side has no fp16-split conv stacks
    std::vector<int> enc_A, dec_A;                  // exponent of every layer's OUTPUT panel ([stack * n_layer + l]; unused for the last layer)
    std::vector<int> enc_Ax, dec_Ax;                // exponent of every stack's input planes (whole-block decoder: all equal)
    std::vector<float> enc_low, dec_low;            // low-side threshold per layer (0: not checked)
    std::vector<float> enc_high, dec_high;          // high-side threshold per layer: 65504, or 2^-3 * 2^A for a layer whose ELU runs as a polynomial
    std::vector<int> enc_kind, dec_kind;            // ELU branch per layer (turboae_h2.hip, TAE_ELU_MODE 2): 0 exp2, 1 polynomial, 2 both
    float dec_r_low = 0.0f;                         // 2^-7 of the largest received value of the calibration batch (0: not checked); x_low = this * 2^A_x
    int enc_min_values = 0, dec_min_values = 0;     // values of one panel a workgroup holds at least (positions x real channels)
    uint32_t* d_cal = nullptr;                      // calibration launches: per-layer / per-stack maxima (float bits), encoder then decoder
    bool calibrating = false;
    bool calibrated = false;
    bool cal_user = false;     // the current exponents were measured on the caller's data (tae_calibrate_range(u, noise)): never replaced silently
    bool cal_perm = false;     // ... on the synthetic batch with an installed (post-create) permutation
    int cal_passes = 0;
    // ---- tae_config.range_fallback: fp32 twin of this handle and what the last flagged call did
    tae_handle* fb = nullptr;
    uint32_t last_flags = 0;                         // range bits accumulated since the last tae_range_status (bit 2: a call was re-run in fp32)
};

namespace tae {
namespace host {

// ---- packed-weight geometry the launch code needs (the packing itself: turboae_api_create.hip)
constexpr int kGH = 100, kGRT = 19, kGKP = 13;
constexpr size_t kGRecF = (size_t)kGRT * kGKP * 128, kGXF = (size_t)kGRT * 128, kGB0 = 25 * 16, kGB1 = 7 * 16;
constexpr size_t kGProjF = 2 * 25 * (size_t)kGRT * 128, kGPB = 2 * (size_t)kGRT * 16;
constexpr size_t kGL0Dir = kGRecF + kGXF + kGB0, kGL1Dir = kGRecF + kGB1;
constexpr size_t kGHTileB = 3 * 2048 + 1024, kGHFragB = 19 * kGHTileB, kGHNiB = 6 * 1024;
constexpr size_t kGHRec0B = kGHFragB + kGHNiB + 25 * 64 + 16, kGHRec1B = kGHFragB + kGHTileB + 7 * 64 + 16;
constexpr int32_t kGruL0UnitMaxB = 12288;    // blocks per chunk up to which layer 0 of the f16x2 GRU decoder runs on gru_rec0u_kernel (equal at 16 384, faster below: LABNOTES 11.11)
constexpr size_t kGHProjDirB = 7 * 19 * 2048, kGHProjB = 2 * kGHProjDirB + 2 * 19 * 64 + 16;
// gates of a recurrent cell (tae_config.enc_rnn / dec_rnn): GRU 3 (r, z, n), LSTM 4 (i, f, g, o), vanilla RNN 1
inline int cell_gates(int rnn) { return rnn == 1 ? 4 : (rnn == 2 ? 1 : 3); }
size_t rnn_u_stack_bytes(size_t nout, int G);     // packed LSTM / RNN decoder stack (turboae_rnn_u.hip layouts)
size_t rnn_packed_stack_floats(size_t nout);
size_t rnn_h_stack_bytes(size_t nout);
size_t rnn_h_l1f_offset(size_t nout);
// Instantiated kernel widths / the width a configured one runs at (0: too wide)
inline int kernel_width(int u) { return u <= 32 ? 32 : (u <= 64 ? 64 : (u <= 100 ? 100 : (u <= 124 ? 124 : 0))); }

// fp32 <-> fp16 bit patterns, round to nearest even, denormals kept (the host-side operand split; tae_debug_split_f16 exposes it to the tests)
inline uint16_t f2h(float f) {
    uint32_t x;
    memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    x &= 0x7fffffffu;
    if (x >= 0x7f800000u) return (uint16_t)(sign | 0x7c00u | (x > 0x7f800000u ? 0x200u : 0));
    if (x >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);                 // rounds to >= 65520 -> inf
    if (x < 0x33000001u) return (uint16_t)sign;                               // <= 2^-25 -> 0 (ties to even)
    int e = (int)(x >> 23) - 127;
    uint32_t m = (x & 0x7fffffu) | 0x800000u;
    int shift = e >= -14 ? 13 : 13 + (-14 - e);                               // denormal: more bits dropped
    const uint32_t half = 1u << (shift - 1), rest = m & ((1u << shift) - 1);
    uint32_t r = m >> shift;
    if (rest > half || (rest == half && (r & 1u))) ++r;
    if (e >= -14) return (uint16_t)(sign | (uint32_t)(((e + 15) << 10) + (r - 0x400u)));   // carry propagates into the exponent
    return (uint16_t)(sign | r);
}
inline float h2f(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    const int e = (h >> 10) & 0x1f;
    const uint32_t m = h & 0x3ffu;
    float v;
    if (e == 0) v = ldexpf((float)m, -24);
    else if (e == 31) v = m ? NAN : INFINITY;
    else v = ldexpf((float)(m | 0x400u), e - 25);
    return sign ? -v : v;
}


// ---- calibration array layout and window constants (tae_handle::d_cal; see turboae_api_launch.hip)
constexpr float kRangeLow = 8.0f;        // low end of the window in scaled units (packed into the tails / launch parameters)
constexpr int kRangeTarget = 11;         // calibrated maxima land in [2^10, 2^11)
constexpr int kRangeMinValues = 2048;    // panels with fewer values per workgroup are not low-checked
size_t cal_dec_offset(const tae_handle* h);
size_t cal_dec_r(const tae_handle* h);
size_t cal_words(const tae_handle* h);

// ---- turboae_api_create.hip
int check_cfg(const tae_config* c);
size_t num_weights(const tae_config* c);
int calibrate_range(tae_handle* h, const float* u_user, const float* noise_user, int32_t B_user);

// ---- turboae_api_launch.hip
int choose_nb(int U, int L, int* lds_out, bool h2 = false, int taps = 5, int range_layers = 0);
bool choose_seg(int U, int L, int n_layer, int* T, int* T0, int* nseg, int* lds, bool dense = false, bool h2 = false, int taps = 5);
int check_handle(tae_handle* h);
int check_batch(tae_handle* h, int32_t B);
tae_noise_opts default_noise_opts();
int check_noise_opts(const tae_noise_opts* o);
int make_noise_gen(const tae_noise_opts* o, float test_sigma, tae::NoiseGen* g);
tae::NormOpts default_norm_opts();
tae::FusedParams base_params(const tae_handle* h, int32_t B, bool decoder);
int run_encoder(tae_handle* h, const float* u, float* xtx, double* stats, int32_t B, hipStream_t st);
int run_decoder(tae_handle* h, const float* rx, float* xdec, int32_t B, hipStream_t st, float* tap_out = nullptr);

}  // namespace host
}  // namespace tae
