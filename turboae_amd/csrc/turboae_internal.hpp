// Internal host<->kernel interface of libturboae_hip.so (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include <algorithm>
#include "../../include/turboae_hip.h"

namespace tae {

// Arguments of the fused encoder / decoder kernels (one workgroup = nb whole codeword blocks).
struct FusedParams {
    const float* wpack;     // packed weights of this network (encoder: 3 stacks, decoder: 2*n_iter stacks)
    const int32_t* perm;    // p[L]   (interleavers.py:15-21)
    const int32_t* inv;     // inv[L] (interleavers.py:29-33)
    const float* in;        // encoder: u (B,L,1); decoder: received (B,L,3)
    float* out;             // encoder: x_tx (B,L,3) before power_constraint; decoder: x_dec (B,L,1)
    double* partials;       // encoder: [grid][2] = (sum, sumsq) of x_tx per workgroup
    int32_t B, L, nb;       // batch, block_len, blocks per workgroup
    int32_t n_layer;        // conv layers per stack
    int32_t n_iter;         // decoder iterations
    int32_t F;              // num_iter_ft
    int32_t extrinsic;
    int32_t act;            // encoder output activation (act_apply codes: 0 elu, 1 linear, 2 tanh, 3 relu, 4 selu, 5 sigmoid)
    int32_t taps;           // f16x2 kernels: conv kernel size of this side (5, 7 or 9; 1 and 3 arrive embedded in 5)
    uint32_t stack_stride;  // floats between consecutive stacks in wpack
    uint32_t wpack_bytes;   // size of the packed weight buffer (buffer-resource bound)
    int32_t lds_bytes;
    int32_t super;          // 1: remainder channels via super-tiles (needs U % 16 == 4 and block_len % 4 == 0)
    uint32_t* flags;        // f16x2 kernels: bit 0 set when an activation left the fp16 range (stack_stride is in BYTES there)
    int32_t n_full, nb_tail; // f16x2 whole-block kernels: workgroups [0, n_full) own nb blocks each, the rest nb_tail each (the last
                            // partial round of workgroups runs thinner, see tail_geometry in turboae_api_launch.hip); n_full < 0: all own nb
    float* tap_out;         // decoder, debug instantiation only (tae_decode_taps): [2*n_iter-1][B][L][F] extrinsic outputs of every
                            // non-final stack in the producing stack's own position order (before the (de)interleave scatter)
    // f16x2 kernels: the stack-input planes hold value * x_scale (a power of two; x_inv = 1 / x_scale); a workgroup whose largest
    // staged input * x_scale is below x_low raises flags bit 1 (0 disables: encoder inputs are +-1)
    float x_scale, x_inv, x_low;
    uint32_t* cal;          // calibration launches only (else nullptr): float bits, atomicMax'ed - [0] max |extrinsic value| a stack handed on
                            // (unscaled), [1 + stack * n_layer + l] max |ELU output| of layer l, [cal_r] max |received value|
    int32_t cal_r;
    int32_t track;          // which instantiation of the f16x2 whole-block kernels: 0 = no range bookkeeping (calibrated encoders), 1 = the
                            // panels (production decoder), 2 = full: + the last layers' maxima, both expm1 branches in the heads (calibration
                            // launches, uncalibrated encoders)
    int32_t head2;          // f16x2 whole-block kernels, track < 2: 1 = the instantiation whose Linear heads evaluate both expm1 branches (one of
                            // this side's last conv layers stays below 1/4: exp2 - 1 alone is not relatively accurate there)
    int32_t prod;           // f16x2 whole-block DECODER: 1 = the one-product instantiation (precision = TAE_PREC_F16X1: hi halves only, not
                            // fp32-grade); anything else = the three products of the f16x2 scheme
};

// Arguments of the per-stack segmented kernel used when a block does not fit one workgroup.
struct SegParams {
    const float* wpack;     // packed weights of the network
    const int32_t* perm;
    const int32_t* inv;
    const float* in;        // encoder: u (B,L,1); decoder: received (B,L,3)
    const float* eprev;     // decoder: previous stack's extrinsic outputs (B,L,8), own-domain order
    float* ecur;            // decoder: this stack's extrinsic outputs (B,L,8)
    float* out;             // encoder: x_tx (B,L,3); last decoder stack: x_dec (B,L,1)
    double* partials;       // encoder: [grid][2]
    int32_t mode;           // 0 = encoder (3 stacks in one launch), 1 = decoder (one stack per launch)
    int32_t stack;          // decoder stack index 0..2*n_iter-1
    int32_t last;           // decoder: final half-iteration
    int32_t B, L, T, nseg;  // T = centre positions per segment, nseg = segments per block
    int32_t T0;             // centre positions of segment 0 (it needs no left halo, so it may own up to H + 3 more than the others)
    int32_t n_layer, F, extrinsic, act;
    int32_t taps;           // f16x2 kernel: conv kernel size (see FusedParams)
    uint32_t stack_stride;
    uint32_t wpack_bytes;
    int32_t lds_bytes;
    int32_t super;
    uint32_t* flags;        // f16x2 kernels: range flag (stack_stride is in BYTES there)
    int32_t dense;          // 1: DenseSameShapeConv1d stacks (cnn_utils.py:49-82), f16x2 long-block kernels only
    float x_scale[3];       // f16x2 kernels: scale of the stack-input planes; decoder: [0] = this launch's stack, encoder: per stack
    float x_low;            // decoder: low end of the window for the staged inputs (see FusedParams)
    uint32_t* cal;          // calibration launches only: as FusedParams::cal, with [cal_x] = max |extrinsic value| this decoder stack staged
    int32_t cal_x, cal_r;
};

// ---- GRU decoder (turboae_gru.hip)
struct GruRecParams {
    const float* w;        // per direction: recurrent A fragments | (layer 0: input A fragments | bias rows) or (layer 1: b_hn rows)
    uint32_t w_dir_stride; // floats between the two directions' blocks
    const float* x;        // layer 0: input panel (B,L,8)
    const float* gi;       // layer 1: input projections (B,L,2,19,16) in gate-tile order
    float* y;              // (B,L,2H); f16x2 kernels: layer 0 only (halves)
    float* hpart;          // f16x2 layer 1: (npos', 2, 8) this direction's share of the Linear head (no Y1 is written)
    int32_t B, L;
};
struct GruProjParams {
    const float* yin;     // (npos, 2H)
    const float* w;       // A fragments (2 dirs x 25 chunks x 19 tiles) followed by the bias rows (2 x 19 x 16)
    float* gi;            // (npos,2,19,16); f16x2 kernels: time-major (L,2,B,19,16)
    size_t npos;
    int32_t B, L;         // f16x2 kernels: blocks in this chunk and block length (npos = B * L)
};
struct GruHeadParams {
    const float* y;       // (npos, 2H); launch_gru_head_part: the (npos', 2, 8) per-direction head products of the layer-1 recurrence
    const float* w;       // (nout, 2H)
    const float* b;       // (nout)
    const float* xcur;    // this stack's input panel (extrinsic subtraction)
    float* xnext;         // the other panel (scatter target), or nullptr when last
    float* xdec;          // last: (B,L,1)
    const int32_t* ptab;  // inv (after dec1) or perm (after dec2)
    size_t npos;
    int32_t L, F, nout, extrinsic, last;
    int32_t grouped, B;   // grouped = 1: rows of y are in block-group-major order (f16x2 GRU path), B = blocks
    // encoder mode (ENC_interRNN, enc_stack >= 0): output = enc_act(Linear) -> xtx[(b, t), enc_stack], per-workgroup (sum, sumsq);
    // `act` = enc_act there, dec_act in decoder mode (act_apply codes)
    int32_t enc_stack, act;
    float* xtx;           // (B, L, 3)
    double* partials;     // [gridDim.x][2]
    float* tap;           // tae_decode_taps: this stack's extrinsic output [b][position of this stack's order][f], or nullptr
};
// f16x2 layer 1 as one kernel (turboae_gru_l1f.hip): input projection + recurrence + this direction's half of the Linear head.
// Weight image of one direction: 6 unit-wave register images (42 fragments of 1 KB: per gate W_hh {slab 0..2: hi, lo; remainder},
// then per gate W_ih1 hi {slab 0..5; remainder}) | remainder-wave image (27 fragments: W_hh 7, W_ih1 13, head 7) | W_ih1 lo of
// k-slabs 0, 1 [ut][gate][slab] (unit waves' registers) | LDS image (W_ih1 lo of k-slabs 2..5 [ut][gate][slab - 2] | bias rows |
// 2^-S, 2^-S_head).
struct GruL1fLayout {
    static constexpr int kUnitB = 42 * 1024, kRemB = 27 * 1024, kLo01B = 6 * 3 * 2 * 1024;
    static constexpr int kLdsImgB = 6 * 3 * 4 * 1024 + 6 * 4 * 64 + 64 + 16;
    static constexpr int kDirB = 6 * kUnitB + kRemB + kLo01B + kLdsImgB;
};
struct GruL1fParams {
    const char* w;         // two GruL1fLayout images (forward, backward)
    uint32_t w_dir_stride; // bytes between them
    const char* y0;        // layer-0 outputs as halves, logically [pos'][hi 200 | lo 200], pos' = ((b / 16) L + t) 16 + b % 16; per (group, step) in the regions of turboae_y0.hpp
    float* hpart;          // [pos'][dir][8]: this direction's share of the Linear head
    int32_t B, L, ngroups; // ngroups = ceil(B / 16)
};
hipError_t launch_gru_l1f(const GruL1fParams& P, hipStream_t st);

// LSTM / vanilla-RNN cells of DEC_LargeRNN in the unit-split f16x2 layout (turboae_rnn_u.hip).  Weight image of one direction of one
// layer (G gates): 6 unit-wave register images (per gate 8 fragments of 1 KB: W_hh {slab 0..2: hi, lo; remainder}, the layer-0 input
// slab) | remainder-wave image (mixed tile: 7 + input slab; head tile: 7) | LDS image (bias rows [ut][gate][16], remainder row, 2^-S,
// 2^-S_head).
struct RnnULayout {
    static constexpr size_t dir_bytes(int G) { return (size_t)(6 * G * 8 + 15) * 1024 + (size_t)((6 * G * 64 + 64 + 16 + 15) / 16 * 16); }
    static constexpr size_t proj_bytes(int G) { return (size_t)2 * 7 * (6 * G + 1) * 2048 + (size_t)2 * (6 * G + 1) * 64 + 16; }
};
struct RnnUParams {
    const char* w;          // two RnnULayout images (forward, backward)
    uint32_t w_dir_stride;
    const float* x;         // layer 0: stack-input panel (B, L, 8)
    const float* gi;        // layer 1: projections [(g16 L + t) 2 + dir][6 G + 1][lane][4], already times the recurrence's 2^S
    char* y0;               // layer 0: outputs as halves, logically [pos'][hi 200 | lo 200], pos' = ((b / 16) L + t) 16 + b % 16; per (group, step) in the regions of turboae_y0.hpp
    float* hpart;           // layer 1: [pos'][dir][8] this direction's share of the Linear head
    int32_t B, L, ngroups;  // ngroups: set by the launcher (ceil(B / 16 NT), NT = N tiles per workgroup, picked by batch)
    int32_t ncu;            // compute units of the handle's device (grid size and the NT choice)
    // layer 1, fused form (r06, launch_rnn_l1f_u): the projection GEMM runs inside the recurrence, two steps at a time, GI never exists
    const char* wproj;      // rnn_proj_u's image: A fragments [dir][slab 7][tile][hi | lo][lane][8 halves] | bias rows [dir][tile][16] | 2^-S
    float gi_mul[2];        // per direction: the layer-1 recurrence's own power-of-two scale (RnnProjParams::gi_mul)
};
struct RnnProjParams {
    const float* yin;       // layer-0 outputs (halves)
    const float* w;         // A fragments [dir][slab 7][tile][hi | lo][lane][8 halves] | bias rows [dir][tile][16] | 2^-S
    float* gi;
    size_t npos;            // positions incl. the padding blocks of the last group of 16
    float gi_mul[2];        // per direction: the layer-1 recurrence's own power-of-two scale
};
hipError_t launch_rnn_rec_u(int gates, bool layer0, const RnnUParams& P, hipStream_t st);
hipError_t launch_rnn_proj_u(int gates, const RnnProjParams& P, hipStream_t st);
hipError_t launch_rnn_l1f_u(int gates, const RnnUParams& P, hipStream_t st);          // layer 1 with the projection inside (y0 = layer-0 outputs, wproj, gi_mul)
int gru_l1f_lds_bytes();
hipError_t launch_gru_prep_enc(const float* u, const int32_t* perm, float* X, int B, int L, int interleaved, hipStream_t st);
int gru_head_grid(size_t npos);
hipError_t launch_gru_prep(const float* rx, const int32_t* perm, float* XA, float* XB, int B, int L, hipStream_t st);
hipError_t launch_gru_rec(bool layer0, const GruRecParams& P, hipStream_t st);
hipError_t launch_gru_proj(const GruProjParams& P, hipStream_t st);
hipError_t launch_gru_rec_h(bool layer0, const GruRecParams& P, hipStream_t st);      // f16x2: w / w_dir_stride in BYTES, layer-0 y as halves
hipError_t launch_gru_rec0u(const GruRecParams& P, hipStream_t st);                    // layer 0 again, one 16-block group per 7-wave workgroup (small batches); bit-identical to launch_gru_rec_h(true, ...)
hipError_t launch_gru_proj_h(const GruProjParams& P, hipStream_t st);
hipError_t launch_gru_head_part(const GruHeadParams& P, hipStream_t st);       // f16x2: bias + act + extrinsic + scatter on the fused head products
hipError_t launch_gru_head(const GruHeadParams& P, hipStream_t st);

hipError_t launch_seg(int U, const SegParams& P, int grid, hipStream_t st);
int seg_lds_bytes(int U, int T, int n_layer);
hipError_t launch_fused(int U, bool decoder, const FusedParams& P, int grid, hipStream_t st);       // P.tap_out != nullptr: tap-exporting decoder
hipError_t launch_fused_h(int U, bool decoder, const FusedParams& P, int grid, hipStream_t st);
int fused_lds_bytes_h(int U, int L, int nb, int taps, int range_layers);     // range_layers: (stack, layer) rows of range bookkeeping = stacks * layers of the side
hipError_t launch_seg_h(int U, const SegParams& P, int grid, hipStream_t st);
int seg_lds_bytes_h(int U, int T, int n_layer, int taps = 5);
int seg_lds_bytes_h_dense(int U, int T, int n_layer);
hipError_t launch_reduce_partials(const double* partials, int n, double count, double* stats, hipStream_t st);
struct NormOpts {          // device-side view of tae_channel_opts
    int32_t norm_mode; float mean, std;
    int32_t ste; float enc_value_limit, enc_quantize_level, enc_truncate_limit;
    int32_t channel, rec_quantize; float rec_quantize_limit, rec_quantize_level;
};
hipError_t launch_normalize(const float* xtx, const double* stats, const float* noise, float* codes, float* rx, size_t n,
                            const NormOpts& o, hipStream_t st);
hipError_t launch_count_errors(const float* xdec, const float* u, int B, int L, unsigned long long* counts, hipStream_t st);
hipError_t launch_gen_inputs(float* u, float* noise, size_t n_bits, size_t bit_offset, unsigned long long seed_bits,
                             unsigned long long seed_noise, float sigma, hipStream_t st);
// device-side view of tae_noise_opts (+ the values derived on the host from test_sigma)
struct NoiseGen {
    int32_t kind;              // TAE_NOISE_*
    float sigma;               // additive kinds: 10^(-test_sigma / 20)
    float p;                   // bec / bsc: erase / flip probability; ge: P(1) in the bad state
    float s_good, s_bad;       // ge_awgn: sigma one dB up / one dB down
    float vv, radar_prob, radar_power, p_gg, p_bb;
};
hipError_t launch_gen_noise(const NoiseGen& g, float* noise, float* fading, size_t n_blocks, size_t first_block, int L,
                            unsigned long long seed, hipStream_t st);
// generic fp32 layer-at-a-time kernels (turboae_generic.hip): configurations outside the MFMA kernels' envelope
struct GenericEngine;
bool generic_needed(const tae_config* c);
const char* generic_check(const tae_config* c);          // nullptr, or why the configuration is out of range
size_t generic_num_weights(const tae_config* c);
int generic_create(const tae_config* c, const float* weights, size_t n_weights, GenericEngine** out);
void generic_destroy(GenericEngine* g);
int generic_reserve(GenericEngine* g, int32_t B);
int generic_encode(GenericEngine* g, const float* u, float* xtx, double* stats, const int32_t* perm, int32_t B, hipStream_t st);
int generic_decode(GenericEngine* g, const float* rx, float* xdec, const int32_t* perm, const int32_t* inv, int32_t B, hipStream_t st, float* tap_out = nullptr);
// Debug knobs: every environment variable that changes the arithmetic, the kernel family or a launch geometry goes through here.
// It is inert (nullptr) unless TAE_DEBUG_KNOBS=1 is set in the same environment, and every override that took effect is recorded
// for tae_overrides().
const char* debug_knob(const char* name);
int fail_msg(int code, const char* msg);          // turboae_api_entry.hip: sets the calling thread's tae_last_error string
int fused_lds_bytes(int U, int L, int nb);
int fused_max_positions();

}  // namespace tae
