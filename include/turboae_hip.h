/*
 * libturboae_hip.so - C ABI of the MI355X-native TurboAE rate-1/3 CNN inference path.
 *
 * The reference (yihanjiang/turboae) is pure Python/PyTorch and has NO FFI/plugin layer; the seam
 * a drop-in must honour is the Python call
 *     x_dec, codes = Channel_AE.forward(input, fwd_noise)            channel_ae.py:20-73
 * plus model.enc(X) (trainer.py:152,243), enc/dec.set_interleaver (channel_ae.py:35-36) and the
 * metrics errors_ber / errors_bler (utils.py:6-18,49-66).  Each entry point below names the
 * reference interface it replaces.  The Python binding a maintainer would add is
 * turboae_amd/_lib.py (ctypes) + turboae_amd/channel_ae.py; see INTEGRATION.md.
 *
 * Conventions
 *  - every tensor pointer is a DEVICE pointer (HBM resident) to contiguous little-endian fp32 in
 *    the reference's own layouts: u (B,L,1), noise / codes / received (B,L,3), x_dec (B,L,1);
 *  - `stream` is a hipStream_t passed as void* (NULL = default stream); all calls are asynchronous
 *    on it, allocate nothing and never synchronise, so a caller may capture them in a hipGraph;
 *  - a handle is single-owner (not thread-safe), bound to the device current at tae_create: one handle per GPU, and
 *    every call that touches the device returns TAE_ESTATE unless that device is the calling thread's current one
 *    (a process driving several GPUs does hipSetDevice first; nothing is ever launched on foreign pointers);
 *  - every function returns 0 on success or a negative TAE_E* code; tae_last_error() gives the
 *    message for the calling thread.  Shape mismatches are rejected, never silently re-viewed
 *    (the reference hard-codes args.batch_size views, decoders.py:221).
 */
#ifndef TURBOAE_HIP_H_
#define TURBOAE_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TAE_ABI_VERSION 12

#if defined(__GNUC__)
#define TAE_API __attribute__((visibility("default")))
#else
#define TAE_API
#endif

enum {
    TAE_OK = 0,
    TAE_EINVAL = -1,    /* bad argument / unsupported configuration */
    TAE_EHIP = -2,      /* HIP runtime error (message in tae_last_error) */
    TAE_ENOMEM = -3,
    TAE_ESTATE = -4     /* call sequence error (e.g. batch larger than reserved) */
};

/* Flags the hot path reads from the reference's argparse namespace (get_args.py:73-122). */
typedef struct tae_config {
    int32_t struct_size;      /* = sizeof(tae_config), for ABI evolution */
    int32_t block_len;        /* -block_len        get_args.py:122 */
    int32_t enc_num_layer;    /* -enc_num_layer    get_args.py:93  */
    int32_t enc_num_unit;     /* -enc_num_unit     get_args.py:98  (1..124 on the fused MFMA kernels - 1..100 in TAE_PREC_F32 and for RNN stacks -, independent of dec_num_unit: instantiated for 32 / 64 / 100 / 124, narrower stacks run embedded; above that to 1024: generic fp32 kernels) */
    int32_t enc_kernel_size;  /* -enc_kernel_size  get_args.py:89  (1, 3, 5, 7, 9 on the fused kernels: 1 and 3 are embedded into 5 taps; 7 / 9 with TAE_PREC_F32, and odd sizes 11..63: generic fp32 kernels) */
    int32_t dec_num_layer;    /* -dec_num_layer    get_args.py:94  */
    int32_t dec_num_unit;     /* -dec_num_unit     get_args.py:97  (as enc_num_unit) */
    int32_t dec_kernel_size;  /* -dec_kernel_size  get_args.py:90  (as enc_kernel_size) */
    int32_t num_iteration;    /* -num_iteration    get_args.py:82  */
    int32_t num_iter_ft;      /* -num_iter_ft      get_args.py:84  */
    int32_t extrinsic;        /* -extrinsic        get_args.py:83  */
    int32_t enc_act;          /* -enc_act (get_args.py:100, encoders.py:86-100): TAE_ACT_* below; 0 = elu, the reference default */
    int32_t max_batch;        /* blocks per call the workspace is sized for (grown by tae_reserve) */
    int32_t dec_type;         /* -decoder: 0 = TurboAE_rate3_cnn (DEC_LargeCNN, decoders.py:157),
                                 1 = TurboAE_rate3_rnn (DEC_LargeRNN with dec_rnn=gru, decoders.py:16; dec_num_unit <= 100, runs in the 100-unit kernels)  main.py:75-88 */
    int32_t enc_type;         /* -encoder: 0 = TurboAE_rate3_cnn (ENC_interCNN, encoders.py:304), 1 = TurboAE_rate3_rnn (ENC_interRNN with
                                 enc_rnn=gru, encoders.py:231-298; needs dec_type = 1, enc_num_layer = 2; enc_num_unit <= 100)  main.py:32-36 */
    int32_t dense;            /* 1: -encoder TurboAE_rate3_cnn_dense: DenseSameShapeConv1d stacks in encoder AND decoder (cnn_utils.py:49-82; the
                                 reference keys both on the encoder name, encoders.py:312-330, decoders.py:173-176); needs enc_type =
                                 dec_type = 0 and precision = TAE_PREC_AUTO */
    int32_t precision;        /* arithmetic of the conv contractions (no reference counterpart; the reference is fp32 on CPU/CUDA):
                                 TAE_PREC_AUTO = 0: fp32 operands carried as two fp16 halves, three v_mfma_f32_16x16x32_f16 products,
                                 fp32 accumulation - fp32-grade results - where the whole-block kernels apply,
                                 TAE_PREC_F32 otherwise; TAE_PREC_F32 = 1: v_mfma_f32_16x16x4_f32 on the fp32 operands everywhere;
                                 TAE_PREC_F16X1 = 2 (r06): NOT fp32-grade and never chosen by the library - the TAE_PREC_AUTO handle with its
                                 whole-block CNN decoder (dec_type = 0, 65 <= dec_num_unit <= 100, block_len <= 320) on the hi halves only
                                 (one fp16 product per 32 k instead of three): soft outputs differ from the reference's at the 1e-3
                                 level; exists to price the fp32-tolerance requirement (bench.py `f16x1_*`), carries no parity claim */
    int32_t dec_act;          /* -dec_act (get_args.py:101, decoders.py:59-73): TAE_ACT_* on the GRU decoder's Linear outputs (DEC_LargeCNN
                                 has no dec_act; ignored for dec_type = 0).  The reference default is TAE_ACT_LINEAR (= 1, NOT 0) */
    int32_t enc_rnn;          /* -enc_rnn (get_args.py:79, encoders.py:242-247): TAE_RNN_GRU / LSTM / RNN cell of ENC_interRNN (enc_type = 1); since r06 every cell
                                 runs on tuned kernels with enc_num_layer = 2 in TAE_PREC_AUTO, whatever cell the decoder uses */
    int32_t dec_rnn;          /* -dec_rnn (get_args.py:80, decoders.py:27-32): the cell of DEC_LargeRNN (dec_type = 1) */
    int32_t range_calibration;/* fp16-split conv kernels (no reference counterpart: the reference's fp32 F.conv1d, cnn_utils.py:36-46, has 24
                                 significant bits at any magnitude; an fp16 hi/lo pair has them only for values in about [2^-3, 2^16)):
                                 0 (default) = tae_create, the FIRST tae_set_interleaver after it and a tae_set_channel_opts that changes what
                                 the decoder receives run the handle's own forward on a synthetic batch, measure every layer's largest
                                 activation and store each panel times a per-layer power of two that puts that maximum at [2^10, 2^11)
                                 (see tae_calibrate_range; later permutations keep the measurement, and a calibration on the caller's
                                 own data is kept until the caller replaces it); 1 = no calibration, every exponent 0 (fp32-grade only
                                 while activations are O(1); testing / A-B) */
    int32_t range_fallback;   /* 1: every compute entry point (tae_forward / encode / encode_prenorm / decode / eval_snr) waits for its
                                 launches, reads the range word and, if an activation left the window (above the fp16 range, or a
                                 workgroup's data 2^7 below the calibration maximum), runs the call again on an fp32 twin of the handle
                                 (fp32 MFMA kernels; generic fp32 kernels for dense stacks / kernel sizes 7, 9) - and every later call goes
                                 there directly.  Such a handle synchronises per call and refuses hipGraph capture (TAE_ESTATE).
                                 0 (default): asynchronous; the caller checks tae_range_status */
} tae_config;

/* Configurations the fused MFMA kernels do not instantiate run on generic fp32 kernels (one launch per layer, fp32 operands on the
 * fp32 matrix cores, activations through HBM: 3..8 times slower than the kernels above, same results): channel widths 125..1024
 * (recurrent cells: 101..1024), odd kernel sizes 11..63,
 * num_iter_ft 7..64, LSTM / vanilla-RNN cells in TAE_PREC_F32 (in TAE_PREC_AUTO they have unit-split fp16-split kernels of their own, decoder
 * and 2-layer encoder alike), ENC_interRNN with enc_num_layer != 2 or in front of a CNN decoder (which the
 * reference then builds from DenseSameShapeConv1d, decoders.py:173-176), and TAE_PREC_F32 for dense stacks / kernel sizes 7, 9. */
#define TAE_RNN_GRU 0
#define TAE_RNN_LSTM 1
#define TAE_RNN_RNN 2

#define TAE_ACT_ELU 0
#define TAE_ACT_LINEAR 1
#define TAE_ACT_TANH 2
#define TAE_ACT_RELU 3
#define TAE_ACT_SELU 4
#define TAE_ACT_SIGMOID 5

#define TAE_PREC_AUTO 0
#define TAE_PREC_F32 1
#define TAE_PREC_F16X1 2

/* Encoder-output / channel variants of the same kernels (reference flags in parentheses).  Defaults
 * (all zero except the limits) are the reference defaults: batch power normalisation, AWGN add. */
typedef struct tae_channel_opts {
    int32_t struct_size;          /* = sizeof(tae_channel_opts) */
    int32_t norm_mode;            /* 0: batch mean/std (encoders.py:107-116); 1: none (--no_code_norm, :104-105);
                                     2: fixed mean/std below (--precompute_norm_stats, :110-114 - the caller keeps the running averages) */
    float   mean, std;            /* norm_mode 2 */
    int32_t ste;                  /* 1: -train_channel_mode block_norm_ste: STEQuantize.forward on the normalised codes (encoders.py:20-36,118-120) */
    float   enc_value_limit;      /* -enc_value_limit (get_args.py:168) */
    float   enc_quantize_level;   /* -enc_quantize_level (get_args.py:167); 2 = sign */
    float   enc_truncate_limit;   /* -enc_truncate_limit > 0: clamp (encoders.py:122-123) */
    int32_t channel;              /* 0: received = codes + noise (awgn, t-dist, radar, ge_awgn; channel_ae.py:41-42);
                                     1: codes * noise (bec, :44-45); 2: codes * (2*noise - 1) (bsc, ge; :47-49);
                                     3: fading_h * codes + noise (fading, :51-56) - the reference draws fading_h inside forward; here
                                        the caller supplies it: `noise` then points to 2*B*L*3 floats, fading_h followed by the noise */
    int32_t rec_quantize;         /* 1: --rec_quantize: STEQuantize.forward(received, limit, level) (channel_ae.py:67-69, ste.py:9-23) */
    float   rec_quantize_limit;   /* the reference passes rec_quantize_level as the limit too (channel_ae.py:69) */
    float   rec_quantize_level;
} tae_channel_opts;

typedef struct tae_handle tae_handle;

/* Number of fp32 values tae_create expects in `weights` for this configuration (canonical order:
 * turboae_amd/weights.py canonical_entries; PyTorch layouts of the reference state_dict,
 * SURVEY.md Appendix B).  Returns 0 for an unsupported configuration. */
TAE_API size_t tae_num_weights(const tae_config* cfg);

/* Replaces model construction + load_state_dict (main.py:146-172).  `weights` is a HOST pointer to
 * the canonical fp32 blob; it is re-tiled into MFMA fragment order and uploaded.  The interleaver
 * is initialised to the identity; call tae_set_interleaver before use. */
TAE_API int tae_create(const tae_config* cfg, const float* weights, size_t n_weights, tae_handle** out);
TAE_API int tae_destroy(tae_handle* h);

/* Grow the internal workspace to `max_batch` blocks (allocates; not stream-ordered; call outside
 * timed / captured regions).  Footprint per handle: CNN paths ~100 B per position; recurrent decoders work in chunks of at most 16 384
 * blocks x 100 positions and hold, per position of min(max_batch, chunk): 1.7 KB (GRU on the fused layer-1 kernel: 2.7 GB per full chunk)
 * or 1.7 KB + the layer-1 projections GI (LSTM: 3.2 KB -> 7.9 GB per full chunk; vanilla RNN 0.9 KB; fp32 GRU path 2.4 KB).  Every
 * variable-block-length engine and every range_fallback twin is a handle of its own and owns its own copy. */
TAE_API int tae_reserve(tae_handle* h, int32_t max_batch);

/* Replaces enc.set_interleaver + dec.set_interleaver (channel_ae.py:35-36, encoders.py:340-341,
 * decoders.py:202-204).  `p` is a HOST array with p[i] in [0,L), a permutation; L must equal
 * block_len.  out[:, i, :] = in[:, p[i], :] (interleavers.py:15-21). */
TAE_API int tae_set_interleaver(tae_handle* h, const int32_t* p, int32_t L);

/* Selects the encoder-output / channel variant used by tae_forward, tae_encode and tae_normalize from now on
 * (NULL restores the defaults).  Replaces reading args.* inside power_constraint / Channel_AE.forward. */
TAE_API int tae_set_channel_opts(tae_handle* h, const tae_channel_opts* opts);

/* Replaces Channel_AE.forward (channel_ae.py:20-73, AWGN branch :41-42):
 *   codes = enc(u) ; received = codes + noise ; x_dec = dec(received).
 * power_constraint statistics are taken over exactly the B blocks of this call
 * (encoders.py:107-108), as in the reference.  codes may be NULL. */
TAE_API int tae_forward(tae_handle* h, const float* u, const float* noise, float* x_dec, float* codes, int32_t B, void* stream);

/* Replaces model.enc(X) = ENC_interCNN.forward incl. power_constraint (encoders.py:351-377). */
TAE_API int tae_encode(tae_handle* h, const float* u, float* codes, int32_t B, void* stream);

/* Split form for multi-GPU sharding: x_tx (B,L,3) is the encoder output BEFORE
 * power_constraint; stats3 (device, 3 doubles) receives (sum, sum of squares, count) of this
 * shard.  The caller all-reduces stats3 over ranks (SUM), then calls tae_normalize. */
TAE_API int tae_encode_prenorm(tae_handle* h, const float* u, float* x_tx, double* stats3, int32_t B, void* stream);

/* codes = (x_tx - mean) / std with mean / unbiased std derived from stats3 (encoders.py:107-116);
 * received = codes + noise (channel_ae.py:42).  codes, or noise+received, may be NULL. */
TAE_API int tae_normalize(tae_handle* h, const float* x_tx, const double* stats3, const float* noise, float* codes,
                  float* received, int32_t B, void* stream);

/* Replaces model.dec(received) = DEC_LargeCNN.forward (decoders.py:206-269) or, with dec_type = 1,
 * DEC_LargeRNN.forward (decoders.py:84-149). */
TAE_API int tae_decode(tae_handle* h, const float* received, float* x_dec, int32_t B, void* stream);

/* tae_decode plus a debug export of what every half-iteration hands to the next one (no reference counterpart: the reference keeps
 * `x_plr` / `prior` as locals of DEC_LargeCNN.forward, decoders.py:229-249): taps (device, (2*num_iteration - 1) * B * L * num_iter_ft
 * floats) receives, for stack s = 2*it (dec1) the extrinsic output x_plr = dec1_outputs[it](...) - prior in natural order, and for
 * s = 2*it + 1 (dec2, it < num_iteration - 1) x_plr = dec2_outputs[it](...) - x_plr_int in INTERLEAVED order (prior = its
 * deinterleave), as [s][b][position][f].  Same results in x_dec as tae_decode; runs a separate instantiation of the decoder kernel
 * (the production kernel carries no tap code).  Used by tests/ to localise a regression to one stack.  DEC_LargeRNN handles (GRU / LSTM /
 * vanilla RNN on the tuned kernels, either arithmetic) export the same values from their head epilogue: dec{1,2}_outputs[it] of
 * decoders.py:84-149 after the extrinsic subtraction.  Not on the generic fp32 kernels, not for DenseSameShapeConv1d stacks. */
TAE_API int tae_decode_taps(tae_handle* h, const float* received, float* x_dec, float* taps, int32_t B, void* stream);

/* Replaces errors_ber / errors_bler (utils.py:6-18,49-66) as integer counts ACCUMULATED into
 * counts2 (device, 2 x uint64): [0] += bit errors, [1] += blocks with >= 1 bit error. */
TAE_API int tae_count_errors(tae_handle* h, const float* x_dec, const float* u, int32_t B, uint64_t* counts2, void* stream);

/* Replaces the test-input draws of trainer.test (trainer.py:167-169; channels.py:27-35;
 * utils.py:69-70): u ~ Bernoulli(0.5) as {0,1} floats, noise = 10^(-snr_db/20) * N(0,1), from the
 * counter-based Philox4x32-10 streams of turboae_amd/philox.py.  `first_block` is the GLOBAL index
 * of the first block, so any shard reproduces the single-device stream.  Either output may be NULL. */
TAE_API int tae_generate_inputs(tae_handle* h, float* u, float* noise, int32_t B, int64_t first_block, uint64_t seed_bits,
                        uint64_t seed_noise, float snr_db, void* stream);

/* Test-time channel noise of every channel of the reference on the device: replaces generate_noise(noise_shape, args, test_sigma=...)
 * (channels.py:7-109, the test_sigma != 'default' branches trainer.test uses, trainer.py:167-169) and, for -channel fading, the
 * Rayleigh coefficients Channel_AE.forward draws itself (channel_ae.py:51-56).  Counter-based: every value is a function of
 * (seed, global element index ((first_block + b) * L + t) * 3 + c) on named Philox streams (turboae_amd/philox.py; the numpy mirror
 * turboae_amd/channels.py::generate_noise reproduces the draws), so any shard of any batch can be drawn on any rank. */
#define TAE_NOISE_AWGN 0      /* sigma * N(0,1)                                                     channels.py:37-38 (== tae_generate_inputs' noise) */
#define TAE_NOISE_TDIST 1     /* sigma * sqrt((vv-2)/vv) * standard_t(vv)                           channels.py:40-41 */
#define TAE_NOISE_RADAR 2     /* sigma * N(0,1) + radar_power * N(0,1) * Bernoulli(radar_prob)      channels.py:43-49 */
#define TAE_NOISE_GE_AWGN 3   /* Gilbert-Elliott chain per (block, code symbol) along time: good state sigma(snr + 1 dB), bad state
                                 sigma(snr - 1 dB), times N(0,1)                                    channels.py:58-82 */
#define TAE_NOISE_BEC 4       /* keep mask: 0 with probability test_sigma, else 1                   channels.py:51-53 */
#define TAE_NOISE_BSC 5       /* the same mask (applied as a sign flip by the channel)              channels.py:55-57 */
#define TAE_NOISE_GE 6        /* the chain; good state: 1, bad state: 1 with probability test_sigma channels.py:84-107 */
#define TAE_NOISE_FADING 7    /* noise = sigma * N(0,1), fading_h = sqrt(N^2 + N^2) / sqrt(3.14/2)  channels.py:37-38, channel_ae.py:53 */
typedef struct tae_noise_opts {
    int32_t struct_size;      /* = sizeof(tae_noise_opts) */
    int32_t kind;             /* TAE_NOISE_* (-channel, get_args.py:43) */
    float   vv;               /* -vv (get_args.py:53): t-dist degrees of freedom, > 2 */
    float   radar_prob;       /* -radar_prob (get_args.py:55) */
    float   radar_power;      /* -radar_power (get_args.py:56) */
    float   p_gg, p_bb;       /* Gilbert-Elliott: P(next good | good), P(next GOOD | bad) - the reference hard-codes 0.8 / 0.8 and, as
                                 written, returns to the good state with p_bb (channels.py:73,79,100,105) */
} tae_noise_opts;
/* test_sigma: SNR in dB for the additive kinds, erase / flip probability for BEC / BSC / GE (channels.py:28-31).  noise (device,
 * B*L*3 floats); fading_h (device, B*L*3 floats) is written for TAE_NOISE_FADING only and may be NULL otherwise - pass
 * fading_h = buf and noise = buf + B*L*3 to obtain the layout tae_channel_opts.channel = 3 consumes. */
TAE_API int tae_generate_noise(tae_handle* h, const tae_noise_opts* opts, float test_sigma, float* noise, float* fading_h, int32_t B,
                               int64_t first_block, uint64_t seed, void* stream);
/* The generator tae_eval_snr draws its noise from (NULL: AWGN, the default); with a non-AWGN generator tae_eval_snr's `snr_db`
 * argument is the test_sigma above, exactly as trainer.test hands its loop variable to generate_noise (trainer.py:160-169). */
TAE_API int tae_set_noise_opts(tae_handle* h, const tae_noise_opts* opts);

/* One SNR point of the reference's eval sweep (trainer.test, trainer.py:160-217; the entry point SURVEY.md section 8b sketches as
 * tae_eval_snr) entirely on the device: n_batches batches of `batch` blocks.  Batch i (global blocks first_block + i*batch ...) gets
 * the Philox inputs of tae_generate_inputs, runs encoder -> power constraint with ITS OWN statistics -> additive noise; the
 * received blocks of a group of batches (about 24 576 blocks) are decoded in one call - the decoder never mixes blocks - and
 * the errors are counted per batch: counts (device, 2*n_batches uint64, zeroed by the call) = (bit errors, block errors) of
 * batch i at [2i], [2i+1], so BER = mean_i counts[2i] / (batch * block_len) exactly as trainer.py:176-177,215-216 average it.
 * The noise comes from the generator installed with tae_set_noise_opts (default AWGN) and is applied by the channel of
 * tae_set_channel_opts (additive, bec, bsc / ge sign flips, fading with device-drawn coefficients).  The first call for a geometry grows the workspace (allocates,
 * synchronises); after that the call only enqueues work on `stream`. */
TAE_API int tae_eval_snr(tae_handle* h, float snr_db, int32_t batch, int32_t n_batches, int64_t first_block, uint64_t seed_bits,
                         uint64_t seed_noise, uint64_t* counts, void* stream);

/* Blocks per workgroup and dynamic LDS bytes of the fused kernels (for DESIGN / bench reporting). */
TAE_API int tae_kernel_info(tae_handle* h, int32_t* blocks_per_workgroup, int32_t* lds_bytes);

/* Which instantiation of the fp16-split whole-block kernels this handle's production launches use (no reference counterpart): 1 when
 * one of that side's last conv layers stayed below 1/4 in calibration, so its Linear heads evaluate both expm1 branches per value
 * (a twin of the plain kernel with the same registers and no scratch - not the calibration instantiation); 0 otherwise, and always 0
 * for fp32 / generic / long-block / GRU handles.  Known limit: the long-block, segmented and dense f16x2 kernels have no such
 * twin and do not track their last layers' maxima - their heads always evaluate exp2 - 1 (3e-8 ABSOLUTE on an ELU output, i.e. a large
 * relative error only for last-layer activations far below 1/4; `var_small_last_L1000` holds such a network to the golden tolerance),
 * so a network with tiny last-layer activations is most accurate on the whole-block path (block_len <= 320) or with TAE_PREC_F32.
 * Either pointer may be NULL. */
TAE_API int tae_kernel_variants(tae_handle* h, int32_t* enc_both_branch_heads, int32_t* dec_both_branch_heads);

/* Debug overrides in effect (no reference counterpart).  Environment variables that change the arithmetic, the kernel family or a
 * launch geometry (TAE_PRECISION, TAE_RANGE_CAL, TAE_FORCE_GENERIC, TAE_FORCE_SEGMENTED, TAE_SEG_T, TAE_FIXED_NB, TAE_NO_SUPER,
 * TAE_GRU_*, TAE_GEN_*) are IGNORED unless TAE_DEBUG_KNOBS=1 is set beside them; every one that took effect in this process is
 * listed here as "NAME=value;NAME=value" (empty string: none - the handle runs exactly what its tae_config asked for).
 * Writes at most n bytes including the terminating NUL; returns the length the full list needs (without the NUL).  h may be NULL. */
TAE_API int tae_overrides(tae_handle* h, char* buf, int32_t n);

/* Measurement support (no reference counterpart; needs no handle): the rate a pure stream of v_mfma_f32_16x16x32_f16 sustains on
 * the current device for at least `min_ms` milliseconds - one 8-wave workgroup per CU (the decoder's residency), 16 independent
 * accumulator tiles per wave, operands refreshed from LDS every iteration; N(0,1) fp16 operands, or all-zero ones with zero_data = 1
 * (zeros clock higher: the limit on real data is power).  *tflops = dense f16 MFMA TFLOP/s executed.  bench.py reports the decoder
 * against this next to the spec-peak fraction.  Allocates, launches on the null stream and SYNCHRONISES. */
TAE_API int tae_probe_mfma_f16(int32_t zero_data, int32_t min_ms, double* tflops, double* ms_measured);

/* Arithmetic actually in use (*precision = 0: fp32 MFMA, 1: fp16-split MFMA) and the sticky range word of the fp16-split kernels,
 * *overflow = bit mask of what happened since the last call:
 *   TAE_RANGE_HIGH  a scaled activation or stack input exceeded the fp16 range (65504): inf / NaN halves went through those launches,
 *                   their results are NOT trustworthy;
 *   TAE_RANGE_LOW   some workgroup's largest value of a panel sat below 2^3 after scaling, i.e. >= 2^7 under what the calibration
 *                   measured: results are finite and close, but no longer fp32-grade (absolute floor 2^-25 / scale per value);
 *   TAE_RANGE_FELL_BACK  (range_fallback handles) a flagged call was re-run on the fp32 twin, which has served every call since:
 *                   the results handed out are the fp32 kernels'.
 * Without range_fallback the way out of HIGH / LOW is tae_calibrate_range on representative data, or a handle with TAE_PREC_F32.
 * Synchronises the WHOLE device (hipDeviceSynchronize, then reads one word back) and clears the HIGH / LOW bits: do not call it
 * while any stream of the process is capturing a hipGraph (the synchronisation invalidates the capture) - check after the replay. */
#define TAE_RANGE_HIGH 1
#define TAE_RANGE_LOW 2
#define TAE_RANGE_FELL_BACK 4
TAE_API int tae_range_status(tae_handle* h, int32_t* precision, int32_t* overflow);

/* Measures the activation ranges of the fp16-split conv kernels on the caller's data and re-derives the per-layer exponents
 * (tae_config.range_calibration describes the scheme): u (B,L,1) and noise as tae_forward takes them (device pointers), or both
 * NULL for the built-in synthetic batch (Bernoulli bits; the configured channel's own kind of noise at 0 dB).  Runs the forward a
 * few times on the NULL stream, rewrites the packed bias / scale tails (the weight fragments are untouched), SYNCHRONISES the
 * device, clears the range word.  Every exponent is a power of two: a different calibration batch moves where the floor and the
 * ceiling of the fp16 pairs sit, never the rounding of a value inside the window.  No-op for fp32 / GRU-only / generic handles.
 * A calibration on the caller's data stays in force through later tae_set_interleaver / tae_set_channel_opts calls (they re-measure
 * only a synthetic calibration); call again - with data, or with NULLs to return to the synthetic batch - after such a change.
 * Inputs: a calibrated handle's encoder runs without range bookkeeping on the assumption that u holds bits (0 / 1, as
 * Channel_AE.forward's input does); soft or scaled inputs belong to a handle created with range_calibration = 1 or TAE_PREC_F32. */
TAE_API int tae_calibrate_range(tae_handle* h, const float* u, const float* noise, int32_t B);

/* Diagnostics: the exponents in use.  *n_encoder / *n_decoder = number of int32 values per side (0: not calibrated): first one
 * exponent per stack (its input planes), then one per (stack, layer) panel; `exponents` (capacity >= n_encoder + n_decoder, or NULL)
 * receives encoder then decoder values; *passes = forward passes the last calibration ran. */
TAE_API int tae_range_info(tae_handle* h, int32_t* n_encoder, int32_t* n_decoder, int32_t* exponents, int32_t capacity, int32_t* passes);

/* Test hook (no device needed): the host-side fp32 -> fp16 hi/lo split used when packing weights for the fp16-split
 * kernels: hi = f16(x * scale) (round to nearest even, denormals kept), lo = f16(x * scale - hi).  Writes n values each. */
TAE_API int tae_debug_split_f16(const float* x, size_t n, float scale, uint16_t* hi, uint16_t* lo);

TAE_API const char* tae_last_error(void);
TAE_API int tae_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* TURBOAE_HIP_H_ */
