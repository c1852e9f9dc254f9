/* The reference's eval sweep (trainer.test, trainer.py:135-248) driven from plain C through the C ABI of
 * libturboae_hip.so - no Python, no torch: what a cgo / JNI / FFI binding of another host language would do.
 *
 *   turboae_sweep <weights.f32> <perm.i32> [blocks_per_snr=10000] [batch=500] [snr_points=12] [snr_lo=-1.5] [snr_hi=4.0] [seed=1] [mode=0] [channel=awgn] [range_fallback=0]
 *
 * range_fallback 1: tae_config.range_fallback - a call whose activations leave the window of the fp16-split kernels is run again on the
 * library's fp32 twin of the handle (and so is every later call); the last line then reports it.  This is how a host without the
 * Python mirror gets fp32-grade numbers from ANY checkpoint.
 *
 * channel: awgn | t-dist | radar | ge_awgn | bec | bsc | ge | fading (-channel, get_args.py:43): the noise of every channel is drawn on
 * the device (tae_generate_noise / tae_set_noise_opts) and applied by the matching branch of Channel_AE.forward (tae_set_channel_opts);
 * for bec / bsc / ge the sweep variable is the erase / flip probability, as in the reference (channels.py:28-31).
 *
 * mode 0: one tae_eval_snr call per SNR point (the library decodes groups of batches in one launch);
 * mode 1: the same protocol spelled out call by call (tae_generate_inputs -> tae_forward -> tae_count_errors per batch).
 *
 * <weights.f32>: the canonical flat float32 weight blob of the configuration below (turboae_amd.weights.pack_blob;
 * tae_num_weights(cfg) floats).  <perm.i32>: the interleaver, block_len int32 values - the reference draws it with numpy's
 * RandomState(0).permutation (interleavers.py / channel_ae.py:33), so a non-Python host takes it from a file like the
 * weights (turboae_amd.interleaver.rand_interleaver(L, 0).astype('<i4').tofile(...)).  Inputs are the library's counter-based Philox streams keyed like
 * turboae_amd/evaluate.py, so the printed error counts equal evaluate.test's for the same seed (tests/test_gpu_parity.py).
 * Build: see examples/c_host/Makefile (gcc; links libturboae_hip.so and the HIP runtime for hipMalloc / hipMemcpy). */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "turboae_hip.h"

#define HIP_OK(call)                                                                         \
    do {                                                                                     \
        hipError_t e_ = (call);                                                              \
        if (e_ != hipSuccess) {                                                              \
            fprintf(stderr, "%s:%d: %s\n", __FILE__, __LINE__, hipGetErrorString(e_));       \
            return 2;                                                                        \
        }                                                                                    \
    } while (0)
#define TAE_CHECK(call)                                                                      \
    do {                                                                                     \
        int r_ = (call);                                                                     \
        if (r_ != TAE_OK) {                                                                  \
            fprintf(stderr, "%s:%d: error %d: %s\n", __FILE__, __LINE__, r_, tae_last_error()); \
            return 3;                                                                        \
        }                                                                                    \
    } while (0)

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

int main(int argc, char** argv) {
    if (argc < 3) {
        fprintf(stderr, "usage: %s <weights.f32> <perm.i32> [blocks_per_snr] [batch] [snr_points] [snr_lo] [snr_hi] [seed]\n", argv[0]);
        return 1;
    }
    const long num_block = argc > 3 ? atol(argv[3]) : 10000;
    const int batch = argc > 4 ? atoi(argv[4]) : 500;
    const int snr_points = argc > 5 ? atoi(argv[5]) : 12;
    const double snr_lo = argc > 6 ? atof(argv[6]) : -1.5;
    const double snr_hi = argc > 7 ? atof(argv[7]) : 4.0;
    const uint64_t seed = argc > 8 ? (uint64_t)strtoull(argv[8], NULL, 10) : 1u;
    const int mode = argc > 9 ? atoi(argv[9]) : 0;
    const char* channel = argc > 10 ? argv[10] : "awgn";
    const int range_fallback = argc > 11 ? atoi(argv[11]) : 0;
    static const char* const kinds[] = {"awgn", "t-dist", "radar", "ge_awgn", "bec", "bsc", "ge", "fading"};      /* TAE_NOISE_* order */
    int kind = -1;
    for (int i = 0; i < 8; ++i)
        if (strcmp(channel, kinds[i]) == 0) kind = i;
    if (kind < 0) {
        fprintf(stderr, "unknown channel %s\n", channel);
        return 1;
    }
    if (tae_abi_version() != TAE_ABI_VERSION) {
        fprintf(stderr, "header / library ABI mismatch (%d vs %d)\n", TAE_ABI_VERSION, tae_abi_version());
        return 1;
    }

    /* the README's pretrained configuration: enc2 / dec5, 100 filters, block_len 100, 6 iterations */
    tae_config cfg;
    memset(&cfg, 0, sizeof(cfg));
    cfg.struct_size = (int32_t)sizeof(cfg);
    cfg.block_len = 100;
    cfg.enc_num_layer = 2;
    cfg.enc_num_unit = 100;
    cfg.enc_kernel_size = 5;
    cfg.dec_num_layer = 5;
    cfg.dec_num_unit = 100;
    cfg.dec_kernel_size = 5;
    cfg.num_iteration = 6;
    cfg.num_iter_ft = 5;
    cfg.extrinsic = 1;
    cfg.max_batch = batch;
    cfg.precision = TAE_PREC_AUTO;
    cfg.dec_act = TAE_ACT_LINEAR;       /* the reference default (only the GRU decoder reads it) */
    cfg.range_calibration = 0;          /* 0 = on: per-layer exponents of the fp16-split panels, measured at create / set_interleaver */
    cfg.range_fallback = range_fallback ? 1 : 0;

    const size_t nw = tae_num_weights(&cfg);
    float* w = (float*)malloc(nw * sizeof(float));
    FILE* f = fopen(argv[1], "rb");
    if (!w || !f || fread(w, sizeof(float), nw, f) != nw) {
        fprintf(stderr, "cannot read %zu floats from %s\n", nw, argv[1]);
        return 1;
    }
    fclose(f);

    tae_handle* h = NULL;
    TAE_CHECK(tae_create(&cfg, w, nw, &h));
    free(w);

    const int L = cfg.block_len;
    int32_t* perm = (int32_t*)malloc((size_t)L * sizeof(int32_t));
    f = fopen(argv[2], "rb");
    if (!perm || !f || fread(perm, sizeof(int32_t), (size_t)L, f) != (size_t)L) {
        fprintf(stderr, "cannot read %d int32 from %s\n", L, argv[2]);
        return 1;
    }
    fclose(f);
    TAE_CHECK(tae_set_interleaver(h, perm, L));     /* enc.set_interleaver / dec.set_interleaver, channel_ae.py:32-36 */
    free(perm);
    /* the channel: which generator draws the noise (channels.py:37-109) and how Channel_AE.forward applies it (channel_ae.py:41-56) */
    tae_noise_opts nz;
    memset(&nz, 0, sizeof(nz));
    nz.struct_size = (int32_t)sizeof(nz);
    nz.kind = kind;
    nz.vv = 5.0f; nz.radar_prob = 0.05f; nz.radar_power = 5.0f;     /* get_args.py:53-56 defaults */
    nz.p_gg = 0.8f; nz.p_bb = 0.8f;                                 /* channels.py:60-61,86-87 */
    TAE_CHECK(tae_set_noise_opts(h, &nz));
    tae_channel_opts co;
    memset(&co, 0, sizeof(co));
    co.struct_size = (int32_t)sizeof(co);
    co.enc_value_limit = 1.0f; co.enc_quantize_level = 2.0f; co.rec_quantize_limit = 2.0f; co.rec_quantize_level = 2.0f;
    co.channel = kind == TAE_NOISE_BEC ? 1 : (kind == TAE_NOISE_BSC || kind == TAE_NOISE_GE) ? 2 : kind == TAE_NOISE_FADING ? 3 : 0;
    TAE_CHECK(tae_set_channel_opts(h, &co));
    const size_t noise_mult = kind == TAE_NOISE_FADING ? 2 : 1;    /* fading: coefficients followed by the noise */
    const int nbatch = (int)(num_block / batch);
    float *u, *noise, *x_dec, *codes;
    uint64_t* counts;                      /* per batch: (bit errors, block errors), accumulated on the device */
    HIP_OK(hipMalloc((void**)&u, (size_t)batch * L * sizeof(float)));
    HIP_OK(hipMalloc((void**)&noise, noise_mult * (size_t)batch * L * 3 * sizeof(float)));
    HIP_OK(hipMalloc((void**)&x_dec, (size_t)batch * L * sizeof(float)));
    HIP_OK(hipMalloc((void**)&codes, (size_t)batch * L * 3 * sizeof(float)));
    HIP_OK(hipMalloc((void**)&counts, (size_t)nbatch * 2 * sizeof(uint64_t)));
    uint64_t* hc = (uint64_t*)malloc((size_t)nbatch * 2 * sizeof(uint64_t));
    hipStream_t st;
    HIP_OK(hipStreamCreate(&st));

    const double t0 = now_s();
    for (int si = 0; si < snr_points; ++si) {
        const double snr = snr_points > 1 ? snr_lo + (snr_hi - snr_lo) * si / (snr_points - 1) : snr_lo;   /* trainer.py:157-158 */
        if (mode == 0) {
            TAE_CHECK(tae_eval_snr(h, (float)snr, batch, nbatch, (int64_t)si * nbatch * batch, seed, seed, counts, st));
        } else {
            HIP_OK(hipMemsetAsync(counts, 0, (size_t)nbatch * 2 * sizeof(uint64_t), st));
        }
        for (int bi = 0; mode != 0 && bi < nbatch; ++bi) {
            const int64_t first = ((int64_t)si * nbatch + bi) * batch;      /* global block index: the Philox key */
            if (kind == TAE_NOISE_AWGN) {
                TAE_CHECK(tae_generate_inputs(h, u, noise, batch, first, seed, seed, (float)snr, st));
            } else {
                float* nz_out = noise + (noise_mult - 1) * (size_t)batch * L * 3;
                TAE_CHECK(tae_generate_inputs(h, u, NULL, batch, first, seed, seed, (float)snr, st));
                TAE_CHECK(tae_generate_noise(h, &nz, (float)snr, nz_out, kind == TAE_NOISE_FADING ? noise : NULL, batch, first, seed, st));
            }
            TAE_CHECK(tae_forward(h, u, noise, x_dec, codes, batch, st));   /* Channel_AE.forward, channel_ae.py:32-78 */
            TAE_CHECK(tae_count_errors(h, x_dec, u, batch, counts + 2 * bi, st));
        }
        HIP_OK(hipMemcpyAsync(hc, counts, (size_t)nbatch * 2 * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
        HIP_OK(hipStreamSynchronize(st));
        double ber = 0.0, bler = 0.0;      /* mean over batches of the per-batch rates (trainer.py:176-177,215-216) */
        uint64_t bits = 0, blocks = 0;
        for (int bi = 0; bi < nbatch; ++bi) {
            ber += (double)hc[2 * bi] / ((double)batch * L);
            bler += (double)hc[2 * bi + 1] / (double)batch;
            bits += hc[2 * bi];
            blocks += hc[2 * bi + 1];
        }
        printf("snr %.6f bit_errors %llu block_errors %llu ber %.9e bler %.9e\n", snr, (unsigned long long)bits,
               (unsigned long long)blocks, ber / nbatch, bler / nbatch);
    }
    const double dt = now_s() - t0;
    int32_t prec = 0, overflow = 0;
    TAE_CHECK(tae_range_status(h, &prec, &overflow));
    printf("arithmetic %s overflow %d\n", prec == 1 ? "f16x2" : "f32", overflow & (TAE_RANGE_HIGH | TAE_RANGE_LOW));
    if (overflow & TAE_RANGE_FELL_BACK) printf("range fall-back: the fp32 kernels served the flagged call and every call after it\n");
    printf("blocks %ld seconds %.3f info_bits_per_s %.3e\n", (long)snr_points * nbatch * batch, dt,
           (double)snr_points * nbatch * batch * L / dt);

    HIP_OK(hipStreamDestroy(st));
    HIP_OK(hipFree(u)); HIP_OK(hipFree(noise)); HIP_OK(hipFree(x_dec)); HIP_OK(hipFree(codes)); HIP_OK(hipFree(counts));
    free(hc);
    TAE_CHECK(tae_destroy(h));
    return (overflow & (TAE_RANGE_HIGH | TAE_RANGE_LOW)) && !(overflow & TAE_RANGE_FELL_BACK) ? 4 : 0;
}
