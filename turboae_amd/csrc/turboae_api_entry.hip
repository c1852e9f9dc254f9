// libturboae_hip.so, host side 3 of 3 - the C ABI declared in include/turboae_hip.h (see turboae_host.hpp): error state, debug knobs, the
// fp32 fall-back wrapper and every compute / option entry point (the handle's life cycle is in turboae_api_create.hip).
// Every compute entry point is a short sequence of asynchronous kernel launches on the caller's stream (graph-capturable).
#include "turboae_host.hpp"


namespace {
thread_local std::string g_err;
std::mutex g_knob_mu;
std::vector<std::string> g_knobs;          // "NAME=value" of every debug knob that took effect in this process
}  // namespace

namespace tae {

namespace host {
int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}
const char* last_error() { return g_err.c_str(); }
std::string knob_report() {
    std::lock_guard<std::mutex> lk(g_knob_mu);
    std::string r;
    for (const std::string& k : g_knobs) { if (!r.empty()) r += ';'; r += k; }
    return r;
}
}  // namespace host

int fail_msg(int code, const char* msg) { return host::fail(code, msg ? msg : "?"); }       // for the library's other translation units

const char* debug_knob(const char* name) {
    const char* on = getenv("TAE_DEBUG_KNOBS");
    if (!on || on[0] != '1' || on[1] != 0) return nullptr;
    const char* v = getenv(name);
    if (!v) return nullptr;
    const std::string rec = std::string(name) + "=" + v;
    std::lock_guard<std::mutex> lk(g_knob_mu);
    if (std::find(g_knobs.begin(), g_knobs.end(), rec) == g_knobs.end()) g_knobs.push_back(rec);
    return v;
}

namespace host {

// tae_config.range_fallback: run `call` on the fp16-split handle, wait for it, read the range word, and if a launch left the window
// run the same call on the fp32 twin - from then on every call goes there (a network that left the window once will again).
template <class F>
int with_fallback(tae_handle* h, hipStream_t st, F&& call) {
    if (!h || !h->fb) return call(h);
    if (h->fb->cap < h->cap) return fail(TAE_ESTATE, "internal: the fp32 fall-back handle's workspace is smaller than the main handle's");
    if (h->last_flags & 4u) return call(h->fb);
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone)
        return fail(TAE_ESTATE, "a handle created with range_fallback synchronises after every call and cannot be captured into a hipGraph");
    int rc = call(h);
    if (rc != TAE_OK) return rc;
    uint32_t f = 0;
    TAE_HIP(hipStreamSynchronize(st));
    TAE_HIP(hipMemcpy(&f, h->d_flags, sizeof(f), hipMemcpyDeviceToHost));
    if ((f & 3u) == 0u) return TAE_OK;
    TAE_HIP(hipMemset(h->d_flags, 0, sizeof(f)));
    h->last_flags |= (f & 3u) | 4u;
    return call(h->fb);
}

}  // namespace host
}  // namespace tae

using namespace tae::host;

extern "C" {

int tae_abi_version(void) { return TAE_ABI_VERSION; }

const char* tae_last_error(void) { return tae::host::last_error(); }

int tae_set_interleaver(tae_handle* h, const int32_t* p, int32_t L) {
    if (!h || !p) return fail(TAE_EINVAL, "NULL argument");
    { const int rc_h = check_handle(h); if (rc_h != TAE_OK) return rc_h; }
    if (L != h->cfg.block_len) return fail(TAE_EINVAL, "interleaver length must equal block_len");
    std::vector<int32_t> inv(L, -1);
    for (int i = 0; i < L; ++i) {
        if (p[i] < 0 || p[i] >= L || inv[p[i]] != -1) return fail(TAE_EINVAL, "p is not a permutation of 0..L-1");
        inv[p[i]] = i;   // interleavers.py:29-33
    }
    TAE_HIP(hipDeviceSynchronize());
    TAE_HIP(hipMemcpy(h->d_perm, p, L * sizeof(int32_t), hipMemcpyHostToDevice));
    TAE_HIP(hipMemcpy(h->d_inv, inv.data(), L * sizeof(int32_t), hipMemcpyHostToDevice));
    if (h->fb) { const int rc_f = tae_set_interleaver(h->fb, p, L); if (rc_f != TAE_OK) return rc_f; }
    // The extrinsic values a trained decoder exchanges depend on the permutation: the synthetic calibration made at create (identity
    // permutation) is repeated ONCE, with the first permutation the caller installs.  Later permutations keep it - another random
    // permutation of the same network moves a layer maximum by far less than the window (2^-7 .. 2^5 around it; both ends stay
    // checked per launch), and -is_same_interleaver 0 installs one per forward (r04 re-measured on every one of them: two forward
    // passes of up to 768 blocks each time) - and a calibration on the caller's own data is never discarded behind its back.
    if (h->calibrated && !h->cal_user && !h->cal_perm) {
        const int rc = calibrate_range(h, nullptr, nullptr, 0);
        h->cal_perm = rc == TAE_OK;
        return rc;
    }
    return TAE_OK;
}

int tae_set_noise_opts(tae_handle* h, const tae_noise_opts* o) {
    if (!h) return fail(TAE_EINVAL, "handle is NULL");
    if (h->fb) { const int rc_f = tae_set_noise_opts(h->fb, o); if (rc_f != TAE_OK) return rc_f; }
    if (!o) { h->noise_opts = default_noise_opts(); return TAE_OK; }
    const int rc = check_noise_opts(o);
    if (rc != TAE_OK) return rc;
    h->noise_opts = *o;
    return TAE_OK;
}

int tae_generate_noise(tae_handle* h, const tae_noise_opts* opts, float test_sigma, float* noise, float* fading_h, int32_t B,
                       int64_t first_block, uint64_t seed, void* stream) {
    { const int rc_h = check_handle(h); if (rc_h != TAE_OK) return rc_h; }
    if (!opts || !noise) return fail(TAE_EINVAL, "NULL argument");
    if (B < 1 || first_block < 0) return fail(TAE_EINVAL, "bad block range");
    int rc = check_noise_opts(opts);
    if (rc != TAE_OK) return rc;
    if (opts->kind == TAE_NOISE_FADING && !fading_h) return fail(TAE_EINVAL, "TAE_NOISE_FADING writes the fading coefficients: fading_h is NULL");
    tae::NoiseGen g;
    rc = make_noise_gen(opts, test_sigma, &g);
    if (rc != TAE_OK) return rc;
    TAE_HIP(tae::launch_gen_noise(g, noise, fading_h, (size_t)B, (size_t)first_block, h->cfg.block_len, seed, (hipStream_t)stream));
    return TAE_OK;
}

int tae_set_channel_opts(tae_handle* h, const tae_channel_opts* o) {
    if (!h) return fail(TAE_EINVAL, "handle is NULL");
    if (h->fb) { const int rc_f = tae_set_channel_opts(h->fb, o); if (rc_f != TAE_OK) return rc_f; }
    const tae::NormOpts before = h->nopts;
    // what the decoder receives changes class with these fields (not with the running mean / std of norm_mode 2): measure again
    auto recalibrate = [&]() {
        const tae::NormOpts& a = before; const tae::NormOpts& b = h->nopts;
        const bool same = a.norm_mode == b.norm_mode && a.ste == b.ste && a.channel == b.channel && a.rec_quantize == b.rec_quantize &&
                          a.enc_truncate_limit == b.enc_truncate_limit && a.enc_value_limit == b.enc_value_limit &&
                          a.enc_quantize_level == b.enc_quantize_level && a.rec_quantize_limit == b.rec_quantize_limit &&
                          a.rec_quantize_level == b.rec_quantize_level;
        if (same || !h->calibrated || h->cal_user || h->dec_tails.empty()) return (int)TAE_OK;      // (a user calibration stays: the caller re-measures)
        if (check_handle(h) != TAE_OK) return (int)TAE_OK;      // no device context here: the next tae_set_interleaver / tae_calibrate_range measures
        return calibrate_range(h, nullptr, nullptr, 0);
    };
    if (!o) { h->nopts = default_norm_opts(); return recalibrate(); }
    if (o->struct_size != (int32_t)sizeof(tae_channel_opts)) return fail(TAE_EINVAL, "tae_channel_opts.struct_size mismatch (ABI)");
    if (o->norm_mode < 0 || o->norm_mode > 2) return fail(TAE_EINVAL, "norm_mode must be 0, 1 or 2");
    if (o->norm_mode == 2 && !(o->std > 0.0f)) return fail(TAE_EINVAL, "fixed std must be > 0");
    if (o->channel < 0 || o->channel > 3) return fail(TAE_EINVAL, "channel must be 0 (additive), 1 (bec), 2 (bsc/ge) or 3 (fading)");
    if (o->ste && (!(o->enc_value_limit > 0.0f) || o->enc_quantize_level < 2.0f)) return fail(TAE_EINVAL, "bad STE quantiser parameters");
    if (o->rec_quantize && (!(o->rec_quantize_limit > 0.0f) || o->rec_quantize_level < 2.0f)) return fail(TAE_EINVAL, "bad receive quantiser parameters");
    tae::NormOpts n;
    n.norm_mode = o->norm_mode; n.mean = o->mean; n.std = o->std;
    n.ste = o->ste; n.enc_value_limit = o->enc_value_limit; n.enc_quantize_level = o->enc_quantize_level;
    n.enc_truncate_limit = o->enc_truncate_limit;
    n.channel = o->channel; n.rec_quantize = o->rec_quantize;
    n.rec_quantize_limit = o->rec_quantize_limit; n.rec_quantize_level = o->rec_quantize_level;
    h->nopts = n;
    return recalibrate();
}

int tae_encode_prenorm(tae_handle* h, const float* u, float* x_tx, double* stats3, int32_t B, void* stream) {
    int rc = check_batch(h, B);
    if (rc != TAE_OK) return rc;
    if (!u || !x_tx || !stats3) return fail(TAE_EINVAL, "NULL tensor");
    return with_fallback(h, (hipStream_t)stream, [&](tae_handle* e) { return run_encoder(e, u, x_tx, stats3, B, (hipStream_t)stream); });
}

int tae_normalize(tae_handle* h, const float* x_tx, const double* stats3, const float* noise, float* codes, float* received,
                  int32_t B, void* stream) {
    int rc = check_batch(h, B);
    if (rc != TAE_OK) return rc;
    if (!x_tx || !stats3) return fail(TAE_EINVAL, "NULL tensor");
    if ((received != nullptr) != (noise != nullptr)) return fail(TAE_EINVAL, "noise and received must be given together");
    if (!codes && !received) return fail(TAE_EINVAL, "nothing to write");
    TAE_HIP(tae::launch_normalize(x_tx, stats3, noise, codes, received, (size_t)B * h->cfg.block_len * 3, h->nopts, (hipStream_t)stream));
    return TAE_OK;
}

int tae_encode(tae_handle* h, const float* u, float* codes, int32_t B, void* stream) {
    int rc = check_batch(h, B);
    if (rc != TAE_OK) return rc;
    if (!u || !codes) return fail(TAE_EINVAL, "NULL tensor");
    return with_fallback(h, (hipStream_t)stream, [&](tae_handle* e) {
        const int r = run_encoder(e, u, e->d_xtx, e->d_stats, B, (hipStream_t)stream);
        if (r != TAE_OK) return r;
        return tae_normalize(e, e->d_xtx, e->d_stats, nullptr, codes, nullptr, B, stream);
    });
}

int tae_decode(tae_handle* h, const float* received, float* x_dec, int32_t B, void* stream) {
    int rc = check_batch(h, B);
    if (rc != TAE_OK) return rc;
    if (!received || !x_dec) return fail(TAE_EINVAL, "NULL tensor");
    return with_fallback(h, (hipStream_t)stream, [&](tae_handle* e) { return run_decoder(e, received, x_dec, B, (hipStream_t)stream); });
}

int tae_decode_taps(tae_handle* h, const float* received, float* x_dec, float* taps, int32_t B, void* stream) {
    int rc = check_batch(h, B);
    if (rc != TAE_OK) return rc;
    if (!received || !x_dec || !taps) return fail(TAE_EINVAL, "NULL tensor");
    if (h->cfg.dense && !h->gen) return fail(TAE_EINVAL, "tae_decode_taps: not built for DenseSameShapeConv1d stacks");
    return run_decoder(h, received, x_dec, B, (hipStream_t)stream, taps);
}

int tae_forward(tae_handle* h, const float* u, const float* noise, float* x_dec, float* codes, int32_t B, void* stream) {
    int rc = check_batch(h, B);
    if (rc != TAE_OK) return rc;
    if (!u || !noise || !x_dec) return fail(TAE_EINVAL, "NULL tensor");
    hipStream_t st = (hipStream_t)stream;
    return with_fallback(h, st, [&](tae_handle* e) {
        int r = run_encoder(e, u, e->d_xtx, e->d_stats, B, st);
        if (r != TAE_OK) return r;
        r = tae_normalize(e, e->d_xtx, e->d_stats, noise, codes, e->d_rx, B, stream);
        if (r != TAE_OK) return r;
        return run_decoder(e, e->d_rx, x_dec, B, st);
    });
}

// One SNR point of trainer.test (trainer.py:160-217) on the device: per batch generate inputs -> encoder -> power constraint with
// that batch's statistics -> AWGN; the received blocks of a group of batches are decoded in one call (the decoder never mixes
// blocks) and the errors are counted per batch.
static int eval_snr_impl(tae_handle* h, float snr_db, int32_t batch, int32_t n_batches, int64_t first_block, uint64_t seed_bits,
                         uint64_t seed_noise, uint64_t* counts, void* stream);
int tae_eval_snr(tae_handle* h, float snr_db, int32_t batch, int32_t n_batches, int64_t first_block, uint64_t seed_bits,
                 uint64_t seed_noise, uint64_t* counts, void* stream) {
    return with_fallback(h, (hipStream_t)stream, [&](tae_handle* e) {
        return eval_snr_impl(e, snr_db, batch, n_batches, first_block, seed_bits, seed_noise, counts, stream);
    });
}
static int eval_snr_impl(tae_handle* h, float snr_db, int32_t batch, int32_t n_batches, int64_t first_block, uint64_t seed_bits,
                         uint64_t seed_noise, uint64_t* counts, void* stream) {
    if (!h || !counts) return fail(TAE_EINVAL, "NULL argument");
    { const int rc_h = check_handle(h); if (rc_h != TAE_OK) return rc_h; }
    if (batch < 1 || n_batches < 1 || first_block < 0) return fail(TAE_EINVAL, "bad batch geometry");
    // the generator must produce what the configured channel consumes (channel_ae.py:41-56): masks for bec / bsc / ge, fading
    // coefficients for fading, additive noise otherwise
    const int nk = h->noise_opts.kind;
    const bool mask_kind = nk == TAE_NOISE_BEC || nk == TAE_NOISE_BSC || nk == TAE_NOISE_GE;
    if ((h->nopts.channel == 1 || h->nopts.channel == 2) != mask_kind || (h->nopts.channel == 3) != (nk == TAE_NOISE_FADING))
        return fail(TAE_EINVAL, "tae_eval_snr: the noise generator (tae_set_noise_opts) does not match the channel (tae_set_channel_opts)");
    const size_t noise_mult = nk == TAE_NOISE_FADING ? 2 : 1;      // fading: coefficients followed by the noise
    hipStream_t st = (hipStream_t)stream;
    const size_t L = h->cfg.block_len;
    int64_t group = (24576 + batch - 1) / batch;           // batches per decoder call: about 24 576 blocks
    if (group > n_batches) group = n_batches;
    if (group * batch > h->cap || group * batch > h->eval_group_blocks || batch > h->eval_batch || (noise_mult == 2 && !h->eval_noise_x2)) {
        // workspace growth: synchronises and allocates (first call for a geometry only - afterwards the call only enqueues work)
        int rc = tae_reserve(h, (int32_t)(group * batch));
        if (rc != TAE_OK) return rc;
        TAE_HIP(hipDeviceSynchronize());
        (void)hipFree(h->d_eval_u); (void)hipFree(h->d_eval_noise); (void)hipFree(h->d_eval_xdec);
        h->d_eval_u = h->d_eval_noise = h->d_eval_xdec = nullptr;
        h->eval_group_blocks = 0; h->eval_batch = 0;
        TAE_HIP(hipMalloc(&h->d_eval_u, (size_t)group * batch * L * sizeof(float)));
        TAE_HIP(hipMalloc(&h->d_eval_xdec, (size_t)group * batch * L * sizeof(float)));
        TAE_HIP(hipMalloc(&h->d_eval_noise, noise_mult * (size_t)batch * L * 3 * sizeof(float)));
        h->eval_noise_x2 = noise_mult == 2;
        h->eval_group_blocks = group * batch;
        h->eval_batch = batch;
    }
    TAE_HIP(hipMemsetAsync(counts, 0, (size_t)n_batches * 2 * sizeof(uint64_t), st));
    for (int64_t g0 = 0; g0 < n_batches; g0 += group) {
        const int64_t ng = g0 + group <= n_batches ? group : n_batches - g0;
        for (int64_t i = 0; i < ng; ++i) {
            float* u = h->d_eval_u + (size_t)i * batch * L;
            const int64_t fb = first_block + (g0 + i) * batch;
            int rc = tae_generate_inputs(h, u, nk == TAE_NOISE_AWGN ? h->d_eval_noise : nullptr, batch, fb, seed_bits, seed_noise, snr_db, stream);
            if (rc != TAE_OK) return rc;
            if (nk != TAE_NOISE_AWGN) {
                float* nz = h->d_eval_noise + (noise_mult - 1) * (size_t)batch * L * 3;
                rc = tae_generate_noise(h, &h->noise_opts, snr_db, nz, nk == TAE_NOISE_FADING ? h->d_eval_noise : nullptr, batch, fb, seed_noise, stream);
                if (rc != TAE_OK) return rc;
            }
            rc = run_encoder(h, u, h->d_xtx, h->d_stats, batch, st);
            if (rc != TAE_OK) return rc;
            rc = tae_normalize(h, h->d_xtx, h->d_stats, h->d_eval_noise, nullptr, h->d_rx + (size_t)i * batch * L * 3, batch, stream);
            if (rc != TAE_OK) return rc;
        }
        int rc = run_decoder(h, h->d_rx, h->d_eval_xdec, (int32_t)(ng * batch), st);
        if (rc != TAE_OK) return rc;
        for (int64_t i = 0; i < ng; ++i)
            TAE_HIP(tae::launch_count_errors(h->d_eval_xdec + (size_t)i * batch * L, h->d_eval_u + (size_t)i * batch * L, batch, (int)L,
                                             (unsigned long long*)(counts + 2 * (g0 + i)), st));
    }
    return TAE_OK;
}

int tae_count_errors(tae_handle* h, const float* x_dec, const float* u, int32_t B, uint64_t* counts2, void* stream) {
    if (!h || !x_dec || !u || !counts2) return fail(TAE_EINVAL, "NULL argument");
    { const int rc_h = check_handle(h); if (rc_h != TAE_OK) return rc_h; }
    if (B < 1) return fail(TAE_EINVAL, "batch must be >= 1");
    TAE_HIP(tae::launch_count_errors(x_dec, u, B, h->cfg.block_len, (unsigned long long*)counts2, (hipStream_t)stream));
    return TAE_OK;
}

int tae_generate_inputs(tae_handle* h, float* u, float* noise, int32_t B, int64_t first_block, uint64_t seed_bits,
                        uint64_t seed_noise, float snr_db, void* stream) {
    { const int rc_h = check_handle(h); if (rc_h != TAE_OK) return rc_h; }
    if (B < 1 || first_block < 0) return fail(TAE_EINVAL, "bad block range");
    if (!u && !noise) return fail(TAE_EINVAL, "nothing to write");
    const float sigma = (float)pow(10.0, -(double)snr_db / 20.0);   // utils.py:69-70
    const size_t L = h->cfg.block_len;
    TAE_HIP(tae::launch_gen_inputs(u, noise, (size_t)B * L, (size_t)first_block * L, seed_bits, seed_noise, sigma, (hipStream_t)stream));
    return TAE_OK;
}

int tae_kernel_info(tae_handle* h, int32_t* blocks_per_workgroup, int32_t* lds_bytes) {
    if (!h) return fail(TAE_EINVAL, "handle is NULL");
    if (h->gen) {          // generic fp32 kernels: no fused geometry to report
        if (blocks_per_workgroup) *blocks_per_workgroup = 0;
        if (lds_bytes) *lds_bytes = 0;
        return TAE_OK;
    }
    if (blocks_per_workgroup) *blocks_per_workgroup = h->nbd;        // the decoder's (the dominant kernel)
    if (lds_bytes) *lds_bytes = h->nbd >= 1 ? (h->prec == 1 ? h->lds_bytes_hd : h->lds_bytes_d) : (h->prec == 1 ? h->dec_lds_h : h->dec_lds);
    return TAE_OK;
}

int tae_kernel_variants(tae_handle* h, int32_t* enc_both, int32_t* dec_both) {
    { const int rc_h = check_handle(h); if (rc_h != TAE_OK) return rc_h; }
    const bool whole = !h->gen && h->prec == 1;
    if (enc_both) *enc_both = (whole && h->cfg.enc_type == 0 && h->nb >= 1) ? base_params(h, 1, false).head2 : 0;
    if (dec_both) *dec_both = (whole && h->cfg.dec_type == 0 && h->nbd >= 1) ? base_params(h, 1, true).head2 : 0;
    return TAE_OK;
}

int tae_overrides(tae_handle*, char* buf, int32_t n) {
    const std::string r = tae::host::knob_report();
    if (buf && n > 0) {
        const size_t m = std::min((size_t)n - 1, r.size());
        memcpy(buf, r.data(), m);
        buf[m] = 0;
    }
    return (int)r.size();
}

int tae_debug_split_f16(const float* x, size_t n, float scale, uint16_t* hi, uint16_t* lo) {
    if (!x || !hi || !lo) return fail(TAE_EINVAL, "NULL argument");
    for (size_t i = 0; i < n; ++i) {
        const float w = x[i] * scale;
        hi[i] = f2h(w);
        lo[i] = f2h(w - h2f(hi[i]));
    }
    return TAE_OK;
}

int tae_range_status(tae_handle* h, int32_t* precision, int32_t* overflow) {
    { const int rc_h = check_handle(h); if (rc_h != TAE_OK) return rc_h; }
    if (precision) *precision = h->x1 ? TAE_PREC_F16X1 : h->prec;
    if (overflow) {
        uint32_t f = 0;
        // the launches whose flag is read may sit on any stream (torch side streams are non-blocking: the null-stream copy
        // below does not order behind them) - wait for the whole device first
        TAE_HIP(hipDeviceSynchronize());
        TAE_HIP(hipMemcpy(&f, h->d_flags, sizeof(f), hipMemcpyDeviceToHost));
        if (f) TAE_HIP(hipMemset(h->d_flags, 0, sizeof(f)));
        *overflow = (int32_t)((f & 3u) | h->last_flags);
        h->last_flags &= 4u;           // the fall-back is permanent, the range bits it reacted to are reported once
    }
    return TAE_OK;
}

int tae_calibrate_range(tae_handle* h, const float* u, const float* noise, int32_t B) {
    { const int rc_h = check_handle(h); if (rc_h != TAE_OK) return rc_h; }
    if ((u == nullptr) != (noise == nullptr)) return fail(TAE_EINVAL, "u and noise must be given together (both NULL: the synthetic batch)");
    if (u && B < 1) return fail(TAE_EINVAL, "batch must be >= 1");
    const int rc = calibrate_range(h, u, noise, B);
    if (rc == TAE_OK) {
        h->cal_user = u != nullptr;           // the caller's data define the window from here on (a synthetic re-measurement gives it back)
        if (!u) h->cal_perm = true;
    }
    return rc;
}

int tae_range_info(tae_handle* h, int32_t* n_encoder, int32_t* n_decoder, int32_t* exponents, int32_t capacity, int32_t* passes) {
    if (!h) return fail(TAE_EINVAL, "handle is NULL");
    const int32_t ne = (int32_t)(h->enc_A.size() + h->enc_Ax.size()), nd = (int32_t)(h->dec_A.size() + h->dec_Ax.size());
    if (n_encoder) *n_encoder = h->calibrated ? ne : 0;
    if (n_decoder) *n_decoder = h->calibrated ? nd : 0;
    if (passes) *passes = h->cal_passes;
    if (exponents && h->calibrated) {
        if (capacity < ne + nd) return fail(TAE_EINVAL, "tae_range_info: capacity too small");
        int32_t* o = exponents;
        for (int v : h->enc_Ax) *o++ = v;
        for (int v : h->enc_A) *o++ = v;
        for (int v : h->dec_Ax) *o++ = v;
        for (int v : h->dec_A) *o++ = v;
    }
    return TAE_OK;
}

}  // extern "C"

