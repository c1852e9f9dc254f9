// libturboae_hip.so, host side 1 of 3 - the handle's life cycle (see turboae_host.hpp): configuration checks, weight packing into the
// kernels' fragment layouts, range calibration of the fp16-split kernels, and tae_num_weights / tae_create / tae_destroy / tae_reserve.
#include "turboae_host.hpp"

namespace tae {
namespace host {
namespace {

struct Layout {   // must mirror tae::Geo<U> in turboae_kernels.hip
    int U, CT, CP, nch_mid, midf, l0f, sup, sfm, sf0;
    explicit Layout(int u) : U(u) {
        CT = (U + 15) / 16;
        CP = CT * 16;
        nch_mid = (5 * U + 7) / 8;
        midf = nch_mid * CT * 128;
        l0f = 5 * CT * 128;
        sup = (U % 16) == 4;
        sfm = sup ? U * 128 : 0;       // super-tile A fragments of a U->U layer: K' = 8 shifts x U -> U chunks of 8
        sf0 = sup ? 8 * 128 : 0;       // first layer: 8 shifts x 8 padded inputs
    }
    size_t stack_stride(int n_layer) const {
        return (size_t)l0f + CP + sf0 + (size_t)(n_layer - 1) * (midf + CP + sfm) + 8 * CP + 8;
    }
};

// im2col-flattened weight W'[co][k], k = tap * cin_pad + ci (tap-major), zero outside the real tensor.
inline float wflat(const float* W, int U, int cin, int cin_pad, int co, int k, int taps = 5) {
    const int j = k / cin_pad, ci = k % cin_pad;
    return (co < U && j < taps && ci < cin) ? W[((size_t)co * cin + ci) * taps + j] : 0.0f;
}

// Tile one Conv1d weight (U, cin, 5) into MFMA A-fragment order.  Lane (i = lane & 15, kq = lane >> 4)
// of k-step s of chunk c holds W'[co = ct*16 + i][k = 8*c + 2*kq + s].  Per chunk the channel tiles are
// stored in pairs so that one 16-byte load fetches two tiles: [pair j][lane][ct=2j: s0 s1 | ct=2j+1: s0 s1],
// followed (odd CT) by the last tile as [lane][s0 s1].
void pack_conv(const float* W, int U, int cin, int cin_pad, int nch, int CT, float* dst) {
    for (int c = 0; c < nch; ++c)
        for (int ct = 0; ct < CT; ++ct)
            for (int lane = 0; lane < 64; ++lane)
                for (int s = 0; s < 2; ++s) {
                    size_t idx = (size_t)c * CT * 128;
                    if (ct < 2 * (CT / 2)) idx += (size_t)(ct / 2) * 256 + lane * 4 + (ct % 2) * 2 + s;
                    else idx += (size_t)(CT / 2) * 256 + lane * 2 + s;
                    dst[idx] = wflat(W, U, cin, cin_pad, ct * 16 + (lane & 15), 8 * c + 2 * (lane >> 4) + s);
                }
}

// Super-tile A fragments for the remainder channels (see tae::super_accumulate): row r = 4*s + c is channel
// crem + c at position shift s, K' = u * cin_pad + ci, A[r][K'] = W[crem + c][ci][u - s] for 0 <= u - s <= 4.
// Stored per chunk PAIR as [pair][lane][even chunk: k-step 0, 1 | odd chunk: k-step 0, 1].
void pack_super(const float* W, int U, int cin, int cin_pad, int nch, int crem, float* dst) {
    for (int c = 0; c < nch; ++c)
        for (int lane = 0; lane < 64; ++lane)
            for (int st = 0; st < 2; ++st) {
                const int r = lane & 15, sh = r / 4, ch = r % 4;
                const int kp = 8 * c + 2 * (lane >> 4) + st;
                const int u = kp / cin_pad, ci = kp % cin_pad, j = u - sh;
                float v = 0.0f;
                if (u < 8 && j >= 0 && j <= 4 && ci < cin && crem + ch < U) v = W[((size_t)(crem + ch) * cin + ci) * 5 + j];
                dst[((size_t)(c / 2) * 64 + lane) * 4 + (c % 2) * 2 + st] = v;
            }
}

// canonical stack (conv layers then Linear head) -> packed stack; returns floats consumed from src
size_t pack_stack(const float* src, const Layout& lo, int n_layer, int cin0, int nout, float* dst) {
    const float* s = src;
    float* d = dst;
    for (int l = 0; l < n_layer; ++l) {
        const int cin = l == 0 ? cin0 : lo.U;
        pack_conv(s, lo.U, cin, l == 0 ? 8 : lo.U, l == 0 ? 5 : lo.nch_mid, lo.CT, d);
        d += l == 0 ? lo.l0f : lo.midf;
        const float* b = s + (size_t)lo.U * cin * 5;
        for (int c = 0; c < lo.CP; ++c) d[c] = c < lo.U ? b[c] : 0.0f;
        d += lo.CP;
        if (lo.sup) {
            pack_super(s, lo.U, cin, l == 0 ? 8 : lo.U, l == 0 ? 8 : lo.U, 16 * (lo.CT - 1), d);
            d += l == 0 ? lo.sf0 : lo.sfm;
        }
        s = b + lo.U;
    }
    for (int f = 0; f < 8; ++f)
        for (int c = 0; c < lo.CP; ++c) d[f * lo.CP + c] = (f < nout && c < lo.U) ? s[(size_t)f * lo.U + c] : 0.0f;
    d += 8 * lo.CP;
    s += (size_t)nout * lo.U;
    for (int f = 0; f < 8; ++f) d[f] = f < nout ? s[f] : 0.0f;
    s += nout;
    return (size_t)(s - src);
}

// ---- f16x2 representation (turboae_h2.hip) -----------------------------------------------------------------
struct LayoutH {   // must mirror tae::GeoH<U> / tae::tap_geo<U>(taps)
    int U, CT, CP, nsl_mid, nsl_l0, taps;
    uint32_t slb, midb, l0b, tailb;
    bool tail20 = false;      // whole-block 100-wide 5-tap engines: every layer's last slab in the two-MFMA form (GeoH<100>::TAIL20, run_stack_h<.., T20>)
    explicit LayoutH(int u, int taps_ = 5) : U(u), taps(taps_) {
        CT = (U + 15) / 16;
        CP = CT * 16;
        nsl_mid = (taps * U + 31) / 32;
        nsl_l0 = (taps * 8 + 31) / 32;
        slb = (uint32_t)CT * 2048u;
        midb = (uint32_t)nsl_mid * slb;
        l0b = (uint32_t)nsl_l0 * slb;
        tailb = (uint32_t)CP * 4u + 32u;
    }
    size_t stack_bytes(int n_layer) const {
        return (size_t)l0b + (size_t)(n_layer - 1) * midb + (size_t)n_layer * tailb + (size_t)8 * CP * 4 + 32;
    }
};

// fp32 -> fp16 bits, round to nearest even, denormals kept (the device side uses v_cvt_f16_f32 in the default mode)
// (f2h / h2f: turboae_host.hpp)
inline float pow2_scale(float maxabs) {       // power of two that brings maxabs into [2^13, 2^14)
    if (!(maxabs > 0.0f) || !std::isfinite(maxabs)) return 1.0f;
    int e = 0;
    (void)frexpf(maxabs, &e);
    int S = 14 - e;
    if (S > 60) S = 60;
    if (S < -60) S = -60;
    return ldexpf(1.0f, S);
}
inline float max_abs(const float* p, size_t n) {
    float m = 0.0f;
    for (size_t i = 0; i < n; ++i) m = fmaxf(m, fabsf(p[i]));
    return m;
}

// Conv1d weight (U, cin, 5) -> [slab][channel tile][hi | lo][lane][8 halves]: lane (i = lane & 15, kq = lane >> 4) holds
// W'[co = 16 ct + i][k = 32 slab + 8 kq + j] * scale, split into hi = f16(w), lo = f16(w - hi)
// tail20 (100-channel, 5-tap U -> U layers: K = 500 = 15 slabs + 20 k): the last slab in the two-MFMA form of conv_accumulate_h -
// where its hi fragments go: A1 = [hi k 4kq.. | hi k 4kq..]; in place of its lo fragments: A2 = [lo k 4kq.. | X(kq)] with
// X = hi, hi, lo, 0 (kq = 0..3) of k 16..19 (k relative to the slab)
void pack_conv_h(const float* W, int U, int cin, int cin_pad, int nslab, int CT, float scale, uint16_t* dst, int taps = 5, bool tail20 = false) {
    for (int sl = 0; sl < nslab; ++sl)
        for (int ct = 0; ct < CT; ++ct)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 8; ++j) {
                    const size_t base = ((size_t)(sl * CT + ct) * 2) * 512 + (size_t)lane * 8 + j;
                    const int kq = lane >> 4;
                    auto split = [&](int k, uint16_t& hi, uint16_t& lo) {
                        const float w = wflat(W, U, cin, cin_pad, ct * 16 + (lane & 15), k, taps) * scale;
                        hi = f2h(w);
                        lo = f2h(w - h2f(hi));
                    };
                    uint16_t hi, lo;
                    if (tail20 && sl == nslab - 1) {
                        uint16_t h0, l0, h1 = 0, l1 = 0;
                        split(32 * sl + 4 * kq + (j & 3), h0, l0);
                        if (kq < 3) split(32 * sl + 16 + (j & 3), h1, l1);
                        dst[base] = h0;                                                        // A1: both quarters hi of k 4kq..
                        dst[base + 512] = j < 4 ? l0 : (kq < 2 ? h1 : (kq == 2 ? l1 : (uint16_t)0));      // A2
                        continue;
                    }
                    split(32 * sl + 8 * kq + j, hi, lo);
                    dst[base] = hi;
                    dst[base + 512] = lo;
                }
}

void tail_values(const TailRef& t, int A_in, int A_out, float low, float high, int elu_kind, std::vector<float>& out) {
    const size_t CP = t.bias_s.size();
    out.assign(CP + 8, 0.0f);
    for (size_t c = 0; c < CP; ++c) out[c] = ldexpf(t.bias_s[c], A_in);
    out[CP] = ldexpf(1.0f, -(t.S + A_in));
    out[CP + 1] = ldexpf(1.0f, A_out);
    out[CP + 2] = low;
    out[CP + 3] = high;
    out[CP + 4] = (float)elu_kind;
}

// canonical stack -> f16x2 packed stack (bytes at dst); returns floats consumed from src.  `tails` (optional) receives one
// TailRef per layer, offsets relative to `base_off` = byte offset of dst inside the side's buffer.
size_t pack_stack_h(const float* src, const LayoutH& lo, int n_layer, int cin0, int nout, char* dst, std::vector<TailRef>* tails = nullptr,
                    uint32_t base_off = 0) {
    const float* s = src;
    char* d = dst;
    for (int l = 0; l < n_layer; ++l) {
        const int cin = l == 0 ? cin0 : lo.U;
        const size_t nw = (size_t)lo.U * cin * lo.taps;
        float maxabs = 0.0f;
        for (size_t i = 0; i < nw; ++i) maxabs = fmaxf(maxabs, fabsf(s[i]));
        int e = 0;
        if (maxabs > 0.0f && std::isfinite(maxabs)) (void)frexpf(maxabs, &e);    // maxabs in [2^(e-1), 2^e)
        int S = (maxabs > 0.0f) ? 14 - e : 0;                                    // scaled max in [2^13, 2^14)
        if (S > 60) S = 60;
        if (S < -60) S = -60;
        const float scale = ldexpf(1.0f, S), inv = ldexpf(1.0f, -S);
        if (l == 0 && cin0 == 1) {
            // the fold covers taps 0..11 (fold_enc_input writes channels a = 1, 2: tap q + 4a, q < 4); check_cfg keeps the f16x2 kernels
            // at <= 9 taps - a wider kernel must never reach this packing
            if (lo.taps > 12) { fprintf(stderr, "libturboae_hip: internal error: encoder tap fold supports <= 12 taps, got %d\n", lo.taps); abort(); }
            // encoder stacks: all taps folded into ONE slab (turboae_h2.hip::fold_enc_input): k = 8 q + a holds tap q + 4 a.  The
            // region keeps its nsl_l0 slabs (offsets unchanged); the kernel walks the first one only.
            std::vector<float> wf((size_t)lo.U * 32, 0.0f);            // as a (U, cin = 4 taps-of-8 ..) tensor: W'[co][k], k < 32
            for (int co = 0; co < lo.U; ++co)
                for (int tap = 0; tap < lo.taps; ++tap) wf[(size_t)co * 32 + 8 * (tap % 4) + tap / 4] = s[(size_t)co * lo.taps + tap];
            // pack_conv_h reads W[(co * cin + ci) * taps + j] with k = j * cin_pad + ci: taps = 1, cin = cin_pad = 32 gives k = ci
            std::vector<uint16_t> one((size_t)lo.CT * 2 * 512);
            pack_conv_h(wf.data(), lo.U, 32, 32, 1, lo.CT, scale, one.data(), 1);
            memset(d, 0, lo.l0b);
            memcpy(d, one.data(), one.size() * sizeof(uint16_t));
        } else
        pack_conv_h(s, lo.U, cin, l == 0 ? 8 : lo.U, l == 0 ? lo.nsl_l0 : lo.nsl_mid, lo.CT, scale, reinterpret_cast<uint16_t*>(d), lo.taps,
                    lo.tail20);       // first layer 1 slab + tail (8 real k), U -> U layers 15 + tail (20 real k)
        d += l == 0 ? lo.l0b : lo.midb;
        const float* b = s + nw;
        float* t = reinterpret_cast<float*>(d);
        for (int c = 0; c < lo.CP; ++c) t[c] = c < lo.U ? b[c] * scale : 0.0f;
        t[lo.CP] = inv; t[lo.CP + 1] = 1.0f; t[lo.CP + 2] = 0.0f; t[lo.CP + 3] = 65504.0f;      // exponents 0, ELU kind 0 (t[CP + 4]) until calibrate_range
        if (tails) {
            TailRef r;
            r.off = base_off + (uint32_t)(d - dst);
            r.S = S;
            r.bias_s.assign(t, t + lo.CP);
            tails->push_back(std::move(r));
        }
        d += lo.tailb;
        s = b + lo.U;
    }
    float* t = reinterpret_cast<float*>(d);
    for (int f = 0; f < 8; ++f)
        for (int c = 0; c < lo.CP; ++c) t[f * lo.CP + c] = (f < nout && c < lo.U) ? s[(size_t)f * lo.U + c] : 0.0f;
    t += 8 * lo.CP;
    s += (size_t)nout * lo.U;
    for (int f = 0; f < 8; ++f) t[f] = f < nout ? s[f] : 0.0f;
    s += nout;
    return (size_t)(s - src);
}


}  // namespace

int check_cfg(const tae_config* c) {
    if (!c) return fail(TAE_EINVAL, "config is NULL");
    if (c->struct_size != (int32_t)sizeof(tae_config)) return fail(TAE_EINVAL, "tae_config.struct_size mismatch (ABI)");
    if ((c->range_calibration | c->range_fallback) & ~1) return fail(TAE_EINVAL, "range_calibration / range_fallback must be 0 or 1");
    if (c->enc_rnn < 0 || c->enc_rnn > 2 || c->dec_rnn < 0 || c->dec_rnn > 2) return fail(TAE_EINVAL, "enc_rnn / dec_rnn must be TAE_RNN_GRU, TAE_RNN_LSTM or TAE_RNN_RNN");
    if (tae::generic_needed(c)) {          // outside the MFMA kernels' envelope: the generic fp32 kernels' (wider) limits apply
        if (const char* msg = tae::generic_check(c)) return fail(TAE_EINVAL, msg);
        if (c->enc_num_layer < 1 || c->dec_num_layer < 1 || c->num_iteration < 1) return fail(TAE_EINVAL, "layer/iteration counts must be >= 1");
        if (c->block_len < 1) return fail(TAE_EINVAL, "block_len must be >= 1");
        if (c->enc_act < 0 || c->enc_act > 5 || c->dec_act < 0 || c->dec_act > 5) return fail(TAE_EINVAL, "enc_act / dec_act must be a TAE_ACT_* code");
        if ((c->dec_type | 1) != 1 || (c->enc_type | 1) != 1 || (c->dense | 1) != 1) return fail(TAE_EINVAL, "dec_type / enc_type / dense must be 0 or 1");
        if (c->precision != TAE_PREC_AUTO && c->precision != TAE_PREC_F32)
            return fail(TAE_EINVAL, "precision must be TAE_PREC_AUTO (0) or TAE_PREC_F32 (1) (TAE_PREC_F16X1 exists for the 100-wide whole-block CNN decoder only)");
        return TAE_OK;
    }
    for (int ks : {c->enc_kernel_size, c->dec_kernel_size}) {
        if (ks != 1 && ks != 3 && ks != 5 && ks != 7 && ks != 9) return fail(TAE_EINVAL, "kernel_size must be 1, 3, 5, 7 or 9");
        if (ks > 5 && ((c->precision != TAE_PREC_AUTO && c->precision != TAE_PREC_F16X1) || c->dense))
            return fail(TAE_EINVAL, "kernel sizes 7 and 9 are built in the fp16-split kernels only (precision = TAE_PREC_AUTO, no dense stacks)");
    }
    if (c->enc_num_unit < 1 || c->enc_num_unit > 124 || c->dec_num_unit < 1 || c->dec_num_unit > 124)
        return fail(TAE_EINVAL, "channel width (enc_num_unit, dec_num_unit) must be in 1..124 (kernels exist for 32 / 64 / 100 / 124; narrower stacks are embedded)");
    if (c->dense && (kernel_width(c->enc_num_unit) != c->enc_num_unit || kernel_width(c->dec_num_unit) != c->dec_num_unit))
        return fail(TAE_EINVAL, "dense stacks need a channel width of 32, 64 or 100");
    if (c->enc_num_layer < 1 || c->dec_num_layer < 1 || c->num_iteration < 1) return fail(TAE_EINVAL, "layer/iteration counts must be >= 1");
    if (c->num_iter_ft < 1 || c->num_iter_ft > 6) return fail(TAE_EINVAL, "num_iter_ft must be in 1..6");
    if (c->block_len < 1) return fail(TAE_EINVAL, "block_len must be >= 1");
    if (c->enc_act < 0 || c->enc_act > 5) return fail(TAE_EINVAL, "enc_act must be 0 (elu), 1 (linear), 2 (tanh), 3 (relu), 4 (selu) or 5 (sigmoid)");
    if (c->dec_act < 0 || c->dec_act > 5) return fail(TAE_EINVAL, "dec_act must be 0 (elu), 1 (linear), 2 (tanh), 3 (relu), 4 (selu) or 5 (sigmoid)");
    if (c->dec_type != 0 && c->dec_type != 1) return fail(TAE_EINVAL, "dec_type must be 0 (cnn) or 1 (rnn/gru)");
    if (c->precision != TAE_PREC_AUTO && c->precision != TAE_PREC_F32 && c->precision != TAE_PREC_F16X1)
        return fail(TAE_EINVAL, "precision must be TAE_PREC_AUTO (0), TAE_PREC_F32 (1) or TAE_PREC_F16X1 (2)");
    if (c->precision == TAE_PREC_F16X1 && (c->dec_type != 0 || c->dense || c->dec_kernel_size > 5 || c->dec_num_unit < 65 || c->dec_num_unit > 100))
        return fail(TAE_EINVAL, "TAE_PREC_F16X1 (one fp16 product, NOT fp32-grade) is instantiated for the 100-wide whole-block CNN decoder only: "
                                "dec_type = 0, no dense stacks, dec_kernel_size <= 5, 65 <= dec_num_unit <= 100");
    if (c->enc_type != 0 && c->enc_type != 1) return fail(TAE_EINVAL, "enc_type must be 0 (cnn) or 1 (rnn/gru)");
    if (c->dense != 0 && c->dense != 1) return fail(TAE_EINVAL, "dense must be 0 or 1");
    if (c->dense && (c->enc_type != 0 || c->dec_type != 0 || c->precision != TAE_PREC_AUTO))
        return fail(TAE_EINVAL, "DenseSameShapeConv1d stacks need the CNN encoder / decoder and precision = TAE_PREC_AUTO (fp16-split long-block kernels)");
    if (c->enc_type == 1 && (c->dec_type != 1 || c->enc_num_layer != 2))
        return fail(TAE_EINVAL, "the GRU encoder needs the GRU decoder (dec_type = 1) and enc_num_layer = 2");
    return TAE_OK;
}

// canonical GRU stack: per layer l and direction d: weight_ih (3H,cin) weight_hh (3H,H) bias_ih (3H) bias_hh (3H);
// then Linear (nout,2H), bias (nout)   (turboae_amd/weights.py canonical_entries)
size_t rnn_stack_floats(size_t H, size_t cin0, size_t nout, size_t G = 3) {
    size_t n = 0;
    for (int l = 0; l < 2; ++l) {
        const size_t cin = l == 0 ? cin0 : 2 * H;
        n += 2 * (G * H * cin + G * H * H + G * H + G * H);
    }
    return n + nout * 2 * H + nout;
}

// ---- DenseSameShapeConv1d (cnn_utils.py:49-82), f16x2 long-block kernels ---------------------------------------
// Layer l of a dense stack has weight (U, cin0 + l * U, 5).  Its input channels [c_off, c_off + csub) form one part of the
// contraction (the stack inputs, or the output of one earlier layer); a part is packed exactly like a Conv1d weight.
void pack_conv_part_h(const float* W, int U, int cin_total, int c_off, int csub, int cin_pad, int nslab, int CT, float scale, uint16_t* dst) {
    for (int sl = 0; sl < nslab; ++sl)
        for (int ct = 0; ct < CT; ++ct)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 8; ++j) {
                    const int co = ct * 16 + (lane & 15), k = 32 * sl + 8 * (lane >> 4) + j;
                    const int tap = k / cin_pad, ci = k % cin_pad;
                    float w = 0.0f;
                    if (co < U && tap < 5 && ci < csub) w = W[((size_t)co * cin_total + c_off + ci) * 5 + tap] * scale;
                    const uint16_t hi = f2h(w), lo = f2h(w - h2f(hi));
                    const size_t base = ((size_t)(sl * CT + ct) * 2) * 512 + (size_t)lane * 8 + j;
                    dst[base] = hi;
                    dst[base + 512] = lo;
                }
}

size_t dense_stack_bytes(const LayoutH& lo, int n_layer) {
    size_t b = 0;
    for (int l = 0; l < n_layer; ++l) b += (size_t)lo.l0b + (size_t)l * lo.midb + lo.tailb;
    return b + (size_t)8 * lo.CP * 4 + 32;
}

// canonical dense stack -> per layer: input-part fragments | l x panel-part fragments | bias * 2^S | 2^-S; then the Linear head
size_t pack_stack_h_dense(const float* src, const LayoutH& lo, int n_layer, int cin0, int nout, char* dst, std::vector<TailRef>* tails = nullptr,
                          uint32_t base_off = 0) {
    const float* s = src;
    char* d = dst;
    for (int l = 0; l < n_layer; ++l) {
        const int cin = cin0 + l * lo.U;
        const size_t nw = (size_t)lo.U * cin * 5;
        const float scale = pow2_scale(max_abs(s, nw)), inv = 1.0f / scale;
        int S = 0;
        (void)frexpf(scale, &S);
        S -= 1;                                    // scale = 2^S
        pack_conv_part_h(s, lo.U, cin, 0, cin0, 8, 2, lo.CT, scale, reinterpret_cast<uint16_t*>(d));
        d += lo.l0b;
        for (int k = 0; k < l; ++k) {
            pack_conv_part_h(s, lo.U, cin, cin0 + k * lo.U, lo.U, lo.U, lo.nsl_mid, lo.CT, scale, reinterpret_cast<uint16_t*>(d));
            d += lo.midb;
        }
        const float* b = s + nw;
        float* t = reinterpret_cast<float*>(d);
        for (int c = 0; c < lo.CP; ++c) t[c] = c < lo.U ? b[c] * scale : 0.0f;
        t[lo.CP] = inv; t[lo.CP + 1] = 1.0f; t[lo.CP + 2] = 0.0f; t[lo.CP + 3] = 65504.0f;
        if (tails) {
            TailRef r;
            r.off = base_off + (uint32_t)(d - dst);
            r.S = S;
            r.bias_s.assign(t, t + lo.CP);
            tails->push_back(std::move(r));
        }
        d += lo.tailb;
        s = b + lo.U;
    }
    float* t = reinterpret_cast<float*>(d);
    for (int f = 0; f < 8; ++f)
        for (int c = 0; c < lo.CP; ++c) t[f * lo.CP + c] = (f < nout && c < lo.U) ? s[(size_t)f * lo.U + c] : 0.0f;
    t += 8 * lo.CP;
    s += (size_t)nout * lo.U;
    for (int f = 0; f < 8; ++f) t[f] = f < nout ? s[f] : 0.0f;
    s += nout;
    return (size_t)(s - src);
}

// ---- GRU decoder packing (kernel-side layout: turboae_gru.hip) ------------------------------------------
// Gate rows of one direction in 19 MFMA row tiles: tile 3*ut + g = gate g (r, z, n) of units 16*ut + m;
// remainder tile 18, row 4*qq + i = gate i of unit 96 + qq.  `slot3` says what the remainder's 4th row holds:
// nothing (-1) or the n gate again (layer-0 input projection), in which case row i = 2 is empty instead.
// (fragment geometry constants kGH, kGRT, kGRecF, ... : turboae_host.hpp - the launch code indexes the packed buffers with them)

inline int gru_row(int T, int m, bool n_in_slot3) {
    if (T < 18) return (T % 3) * kGH + 16 * (T / 3) + m;
    const int qq = m >> 2, i = m & 3;
    if (n_in_slot3) return i < 2 ? i * kGH + 96 + qq : (i == 3 ? 2 * kGH + 96 + qq : -1);
    return i < 3 ? i * kGH + 96 + qq : -1;
}
// unit contracted by lane group kq in k-step s of the recurrent product
inline int gru_kunit(int s, int kq) { return s < 24 ? 16 * (s / 4) + 4 * kq + (s % 4) : 96 + kq; }

// W_hh (3H,H) -> [tile][k-step pair][lane][2]
void pack_gru_rec(const float* Whh, float* dst) {
    for (int T = 0; T < kGRT; ++T)
        for (int kp = 0; kp < kGKP; ++kp)
            for (int lane = 0; lane < 64; ++lane)
                for (int st = 0; st < 2; ++st) {
                    const int s = 2 * kp + st, row = gru_row(T, lane & 15, false);
                    dst[(((size_t)T * kGKP + kp) * 64 + lane) * 2 + st] = (s <= 24 && row >= 0) ? Whh[(size_t)row * kGH + gru_kunit(s, lane >> 4)] : 0.0f;
                }
}
// layer-0 W_ih (3H,cin) -> [tile][lane][2]: k-step 0 = panel columns 0..3, k-step 1 = columns 4..7
void pack_gru_x(const float* Wih, int cin, float* dst) {
    for (int T = 0; T < kGRT; ++T)
        for (int lane = 0; lane < 64; ++lane)
            for (int st = 0; st < 2; ++st) {
                const int k = 4 * st + (lane >> 4), row = gru_row(T, lane & 15, true);
                dst[((size_t)T * 64 + lane) * 2 + st] = (row >= 0 && k < cin) ? Wih[(size_t)row * cin + k] : 0.0f;
            }
}
// layer-0 accumulator-init rows: 19 tiles (r, z: b_ih + b_hh; n: b_hn; remainder (r, z, b_hn, b_in)) + 6 n-input tiles (b_in)
void pack_gru_bias0(const float* bih, const float* bhh, float* dst) {
    for (int T = 0; T < kGRT; ++T)
        for (int m = 0; m < 16; ++m) {
            float v;
            if (T < 18) { const int r = gru_row(T, m, false); v = (T % 3 < 2) ? bih[r] + bhh[r] : bhh[r]; }
            else { const int qq = m >> 2, i = m & 3, u = 96 + qq;
                   v = i < 2 ? bih[i * kGH + u] + bhh[i * kGH + u] : (i == 2 ? bhh[2 * kGH + u] : bih[2 * kGH + u]); }
            dst[T * 16 + m] = v;
        }
    for (int ut = 0; ut < 6; ++ut)
        for (int m = 0; m < 16; ++m) dst[(kGRT + ut) * 16 + m] = bih[2 * kGH + 16 * ut + m];
}
// layer-1 b_hn rows: 6 unit tiles + remainder (0, 0, b_hn, 0)
void pack_gru_bias1(const float* bhh, float* dst) {
    for (int ut = 0; ut < 6; ++ut)
        for (int m = 0; m < 16; ++m) dst[ut * 16 + m] = bhh[2 * kGH + 16 * ut + m];
    for (int m = 0; m < 16; ++m) dst[6 * 16 + m] = (m & 3) == 2 ? bhh[2 * kGH + 96 + (m >> 2)] : 0.0f;
}
// layer-1 W_ih (3H,2H) of one direction -> conv_accumulate's chunk-major pair layout over 19 tiles (see pack_conv)
void pack_gru_proj(const float* Wih, float* dst) {
    for (int c = 0; c < 25; ++c)
        for (int ct = 0; ct < kGRT; ++ct)
            for (int lane = 0; lane < 64; ++lane)
                for (int st = 0; st < 2; ++st) {
                    size_t idx = (size_t)c * kGRT * 128;
                    if (ct < 18) idx += (size_t)(ct / 2) * 256 + lane * 4 + (ct % 2) * 2 + st;
                    else idx += (size_t)9 * 256 + lane * 2 + st;
                    const int row = gru_row(ct, lane & 15, false);
                    dst[idx] = row >= 0 ? Wih[(size_t)row * 2 * kGH + 8 * c + 2 * (lane >> 4) + st] : 0.0f;
                }
}
// projection bias rows: b_ih, plus b_hh for the r and z gates (b_hn stays inside r * (...))
void pack_gru_pbias(const float* bih, const float* bhh, float* dst) {
    for (int T = 0; T < kGRT; ++T)
        for (int m = 0; m < 16; ++m) {
            const int r = gru_row(T, m, false);
            dst[T * 16 + m] = r < 0 ? 0.0f : (r < 2 * kGH ? bih[r] + bhh[r] : bih[r]);
        }
}

size_t rnn_packed_stack_floats(size_t nout) { return 2 * kGL0Dir + kGProjF + kGPB + 2 * kGL1Dir + (nout * 2 * kGH + nout + 3) / 4 * 4; }

// canonical GRU decoder -> per stack: L0 {dir: REC | XF | BIAS0} | L1 {PROJ (2 dirs) | PBIAS (2 dirs) | dir: REC | BIAS1} | Linear w | b
void repack_rnn(const float* src, float* dst, size_t H, size_t cin0, const std::vector<size_t>& nouts) {
    for (size_t s = 0; s < nouts.size(); ++s) {
        const size_t nout = nouts[s];
        const size_t cin1 = 2 * H;
        const size_t per0 = 3 * H * cin0 + 3 * H * H + 6 * H, per1 = 3 * H * cin1 + 3 * H * H + 6 * H;
        for (int d = 0; d < 2; ++d) {
            const float* p = src + d * per0;           // weight_ih | weight_hh | bias_ih | bias_hh
            float* o = dst + d * kGL0Dir;
            pack_gru_rec(p + 3 * H * cin0, o);
            pack_gru_x(p, (int)cin0, o + kGRecF);
            pack_gru_bias0(p + 3 * H * cin0 + 3 * H * H, p + 3 * H * cin0 + 3 * H * H + 3 * H, o + kGRecF + kGXF);
        }
        src += 2 * per0;
        dst += 2 * kGL0Dir;
        for (int d = 0; d < 2; ++d) {
            const float* p = src + d * per1;
            pack_gru_proj(p, dst + d * (kGProjF / 2));
            pack_gru_pbias(p + 3 * H * cin1 + 3 * H * H, p + 3 * H * cin1 + 3 * H * H + 3 * H, dst + kGProjF + d * (kGPB / 2));
            float* o = dst + kGProjF + kGPB + d * kGL1Dir;
            pack_gru_rec(p + 3 * H * cin1, o);
            pack_gru_bias1(p + 3 * H * cin1 + 3 * H * H + 3 * H, o + kGRecF);
        }
        src += 2 * per1;
        dst += kGProjF + kGPB + 2 * kGL1Dir;
        memcpy(dst, src, (nout * 2 * H + nout) * sizeof(float));
        src += nout * 2 * H + nout;
        dst += (nout * 2 * H + nout + 3) / 4 * 4;        // keep every stack 16-byte aligned
    }
}

size_t rnn_packed_floats(const std::vector<size_t>& nouts) {
    size_t n = 0;
    for (size_t nout : nouts) n += rnn_packed_stack_floats(nout);
    return n;
}
// Linear output widths of the GRU stacks: decoder = F for every half-iteration except the last (1); encoder = 1, 1, 1
std::vector<size_t> dec_rnn_nouts(size_t F, int n_iter) {
    std::vector<size_t> v((size_t)2 * n_iter, F);
    v.back() = 1;
    return v;
}

// ---- GRU decoder, f16x2 representation (turboae_gru_h2.hip) -----------------------------------------------
// (kGHTileB, kGHRec0B, kGHProjB, ... : turboae_host.hpp)

inline void put_split(char* dst, size_t hi_off, size_t lo_off, float w) {
    const uint16_t hi = f2h(w), lo = f2h(w - h2f(hi));
    memcpy(dst + hi_off, &hi, 2);
    memcpy(dst + lo_off, &lo, 2);
}
// W_hh (3H,H) [+ layer-0 W_ih (3H,cin)] -> per gate tile: 3 slabs x (hi | lo) x [lane][8 halves], then the K = 16 remainder
// [lane][4 hi halves | 4 lo halves]: k0 = unit 96 + kq, k1..3 = stack inputs 3kq .. 3kq+2 (layer 0)
void pack_gru_rec_h(const float* Whh, const float* Wih0, int cin, float scale, char* dst) {
    for (int T = 0; T < kGRT; ++T)
        for (int lane = 0; lane < 64; ++lane) {
            const int m = lane & 15, kq = lane >> 4;
            const int rowh = gru_row(T, m, false) >= 0 && !(Wih0 && T == 18 && (m & 3) == 3) ? gru_row(T, m, false) : -1;
            for (int sl = 0; sl < 3; ++sl)
                for (int j = 0; j < 8; ++j) {
                    const int unit = j < 4 ? 16 * (2 * sl) + 4 * kq + j : 16 * (2 * sl + 1) + 4 * kq + (j - 4);
                    const float w = rowh >= 0 ? Whh[(size_t)rowh * kGH + unit] * scale : 0.0f;
                    const size_t o = (size_t)T * kGHTileB + sl * 2048 + lane * 16 + j * 2;
                    put_split(dst, o, o + 1024, w);
                }
            for (int j = 0; j < 4; ++j) {
                float w = 0.0f;
                if (j == 0) w = rowh >= 0 ? Whh[(size_t)rowh * kGH + 96 + kq] * scale : 0.0f;
                else if (Wih0) {
                    const int xi = 3 * kq + j - 1;
                    int rowx;
                    if (T < 18) rowx = (T % 3 < 2) ? (T % 3) * kGH + 16 * (T / 3) + m : -1;      // the n-gate input part has its own tiles
                    else { const int qq = m >> 2, i = m & 3; rowx = i < 2 ? i * kGH + 96 + qq : (i == 3 ? 2 * kGH + 96 + qq : -1); }
                    if (rowx >= 0 && xi < cin) w = Wih0[(size_t)rowx * cin + xi] * scale;
                }
                const size_t o = (size_t)T * kGHTileB + 6144 + lane * 16 + j * 2;      // [lane][4 hi halves | 4 lo halves]
                put_split(dst, o, o + 8, w);
            }
        }
}
// Linear head (nout, 2H): direction d's half as one more row tile in the recurrence's fragment order (rows >= nout zero), so the
// layer-1 recurrence contracts h_t with it on the spot: 3 slabs x (hi | lo) x [lane][8 halves] + the K = 16 remainder (k0 = unit 96 + kq)
void pack_gru_head_h(const float* Wlin, int nout, int d, float scale, char* dst) {
    for (int lane = 0; lane < 64; ++lane) {
        const int m = lane & 15, kq = lane >> 4;
        for (int sl = 0; sl < 3; ++sl)
            for (int j = 0; j < 8; ++j) {
                const int unit = j < 4 ? 16 * (2 * sl) + 4 * kq + j : 16 * (2 * sl + 1) + 4 * kq + (j - 4);
                const float w = m < nout ? Wlin[(size_t)m * 2 * kGH + d * kGH + unit] * scale : 0.0f;
                const size_t o = (size_t)sl * 2048 + lane * 16 + j * 2;
                put_split(dst, o, o + 1024, w);
            }
        for (int j = 0; j < 4; ++j) {
            const float w = (j == 0 && m < nout) ? Wlin[(size_t)m * 2 * kGH + d * kGH + 96 + kq] * scale : 0.0f;
            const size_t o = 6144 + lane * 16 + j * 2;
            put_split(dst, o, o + 8, w);
        }
    }
}
// layer 0: n-gate input tiles (K = 16 fragments, x only)
void pack_gru_ni_h(const float* Wih0, int cin, float scale, char* dst) {
    for (int ut = 0; ut < 6; ++ut)
        for (int lane = 0; lane < 64; ++lane)
            for (int j = 0; j < 4; ++j) {
                const int m = lane & 15, kq = lane >> 4, xi = 3 * kq + j - 1;
                const float w = (j >= 1 && xi < cin) ? Wih0[(size_t)(2 * kGH + 16 * ut + m) * cin + xi] * scale : 0.0f;
                const size_t o = (size_t)ut * 1024 + lane * 16 + j * 2;
                put_split(dst, o, o + 8, w);
            }
}
// layer-1 W_ih (3H,2H) of one direction -> [slab 7][gate tile 19][hi | lo][lane][8 halves]
void pack_gru_proj_h(const float* Wih, float scale, char* dst) {
    for (int sl = 0; sl < 7; ++sl)
        for (int ct = 0; ct < kGRT; ++ct)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 8; ++j) {
                    const int row = gru_row(ct, lane & 15, false), k = 32 * sl + 8 * (lane >> 4) + j;
                    const float w = (row >= 0 && k < 2 * kGH) ? Wih[(size_t)row * 2 * kGH + k] * scale : 0.0f;
                    const size_t o = ((size_t)(sl * kGRT + ct) * 2) * 1024 + lane * 16 + j * 2;
                    put_split(dst, o, o + 1024, w);
                }
}

// One direction of layer 1 for the fused kernel (turboae_gru_l1f.hip, tae::GruL1fLayout): W_ih1 and W_hh share ONE power-of-two
// scale (their products meet in the same accumulators), the head tile has its own.
void pack_gru_l1f_dir(const float* Wih, const float* Whh, const float* bih, const float* bhh, const float* Wlin, int nout, int d,
                      float scale, float scale_h, char* dst) {
    using Lay = tae::GruL1fLayout;
    const int H = kGH;
    memset(dst, 0, Lay::kDirB);
    auto hh_unit = [](int sl, int kq, int j) { return j < 4 ? 16 * (2 * sl) + 4 * kq + j : 16 * (2 * sl + 1) + 4 * kq + (j - 4); };
    // 32-k slab fragment pair: [lane][8 halves] hi at `hi`, lo at `lo`; w(m, kq, j) = weight of row slot m, k slot (kq, j)
    auto slab = [&](size_t hi, size_t lo, auto&& w) {
        for (int lane = 0; lane < 64; ++lane)
            for (int j = 0; j < 8; ++j) put_split(dst, hi + lane * 16 + j * 2, lo + lane * 16 + j * 2, w(lane & 15, lane >> 4, j));
    };
    // K = 8 remainder fragment of the fused kernel (late r06: the three products of the slab in ONE MFMA): [lane][hi k0 k1 | hi k0 k1 | lo k0 k1 | 0 0]
    auto rem = [&](size_t off, auto&& w) {
        for (int lane = 0; lane < 64; ++lane)
            for (int j = 0; j < 2; ++j) {
                const float v = w(lane & 15, lane >> 4, j);
                put_split(dst, off + lane * 16 + j * 2, off + lane * 16 + 8 + j * 2, v);
                put_split(dst, off + lane * 16 + 4 + j * 2, off + lane * 16 + 8 + j * 2, v);
            }
    };
    const size_t lo01 = (size_t)6 * Lay::kUnitB + Lay::kRemB, lds0 = lo01 + Lay::kLo01B;
    // W_ih1 lo fragment of tile (ut, g), k-slab sl: slabs 0, 1 in the unit waves' register image, 2..5 in the LDS image
    auto ih_lo_off = [&](int ut, int g, int sl) {
        return sl < 2 ? lo01 + ((size_t)(ut * 3 + g) * 2 + sl) * 1024 : lds0 + ((size_t)(ut * 3 + g) * 4 + (sl - 2)) * 1024;
    };
    for (int ut = 0; ut < 6; ++ut) {
        const size_t ub = (size_t)ut * Lay::kUnitB;
        for (int g = 0; g < 3; ++g) {
            auto row = [&](int m) { return (size_t)(g * H + 16 * ut + m); };
            for (int sl = 0; sl < 3; ++sl)
                slab(ub + (g * 7 + 2 * sl) * 1024, ub + (g * 7 + 2 * sl + 1) * 1024,
                     [&](int m, int kq, int j) { return Whh[row(m) * H + hh_unit(sl, kq, j)] * scale; });
            rem(ub + (g * 7 + 6) * 1024, [&](int m, int kq, int j) { return j == 0 ? Whh[row(m) * H + 96 + kq] * scale : 0.0f; });
            for (int sl = 0; sl < 6; ++sl)
                slab(ub + (21 + g * 7 + sl) * 1024, ih_lo_off(ut, g, sl),
                     [&](int m, int kq, int j) { return Wih[row(m) * 2 * H + 32 * sl + 8 * kq + j] * scale; });
            rem(ub + (21 + g * 7 + 6) * 1024, [&](int m, int kq, int j) { return Wih[row(m) * 2 * H + 192 + 2 * kq + j] * scale; });
        }
    }
    {   // remainder wave: mixed tile, row 4 qq + i = (r, z, n_h, n_i) of unit 96 + qq; then the head tile
        const size_t rb = (size_t)6 * Lay::kUnitB;
        auto rowh = [&](int m) { const int qq = m >> 2, i = m & 3; return i < 3 ? i * H + 96 + qq : -1; };
        auto rowi = [&](int m) { const int qq = m >> 2, i = m & 3; return i < 2 ? i * H + 96 + qq : (i == 3 ? 2 * H + 96 + qq : -1); };
        for (int sl = 0; sl < 3; ++sl)
            slab(rb + (2 * sl) * 1024, rb + (2 * sl + 1) * 1024,
                 [&](int m, int kq, int j) { return rowh(m) >= 0 ? Whh[(size_t)rowh(m) * H + hh_unit(sl, kq, j)] * scale : 0.0f; });
        rem(rb + 6 * 1024, [&](int m, int kq, int j) { return (j == 0 && rowh(m) >= 0) ? Whh[(size_t)rowh(m) * H + 96 + kq] * scale : 0.0f; });
        for (int sl = 0; sl < 6; ++sl)
            slab(rb + (7 + 2 * sl) * 1024, rb + (8 + 2 * sl) * 1024,
                 [&](int m, int kq, int j) { return rowi(m) >= 0 ? Wih[(size_t)rowi(m) * 2 * H + 32 * sl + 8 * kq + j] * scale : 0.0f; });
        rem(rb + 19 * 1024, [&](int m, int kq, int j) { return rowi(m) >= 0 ? Wih[(size_t)rowi(m) * 2 * H + 192 + 2 * kq + j] * scale : 0.0f; });
        for (int sl = 0; sl < 3; ++sl)
            slab(rb + (20 + 2 * sl) * 1024, rb + (21 + 2 * sl) * 1024,
                 [&](int m, int kq, int j) { return m < nout ? Wlin[(size_t)m * 2 * H + d * H + hh_unit(sl, kq, j)] * scale_h : 0.0f; });
        rem(rb + 26 * 1024, [&](int m, int kq, int j) { return (j == 0 && m < nout) ? Wlin[(size_t)m * 2 * H + d * H + 96 + kq] * scale_h : 0.0f; });
    }
    float* b = reinterpret_cast<float*>(dst + lds0 + (size_t)6 * 3 * 4 * 1024);
    for (int ut = 0; ut < 6; ++ut)
        for (int m = 0; m < 16; ++m) {
            const int u = 16 * ut + m;
            b[(ut * 4 + 0) * 16 + m] = (bih[u] + bhh[u]) * scale;
            b[(ut * 4 + 1) * 16 + m] = (bih[H + u] + bhh[H + u]) * scale;
            b[(ut * 4 + 2) * 16 + m] = bih[2 * H + u] * scale;
            b[(ut * 4 + 3) * 16 + m] = bhh[2 * H + u] * scale;
        }
    for (int m = 0; m < 16; ++m) {
        const int u = 96 + (m >> 2), i = m & 3;
        b[6 * 64 + m] = (i < 2 ? bih[i * H + u] + bhh[i * H + u] : (i == 2 ? bhh[2 * H + u] : bih[2 * H + u])) * scale;
    }
    b[6 * 64 + 16] = 1.0f / scale;
    b[6 * 64 + 17] = 1.0f / scale_h;
}

// ---- LSTM / vanilla-RNN decoder stacks, unit-split f16x2 layouts (turboae_rnn_u.hip, tae::RnnULayout) ---------------------------------
// One direction of one layer.  Gate rows: tile (ut, g) row m = gate g of unit 16 ut + m; remainder tile row 4 qq + g = gate g of unit
// 96 + qq.  `Wx` (layer 0: W_ih (G H, cin), cin <= 8) rides as one K = 16 slab; `Wlin` (layer 1) gives the head tile of direction d.
void pack_rnn_u_dir(int G, const float* Wx, int cin, const float* Whh, const float* bih, const float* bhh, const float* Wlin, int nout,
                    int d, float scale, float scale_h, char* dst) {
    const int H = kGH;
    const size_t dirb = tae::RnnULayout::dir_bytes(G);
    memset(dst, 0, dirb);
    auto hh_unit = [](int sl, int kq, int j) { return j < 4 ? 16 * (2 * sl) + 4 * kq + j : 16 * (2 * sl + 1) + 4 * kq + (j - 4); };
    auto slab = [&](size_t hi, size_t lo, auto&& w) {
        for (int lane = 0; lane < 64; ++lane)
            for (int j = 0; j < 8; ++j) put_split(dst, hi + lane * 16 + j * 2, lo + lane * 16 + j * 2, w(lane & 15, lane >> 4, j));
    };
    // K = 8 fragment (the three products of the slab in ONE MFMA since late r06, turboae_rnn_u.hip::mma_gr): [lane][hi k0 k1 | hi k0 k1 | lo k0 k1 | 0 0]
    auto rem = [&](size_t off, auto&& w) {
        for (int lane = 0; lane < 64; ++lane)
            for (int j = 0; j < 2; ++j) {
                const float v = w(lane & 15, lane >> 4, j);
                put_split(dst, off + lane * 16 + j * 2, off + lane * 16 + 8 + j * 2, v);
                put_split(dst, off + lane * 16 + 4 + j * 2, off + lane * 16 + 8 + j * 2, v);
            }
    };
    auto tile = [&](size_t base, auto&& row) {       // 8 fragments of one gate tile: W_hh slabs, remainder, input slab
        for (int sl = 0; sl < 3; ++sl)
            slab(base + (2 * sl) * 1024, base + (2 * sl + 1) * 1024,
                 [&](int m, int kq, int j) { return row(m) >= 0 ? Whh[(size_t)row(m) * H + hh_unit(sl, kq, j)] * scale : 0.0f; });
        rem(base + 6 * 1024, [&](int m, int kq, int j) { return (j == 0 && row(m) >= 0) ? Whh[(size_t)row(m) * H + 96 + kq] * scale : 0.0f; });
        if (Wx) rem(base + 7 * 1024, [&](int m, int kq, int j) { return (2 * kq + j < cin && row(m) >= 0) ? Wx[(size_t)row(m) * cin + 2 * kq + j] * scale : 0.0f; });
    };
    for (int ut = 0; ut < 6; ++ut)
        for (int g = 0; g < G; ++g) tile(((size_t)ut * G + g) * 8 * 1024, [&](int m) { return g * H + 16 * ut + m; });
    const size_t rb = (size_t)6 * G * 8 * 1024;
    auto rrow = [&](int m) { const int qq = m >> 2, g = m & 3; return g < G ? g * H + 96 + qq : -1; };
    tile(rb, rrow);
    if (Wlin) {
        for (int sl = 0; sl < 3; ++sl)
            slab(rb + (8 + 2 * sl) * 1024, rb + (9 + 2 * sl) * 1024,
                 [&](int m, int kq, int j) { return m < nout ? Wlin[(size_t)m * 2 * H + d * H + hh_unit(sl, kq, j)] * scale_h : 0.0f; });
        rem(rb + 14 * 1024, [&](int m, int kq, int j) { return (j == 0 && m < nout) ? Wlin[(size_t)m * 2 * H + d * H + 96 + kq] * scale_h : 0.0f; });
    }
    float* b = reinterpret_cast<float*>(dst + rb + 15 * 1024);
    if (Wx) {      // layer 0: accumulators start at b_ih + b_hh (layer 1: at GI, which carries both)
        for (int ut = 0; ut < 6; ++ut)
            for (int g = 0; g < G; ++g)
                for (int m = 0; m < 16; ++m) b[(ut * G + g) * 16 + m] = (bih[g * H + 16 * ut + m] + bhh[g * H + 16 * ut + m]) * scale;
        for (int m = 0; m < 16; ++m) b[6 * G * 16 + m] = rrow(m) >= 0 ? (bih[rrow(m)] + bhh[rrow(m)]) * scale : 0.0f;
    }
    b[6 * G * 16 + 16] = 1.0f / scale;
    b[6 * G * 16 + 17] = 1.0f / scale_h;
}
// layer-1 W_ih (G H, 2H) of both directions -> [dir][slab 7][tile][hi | lo][lane][8 halves] | bias rows (b_ih + b_hh) | 2^-S
void pack_rnn_u_proj(int G, const float* l1, size_t per1, char* dst) {
    const int H = kGH, CTT = 6 * G + 1;
    const size_t cin1 = 2 * H, dirb = (size_t)7 * CTT * 2048;
    const float scale = pow2_scale(fmaxf(max_abs(l1, (size_t)G * H * cin1), max_abs(l1 + per1, (size_t)G * H * cin1)));
    float* pb = reinterpret_cast<float*>(dst + 2 * dirb);
    for (int d = 0; d < 2; ++d) {
        const float* Wih = l1 + d * per1;
        const float* bih = Wih + (size_t)G * H * cin1 + (size_t)G * H * H;
        const float* bhh = bih + G * H;
        for (int T = 0; T < CTT; ++T)
            for (int m = 0; m < 16; ++m) {
                const int row = T < 6 * G ? (T % G) * H + 16 * (T / G) + m : ((m & 3) < G ? (m & 3) * H + 96 + (m >> 2) : -1);
                pb[(d * CTT + T) * 16 + m] = row >= 0 ? (bih[row] + bhh[row]) * scale : 0.0f;
                for (int sl = 0; sl < 7; ++sl)
                    for (int kq = 0; kq < 4; ++kq)
                        for (int j = 0; j < 8; ++j) {
                            const int lane = kq * 16 + m;
                            const size_t o = d * dirb + ((size_t)(sl * CTT + T) * 2) * 1024 + lane * 16 + j * 2;
                            if (sl == 6) {
                                // the K = 8 tail (k = 192..199) as ONE MFMA (late r06, turboae_rnn_u.hip::lds_tail8): where the slab's hi fragment goes,
                                // [hi k0 k1 | hi k0 k1 | lo k0 k1 | 0 0] with k = 192 + 2 kq + j; its lo fragment stays zero and is never fetched
                                if (j >= 2) continue;
                                const int k = 192 + 2 * kq + j;
                                const float w = row >= 0 ? Wih[(size_t)row * cin1 + k] * scale : 0.0f;
                                put_split(dst, o, o + 8, w);           // hi -> slots 0, 1; lo -> slots 4, 5
                                put_split(dst, o + 4, o + 8, w);       // hi -> slots 2, 3
                                continue;
                            }
                            const int k = 32 * sl + 8 * kq + j;
                            const float w = (row >= 0 && k < 2 * H) ? Wih[(size_t)row * cin1 + k] * scale : 0.0f;
                            put_split(dst, o, o + 1024, w);
                        }
            }
    }
    pb[2 * CTT * 16] = 1.0f / scale;
}
size_t rnn_u_stack_bytes(size_t nout, int G) {
    return 4 * tae::RnnULayout::dir_bytes(G) + tae::RnnULayout::proj_bytes(G) + ((nout * 2 * kGH + nout) * 4 + 15) / 16 * 16;
}
// canonical LSTM / RNN decoder -> per stack: L0 {dir image} x 2 | PROJ | L1 {dir image} x 2 | Linear w | b; gimul[2 s + d] = 2^S of L1 dir d
void repack_rnn_u(const float* src, char* dst, size_t cin0, const std::vector<size_t>& nouts, int G, std::vector<float>& gimul) {
    const size_t H = kGH, cin1 = 2 * H, GH = (size_t)G * H;
    const size_t per0 = GH * cin0 + GH * H + 2 * GH, per1 = GH * cin1 + GH * H + 2 * GH, dirb = tae::RnnULayout::dir_bytes(G);
    gimul.assign(2 * nouts.size(), 1.0f);
    for (size_t s = 0; s < nouts.size(); ++s) {
        const size_t nout = nouts[s];
        for (int d = 0; d < 2; ++d) {
            const float* p = src + d * per0;           // weight_ih | weight_hh | bias_ih | bias_hh
            const float scale = pow2_scale(fmaxf(max_abs(p, GH * cin0), max_abs(p + GH * cin0, GH * H)));
            pack_rnn_u_dir(G, p, (int)cin0, p + GH * cin0, p + GH * cin0 + GH * H, p + GH * cin0 + GH * H + GH, nullptr, 0, d, scale, 1.0f, dst + d * dirb);
        }
        src += 2 * per0;
        dst += 2 * dirb;
        const float* l1 = src;
        pack_rnn_u_proj(G, l1, per1, dst);
        dst += tae::RnnULayout::proj_bytes(G);
        const float* wlin = l1 + 2 * per1;
        const float scale_h = pow2_scale(max_abs(wlin, nout * 2 * H));
        for (int d = 0; d < 2; ++d) {
            const float* p = l1 + d * per1;
            const float scale = pow2_scale(max_abs(p + GH * cin1, GH * H));
            gimul[2 * s + d] = scale;
            pack_rnn_u_dir(G, nullptr, 0, p + GH * cin1, nullptr, nullptr, wlin, (int)nout, d, scale, scale_h, dst + d * dirb);
        }
        dst += 2 * dirb;
        src += 2 * per1;
        memcpy(dst, src, (nout * 2 * H + nout) * sizeof(float));
        src += nout * 2 * H + nout;
        dst += ((nout * 2 * H + nout) * 4 + 15) / 16 * 16;
    }
}

size_t rnn_h_stack_bytes(size_t nout) {
    return 2 * kGHRec0B + kGHProjB + 2 * kGHRec1B + ((nout * 2 * kGH + nout) * 4 + 15) / 16 * 16 + 2 * (size_t)tae::GruL1fLayout::kDirB;
}
size_t rnn_h_l1f_offset(size_t nout) { return rnn_h_stack_bytes(nout) - 2 * (size_t)tae::GruL1fLayout::kDirB; }
size_t rnn_h_packed_bytes(const std::vector<size_t>& nouts) {
    size_t n = 0;
    for (size_t nout : nouts) n += rnn_h_stack_bytes(nout);
    return n;
}

// canonical GRU decoder -> per stack: L0 {dir: REC | NI | BIAS0 * 2^S | 2^-S} | L1 {PROJ (2 dirs) | PBIAS * 2^Sp | 2^-Sp | dir: REC | BN1 * 2^S | 2^-S} | Linear
void repack_rnn_h(const float* src, char* dst, size_t cin0, const std::vector<size_t>& nouts) {
    const size_t H = kGH;
    for (size_t s = 0; s < nouts.size(); ++s) {
        const size_t nout = nouts[s];
        const size_t cin1 = 2 * H;
        const size_t per0 = 3 * H * cin0 + 3 * H * H + 6 * H, per1 = 3 * H * cin1 + 3 * H * H + 6 * H;
        for (int d = 0; d < 2; ++d) {
            const float* p = src + d * per0;           // weight_ih | weight_hh | bias_ih | bias_hh
            const float scale = pow2_scale(fmaxf(max_abs(p, 3 * H * cin0), max_abs(p + 3 * H * cin0, 3 * H * H)));
            char* o = dst + d * kGHRec0B;
            pack_gru_rec_h(p + 3 * H * cin0, p, (int)cin0, scale, o);
            pack_gru_ni_h(p, (int)cin0, scale, o + kGHFragB);
            float* b = reinterpret_cast<float*>(o + kGHFragB + kGHNiB);
            pack_gru_bias0(p + 3 * H * cin0 + 3 * H * H, p + 3 * H * cin0 + 3 * H * H + 3 * H, b);
            for (int i = 0; i < 25 * 16; ++i) b[i] *= scale;
            for (int i = 0; i < 4; ++i) b[25 * 16 + i] = 1.0f / scale;
        }
        src += 2 * per0;
        dst += 2 * kGHRec0B;
        {
            const float scale = pow2_scale(fmaxf(max_abs(src, 3 * H * cin1), max_abs(src + per1, 3 * H * cin1)));
            float* pb = reinterpret_cast<float*>(dst + 2 * kGHProjDirB);
            for (int d = 0; d < 2; ++d) {
                const float* p = src + d * per1;
                pack_gru_proj_h(p, scale, dst + d * kGHProjDirB);
                pack_gru_pbias(p + 3 * H * cin1 + 3 * H * H, p + 3 * H * cin1 + 3 * H * H + 3 * H, pb + d * (kGRT * 16));
            }
            for (int i = 0; i < 2 * kGRT * 16; ++i) pb[i] *= scale;
            for (int i = 0; i < 4; ++i) pb[2 * kGRT * 16 + i] = 1.0f / scale;
        }
        dst += kGHProjB;
        for (int d = 0; d < 2; ++d) {
            const float* p = src + d * per1;
            const float scale = pow2_scale(max_abs(p + 3 * H * cin1, 3 * H * H));
            char* o = dst + d * kGHRec1B;
            pack_gru_rec_h(p + 3 * H * cin1, nullptr, 0, scale, o);
            const float* wlin = src + 2 * per1;                    // Linear(2H -> nout) follows the two directions of layer 1
            const float scale_h = pow2_scale(max_abs(wlin, nout * 2 * H));
            pack_gru_head_h(wlin, (int)nout, d, scale_h, o + kGHFragB);
            float* b = reinterpret_cast<float*>(o + kGHFragB + kGHTileB);
            pack_gru_bias1(p + 3 * H * cin1 + 3 * H * H + 3 * H, b);
            for (int i = 0; i < 7 * 16; ++i) b[i] *= scale;
            b[7 * 16] = 1.0f / scale;
            b[7 * 16 + 1] = 1.0f / scale_h;
            b[7 * 16 + 2] = b[7 * 16 + 3] = 0.0f;
        }
        const float* l1 = src;                                     // layer 1, two directions; the Linear head follows
        src += 2 * per1;
        dst += 2 * kGHRec1B;
        memcpy(dst, src, (nout * 2 * H + nout) * sizeof(float));
        dst += ((nout * 2 * H + nout) * 4 + 15) / 16 * 16;
        const float scale_h = pow2_scale(max_abs(src, nout * 2 * H));
        for (int d = 0; d < 2; ++d) {
            const float* p = l1 + d * per1;                         // weight_ih | weight_hh | bias_ih | bias_hh
            const float scale = pow2_scale(fmaxf(max_abs(p, 3 * H * cin1), max_abs(p + 3 * H * cin1, 3 * H * H)));
            pack_gru_l1f_dir(p, p + 3 * H * cin1, p + 3 * H * cin1 + 3 * H * H, p + 3 * H * cin1 + 3 * H * H + 3 * H, src, (int)nout, d,
                             scale, scale_h, dst);
            dst += tae::GruL1fLayout::kDirB;
        }
        src += nout * 2 * H + nout;
    }
}

// Canonical weight order (turboae_amd/weights.py::canonical_entries): f(true, co, cin, ks) for a Conv1d weight (co, cin, ks),
// f(false, n, 0, 0) for a plain run of n floats (biases, Linear layers, whole GRU stacks).
template <class Fn>
void walk_weights(const tae_config* c, Fn&& f) {
    size_t U = c->enc_num_unit;
    const size_t F = c->num_iter_ft;
    for (int s = 0; s < 3; ++s) {
        if (c->enc_type == 1) { f(false, rnn_stack_floats(U, 1, 1, (size_t)cell_gates(c->enc_rnn)), 0, 0); continue; }
        for (int l = 0; l < c->enc_num_layer; ++l) {
            f(true, U, c->dense ? 1 + l * U : (l == 0 ? 1 : U), (size_t)c->enc_kernel_size);
            f(false, U, 0, 0);
        }
        f(false, U + 1, 0, 0);
    }
    U = c->dec_num_unit;
    for (int it = 0; it < c->num_iteration; ++it)
        for (int half = 0; half < 2; ++half) {
            const size_t nout = (half == 1 && it == c->num_iteration - 1) ? 1 : F;
            if (c->dec_type == 1) { f(false, rnn_stack_floats(U, 2 + F, nout, (size_t)cell_gates(c->dec_rnn)), 0, 0); continue; }
            for (int l = 0; l < c->dec_num_layer; ++l) {
                f(true, U, c->dense ? 2 + F + l * U : (l == 0 ? 2 + F : U), (size_t)c->dec_kernel_size);
                f(false, U, 0, 0);
            }
            f(false, nout * U + nout, 0, 0);
        }
}

size_t num_weights(const tae_config* c) {
    if (tae::generic_needed(c)) return tae::generic_num_weights(c);
    size_t n = 0;
    walk_weights(c, [&](bool conv, size_t a, size_t b, size_t ks) { n += conv ? a * b * ks : a; });
    return n;
}

// The conv kernels exist for 32 / 64 / 100 channels and contract >= 5 taps.  Any narrower stack is the next wider one
// with zero weights and biases for the extra channels (ELU(0) = 0: they stay zero and feed nothing), and a SameShapeConv1d
// of kernel size 1 or 3 (padding ks / 2) is the 5-tap convolution whose outer taps are zero - so such configurations
// are embedded, exactly, into the instantiated geometry before packing.  `out_cfg` receives that geometry.
int conv_kernel_width(int u, int ks);
std::vector<float> embed_weights(const tae_config* c, const float* w, tae_config* out_cfg) {
    std::vector<float> out;
    *out_cfg = *c;
    const size_t F = c->num_iter_ft;
    auto conv = [&](size_t U, size_t U2, size_t cin, bool cin_is_u, size_t ks, size_t ks2) {     // weight (U, cin, ks) + bias (U)
        const size_t cin2 = cin_is_u ? U2 : cin, off = (ks2 - ks) / 2;
        for (size_t co = 0; co < U2; ++co)
            for (size_t ci = 0; ci < cin2; ++ci)
                for (size_t j = 0; j < ks2; ++j)
                    out.push_back(co < U && ci < cin && j >= off && j < off + ks ? w[(co * cin + ci) * ks + (j - off)] : 0.0f);
        w += U * cin * ks;
        for (size_t co = 0; co < U2; ++co) out.push_back(co < U ? w[co] : 0.0f);
        w += U;
    };
    auto linear = [&](size_t nout, size_t U, size_t U2) {                                          // weight (nout, U) + bias (nout)
        for (size_t f = 0; f < nout; ++f)
            for (size_t ci = 0; ci < U2; ++ci) out.push_back(ci < U ? w[f * U + ci] : 0.0f);
        w += nout * U;
        out.insert(out.end(), w, w + nout);
        w += nout;
    };
    // 2-layer bidirectional GRU(cin0 -> H) + Linear(2H -> nout) widened to H2 = 100 units: gate rows g*H + u -> g*H2 + u, the
    // layer-1 / Linear input columns (forward | backward halves) likewise.  A unit with zero weights and biases has
    // r = z = 1/2, n = tanh(0) = 0, so h' = (1 - z) n + z h stays at its initial 0 and feeds nothing (LSTM: i = f = o = 1/2, g = 0, so
    // c and h stay 0; vanilla RNN: h = tanh(0) = 0).
    auto rnn = [&](size_t H, size_t H2, size_t cin0, size_t nout, size_t NG = 3) {
        auto col2 = [&](size_t c2) -> long { const size_t half = c2 / H2, u = c2 % H2; return u < H ? (long)(half * H + u) : -1; };
        for (int l = 0; l < 2; ++l) {
            const size_t cin = l == 0 ? cin0 : 2 * H, cin2 = l == 0 ? cin0 : 2 * H2;
            for (int d = 0; d < 2; ++d) {
                for (size_t g = 0; g < NG; ++g)                       // weight_ih (G H, cin): G = 3 gates of a GRU, 4 of an LSTM, 1 of a vanilla RNN
                    for (size_t u = 0; u < H2; ++u)
                        for (size_t c2 = 0; c2 < cin2; ++c2) {
                            const long cc = l == 0 ? (long)c2 : col2(c2);
                            out.push_back(u < H && cc >= 0 ? w[(g * H + u) * cin + (size_t)cc] : 0.0f);
                        }
                w += NG * H * cin;
                for (size_t g = 0; g < NG; ++g)                       // weight_hh (G H, H)
                    for (size_t u = 0; u < H2; ++u)
                        for (size_t c2 = 0; c2 < H2; ++c2) out.push_back(u < H && c2 < H ? w[(g * H + u) * H + c2] : 0.0f);
                w += NG * H * H;
                for (int b = 0; b < 2; ++b) {                         // bias_ih, bias_hh (G H)
                    for (size_t g = 0; g < NG; ++g)
                        for (size_t u = 0; u < H2; ++u) out.push_back(u < H ? w[g * H + u] : 0.0f);
                    w += NG * H;
                }
            }
        }
        for (size_t f = 0; f < nout; ++f)                             // Linear (nout, 2H) + bias
            for (size_t c2 = 0; c2 < 2 * H2; ++c2) { const long cc = col2(c2); out.push_back(cc >= 0 ? w[f * 2 * H + (size_t)cc] : 0.0f); }
        w += nout * 2 * H;
        out.insert(out.end(), w, w + nout);
        w += nout;
    };
    {
        const size_t U = c->enc_num_unit, U2 = c->enc_type == 1 ? 100 : (size_t)conv_kernel_width((int)U, c->enc_kernel_size);
        const size_t ks = c->enc_kernel_size, ks2 = ks < 5 ? 5 : ks;
        out_cfg->enc_num_unit = (int32_t)U2;
        out_cfg->enc_kernel_size = (int32_t)ks2;
        for (int s = 0; s < 3; ++s) {
            if (c->enc_type == 1) { rnn(U, U2, 1, 1, (size_t)cell_gates(c->enc_rnn)); continue; }
            // dense stacks (widths are exact there, check_cfg): layer l sees cat(input, out_0 .. out_{l-1}) = 1 + l * U channels
            for (int l = 0; l < c->enc_num_layer; ++l) {
                if (c->dense) conv(U, U2, 1 + l * U, false, ks, ks2);
                else conv(U, U2, l == 0 ? 1 : U, l != 0, ks, ks2);
            }
            linear(1, U, U2);
        }
    }
    {
        const size_t U = c->dec_num_unit, U2 = c->dec_type == 1 ? 100 : (size_t)conv_kernel_width((int)U, c->dec_kernel_size);
        const size_t ks = c->dec_kernel_size, ks2 = ks < 5 ? 5 : ks;
        out_cfg->dec_num_unit = (int32_t)U2;
        out_cfg->dec_kernel_size = (int32_t)ks2;
        for (int it = 0; it < c->num_iteration; ++it)
            for (int half = 0; half < 2; ++half) {
                const size_t nout = (half == 1 && it == c->num_iteration - 1) ? 1 : F;
                if (c->dec_type == 1) { rnn(U, U2, 2 + F, nout, (size_t)cell_gates(c->dec_rnn)); continue; }
                for (int l = 0; l < c->dec_num_layer; ++l) {
                    if (c->dense) conv(U, U2, 2 + F + l * U, false, ks, ks2);
                    else conv(U, U2, l == 0 ? 2 + F : U, l != 0, ks, ks2);
                }
                linear(nout, U, U2);
            }
    }
    return out;
}

// The 100-wide f16x2 conv kernels are 5-tap kernels since late r06 (GeoH<100>::TAIL20: every layer ends in a two-MFMA tail slab, a
// compile-time property): kernel sizes 7 and 9 at widths 65..100 run, exactly, embedded in the 124-wide instantiation (whose slab
// count follows the kernel size at run time); kernel sizes 1 and 3 are embedded in 5 taps as before.
int conv_kernel_width(int u, int ks) {
    const int w = kernel_width(u);
    return (w == 100 && ks > 5) ? 124 : w;
}
bool needs_embedding(const tae_config* c) {
    return c->enc_kernel_size < 5 || c->dec_kernel_size < 5 ||
           (c->enc_type == 0 ? conv_kernel_width(c->enc_num_unit, c->enc_kernel_size) : 100) != c->enc_num_unit ||
           (c->dec_type == 0 ? conv_kernel_width(c->dec_num_unit, c->dec_kernel_size) : 100) != c->dec_num_unit;
}


// ---- range calibration of the fp16-split conv kernels ---------------------------------------------------------------------
// The reference convolves in fp32 (cnn_utils.py:36-46: F.conv1d on fp32 tensors), 24 significant bits at any magnitude; an fp16
// hi/lo pair has them only inside a window (turboae_h2.hip, "range bookkeeping").  calibrate_range puts every panel into that
// window: it runs the handle's own forward on a calibration batch with the kernels collecting each layer's max |activation|
// (FusedParams::cal), picks per layer the power of two that brings the maximum to [2^10, 2^11), rewrites the packed tails (bias,
// scales, thresholds - the weight fragments are untouched) and repeats until no exponent moves (a layer whose input was out of
// the window in one pass is measured correctly in the next; 2 passes for anything sane, at most kCalMaxPass).  Everything is a
// power of two, so the calibration batch only decides where the floor and the ceiling sit, never a rounding of an in-window value.
constexpr int kCalMaxPass = 5;

// tae_config.range_calibration (0 = on), overridden by env TAE_RANGE_CAL=0|1 (testing knob: 0 reproduces the uncalibrated r03 arithmetic)
bool range_calibration_on(const tae_config* c) {
    if (const char* e = tae::debug_knob("TAE_RANGE_CAL")) {
        if (e[0] == '0') return false;
        if (e[0] == '1') return true;
    }
    return c->range_calibration == 0;
}

int upload_tails(tae_handle* h, bool decoder) {
    const std::vector<TailRef>& tails = decoder ? h->dec_tails : h->enc_tails;
    if (tails.empty()) return TAE_OK;
    const std::vector<int>& A = decoder ? h->dec_A : h->enc_A;
    const std::vector<int>& Ax = decoder ? h->dec_Ax : h->enc_Ax;
    const std::vector<float>& low = decoder ? h->dec_low : h->enc_low;
    const std::vector<float>& high = decoder ? h->dec_high : h->enc_high;
    const std::vector<int>& kind = decoder ? h->dec_kind : h->enc_kind;
    const int nl = decoder ? h->cfg.dec_num_layer : h->cfg.enc_num_layer;
    char* base = decoder ? h->d_wdec_h : h->d_wenc_h;
    std::vector<float> t;
    for (size_t i = 0; i < tails.size(); ++i) {
        const int s = (int)i / nl, l = (int)i % nl;
        const int a_in = l == 0 ? Ax[s] : A[i - 1];
        const int a_out = l + 1 < nl ? A[i] : 0;               // the last layer feeds the Linear head in fp32: no panel, no scale
        tail_values(tails[i], a_in, a_out, l + 1 < nl ? low[i] : 0.0f, high[i], kind[i], t);
        TAE_HIP(hipMemcpy(base + tails[i].off, t.data(), t.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    return TAE_OK;
}

// exponent that brings a measured maximum into the target window; `overflowed`: the pass that measured it saturated
inline int range_exponent(float m, int a_old, bool* again) {
    if (!std::isfinite(m)) { *again = true; return a_old - 8 < -80 ? -80 : a_old - 8; }     // overflowed under the old exponent: back off, measure again
    if (!(m > 0.0f)) return 0;                                     // an all-zero panel is exact under any exponent
    int e = 0;
    (void)frexpf(m, &e);                                           // m in [2^(e-1), 2^e)
    int a = kRangeTarget - e;
    return a > 80 ? 80 : (a < -80 ? -80 : a);
}

// u (B,L,1), noise as tae_forward takes it (device pointers), or nullptr: a synthetic batch - Bernoulli bits and the configured
// channel's own kind of noise at 0 dB (masks with p = 0.1 for bec / bsc).  Synchronises the device.
int calibrate_range(tae_handle* h, const float* u_user, const float* noise_user, int32_t B_user) {
    if (h->gen || h->prec != 1 || (h->enc_tails.empty() && h->dec_tails.empty())) return TAE_OK;
    const int L = h->cfg.block_len, nle = h->cfg.enc_num_layer, nld = h->cfg.dec_num_layer, n_stack = 2 * h->cfg.num_iteration;
    const bool do_dec = !h->dec_tails.empty(), do_enc = !h->enc_tails.empty();
    int32_t B = B_user;
    if (!u_user) {
        B = 76800 / L;
        if (B > 768) B = 768;
        if (B < 8) B = 8;
        if (h->cfg.dec_type == 1 && B > 64) B = 64;            // GRU decoder: only the CNN encoder is calibrated; keep its chunk workspace small
    }
    TAE_HIP(hipDeviceSynchronize());
    if (B > h->cap) { const int rc = tae_reserve(h, B); if (rc != TAE_OK) return rc; }
    if (!h->d_cal) TAE_HIP(hipMalloc(&h->d_cal, cal_words(h) * sizeof(uint32_t)));
    float *d_u = nullptr, *d_noise = nullptr, *d_x = nullptr;
    const size_t nbits = (size_t)B * L;
    const int nmult = h->nopts.channel == 3 ? 2 : 1;
    auto cleanup = [&]() { (void)hipFree(d_u); (void)hipFree(d_noise); (void)hipFree(d_x); h->calibrating = false; };
#define TAE_CAL(expr) do { hipError_t e__ = (expr); if (e__ != hipSuccess) { cleanup(); return fail(TAE_EHIP, std::string(#expr) + ": " + hipGetErrorString(e__)); } } while (0)
    TAE_CAL(hipMalloc(&d_x, nbits * sizeof(float)));
    const float* u = u_user;
    const float* noise = noise_user;
    if (!u_user) {
        TAE_CAL(hipMalloc(&d_u, nbits * sizeof(float)));
        TAE_CAL(hipMalloc(&d_noise, (size_t)nmult * nbits * 3 * sizeof(float)));
        TAE_CAL(tae::launch_gen_inputs(d_u, nullptr, nbits, 0, 0xCA11B8A7E0ull, 0, 1.0f, nullptr));
        tae_noise_opts no = default_noise_opts();
        float ts = 0.0f;
        if (h->nopts.channel == 1) { no.kind = TAE_NOISE_BEC; ts = 0.1f; }
        else if (h->nopts.channel == 2) { no.kind = TAE_NOISE_BSC; ts = 0.1f; }
        else if (h->nopts.channel == 3) no.kind = TAE_NOISE_FADING;
        tae::NoiseGen g;
        if (make_noise_gen(&no, ts, &g) != TAE_OK) { cleanup(); return TAE_EINVAL; }
        TAE_CAL(tae::launch_gen_noise(g, d_noise + (size_t)(nmult - 1) * nbits * 3, nmult == 2 ? d_noise : nullptr, (size_t)B, 0, L, 0xCA11B8A7E1ull, nullptr));
        u = d_u;
        noise = d_noise;
    }
    // start from the current exponents (first call: all 0) with the low-side checks off
    if (h->enc_A.empty()) { h->enc_A.assign(h->enc_tails.size(), 0); h->enc_Ax.assign(3, 0); }
    if (h->dec_A.empty()) { h->dec_A.assign(h->dec_tails.size(), 0); h->dec_Ax.assign(do_dec ? n_stack : 0, 0); }
    h->enc_low.assign(h->enc_tails.size(), 0.0f);
    h->dec_low.assign(h->dec_tails.size(), 0.0f);
    if (h->enc_high.empty()) { h->enc_high.assign(h->enc_tails.size(), 65504.0f); h->enc_kind.assign(h->enc_tails.size(), 0); }
    if (h->dec_high.empty()) { h->dec_high.assign(h->dec_tails.size(), 65504.0f); h->dec_kind.assign(h->dec_tails.size(), 0); }
    std::vector<uint32_t> cal(cal_words(h));
    const size_t doff = cal_dec_offset(h);
    auto word = [&](size_t i) { float f; memcpy(&f, &cal[i], 4); return f; };
    h->calibrating = true;
    h->cal_passes = 0;
    int rc = TAE_OK;
    for (int pass = 0; pass < kCalMaxPass; ++pass) {
        if ((rc = upload_tails(h, false)) != TAE_OK || (rc = upload_tails(h, true)) != TAE_OK) break;
        TAE_CAL(hipMemset(h->d_cal, 0, cal.size() * sizeof(uint32_t)));
        if (do_enc || do_dec) {
            if ((rc = run_encoder(h, u, h->d_xtx, h->d_stats, B, nullptr)) != TAE_OK) break;
            if (do_dec) {
                if ((rc = tae_normalize(h, h->d_xtx, h->d_stats, noise, nullptr, h->d_rx, B, nullptr)) != TAE_OK) break;
                if ((rc = run_decoder(h, h->d_rx, d_x, B, nullptr)) != TAE_OK) break;
            }
        }
        TAE_CAL(hipDeviceSynchronize());
        TAE_CAL(hipMemcpy(cal.data(), h->d_cal, cal.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
        ++h->cal_passes;
        bool again = false, moved = false;
        auto side = [&](bool decoder) {
            std::vector<int>& A = decoder ? h->dec_A : h->enc_A;
            std::vector<int>& Ax = decoder ? h->dec_Ax : h->enc_Ax;
            const int nl = decoder ? nld : nle, ns = decoder ? n_stack : 3;
            const size_t o = decoder ? doff : 0;
            if (A.empty()) return;
            for (int s = 0; s < ns; ++s) {
                // stack inputs: the encoder's are +-1; decoder: one exponent from the global maximum (dense stacks: one per stack)
                float mx = 1.0f;
                const float rmax = decoder ? word(o + cal_dec_r(h)) : 0.0f;
                auto both = [](float a, float b) { return !(a <= b) ? a : b; };      // max where inf wins
                if (decoder && h->nbd >= 1) mx = both(word(o), rmax);
                else if (decoder && h->cfg.dense) mx = both(word(o + 1 + (size_t)ns * nl + s), rmax);
                else if (decoder) {
                    // plain long-block decoder: the same ONE exponent the whole-block kernel would use (the union of what the stacks
                    // stage is what that kernel's planes see), so both paths stay bit-identical on blocks either can run - for networks whose
                    // last layers stay at or above 1/4 (the long-block kernels always use exp2 - 1 in their heads; the whole-block
                    // kernels switch to both expm1 branches below that, FusedParams::head2, and tae_decode_taps always does)
                    mx = rmax;
                    for (int k = 0; k < ns; ++k) mx = both(word(o + 1 + (size_t)ns * nl + k), mx);
                }
                int ax = decoder ? range_exponent(mx, Ax[s], &again) : 0;
                if (h->cfg.dense) {
                    // dense stacks contract the inputs and every earlier panel into one accumulator: one exponent per stack
                    float m = mx;
                    bool inf = !std::isfinite(mx);
                    for (int l = 0; l + 1 < nl; ++l) { const float v = word(o + 1 + (size_t)s * nl + l); if (!std::isfinite(v)) inf = true; else if (v > m) m = v; }
                    ax = range_exponent(inf ? INFINITY : m, Ax[s], &again);
                    for (int l = 0; l < nl; ++l) { if (A[(size_t)s * nl + l] != ax) moved = true; A[(size_t)s * nl + l] = ax; }
                } else {
                    for (int l = 0; l < nl; ++l) {
                        const size_t i = (size_t)s * nl + l;
                        const float m = word(o + 1 + i);
                        const bool panel = l + 1 < nl;       // the last layer's ELU output feeds the Linear head in fp32: no panel, exponent 0
                        const int a = panel ? range_exponent(m, A[i], &again) : 0;
                        if (a != A[i]) moved = true;
                        A[i] = a;
                        // ELU branch of the layer (turboae_h2.hip, TAE_ELU_MODE 2): all |x| <= 2^-5 -> polynomial, valid to |x| = 2^-3;
                        // maximum below 1/4 -> both expm1 branches per value; else exp2 - 1 (3e-8 absolute <= 2^-23 of the maximum)
                        const bool ok = std::isfinite(m) && m > 0.0f;
                        const int k = ok && m <= 0.03125f && panel ? 1 : (ok && m < 0.25f ? 2 : 0);     // a last layer: 0 or 2 (nothing checks a polynomial's bound there)
                        const float hi = k == 1 ? ldexpf(0.125f, a) : 65504.0f;
                        std::vector<float>& H = decoder ? h->dec_high : h->enc_high;
                        std::vector<int>& K = decoder ? h->dec_kind : h->enc_kind;
                        if (H[i] != hi || K[i] != k) moved = true;
                        H[i] = hi;
                        K[i] = k;
                    }
                }
                if (ax != Ax[s]) moved = true;
                Ax[s] = ax;
            }
        };
        side(false);
        side(true);
        if (!moved && !again) break;
    }
    if (rc == TAE_OK) {
        // low-side checks on for every panel that holds something (an all-zero panel - embedded channels, dead layers - stays unchecked)
        // ... and that is large enough for its maximum to say something about the data's scale: the check is per workgroup, and the
        // largest of a few dozen values (block_len 1, width 3) can sit 2^7 under the calibration batch's by chance
        const bool ce = h->enc_min_values >= kRangeMinValues, cd = h->dec_min_values >= kRangeMinValues;
        for (size_t i = 0; i < h->enc_tails.size(); ++i) h->enc_low[i] = ce && (int)(i % nle) + 1 < nle && word(1 + i) > 0.0f ? kRangeLow : 0.0f;
        for (size_t i = 0; i < h->dec_tails.size(); ++i) h->dec_low[i] = cd && (int)(i % nld) + 1 < nld && word(doff + 1 + i) > 0.0f ? kRangeLow : 0.0f;
        // received values: flagged when a workgroup's largest one is 2^7 under the calibration batch's largest (they share the
        // extrinsic values' exponent, which may put them well under the layer window to begin with)
        h->dec_r_low = 0.0f;
        if (do_dec && L * 3 >= 48) {
            const float rmax = word(doff + cal_dec_r(h));
            if (std::isfinite(rmax) && rmax > 0.0f) h->dec_r_low = ldexpf(rmax, -7);
        }
        if ((rc = upload_tails(h, false)) == TAE_OK) rc = upload_tails(h, true);
    }
    h->calibrated = rc == TAE_OK;
    if (hipMemset(h->d_flags, 0, sizeof(uint32_t)) != hipSuccess) rc = rc == TAE_OK ? TAE_EHIP : rc;     // whatever the calibration passes raised
    cleanup();
#undef TAE_CAL
    return rc;
}

// tae_config.range_fallback: run `call` on the fp16-split handle, wait for it, read the range word, and if a launch left the window

}  // namespace host
}  // namespace tae

using namespace tae::host;

extern "C" {

size_t tae_num_weights(const tae_config* cfg) {
    if (check_cfg(cfg) != TAE_OK) return 0;
    return num_weights(cfg);
}

int tae_create(const tae_config* cfg, const float* weights, size_t n_weights, tae_handle** out) {
    if (!out) return fail(TAE_EINVAL, "out is NULL");
    *out = nullptr;
    int rc = check_cfg(cfg);
    if (rc != TAE_OK) return rc;
    if (!weights) return fail(TAE_EINVAL, "weights is NULL");
    if (n_weights != num_weights(cfg)) {
        char buf[160];
        snprintf(buf, sizeof(buf), "weights blob has %zu floats, configuration needs %zu", n_weights, num_weights(cfg));
        return fail(TAE_EINVAL, buf);
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return fail(TAE_EHIP, "no HIP device available: libturboae_hip needs an AMD GPU (no CPU fallback)");
    {   // the library carries gfx950 code objects only (MFMA shapes, 160 KB LDS per workgroup): say so instead of a launch error later
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess)
            return fail(TAE_EHIP, "cannot query the current HIP device");
        if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
            return fail(TAE_EHIP, std::string("libturboae_hip is built for gfx950 (MI355X) only; the current device is ") + prop.gcnArchName);
    }
    if (tae::generic_needed(cfg)) {
        // outside the MFMA kernels' envelope: generic fp32 kernels, one launch per layer (turboae_generic.hip)
        tae_handle* h = new tae_handle();
        h->cfg = *cfg;
        h->nopts = default_norm_opts();
        h->noise_opts = default_noise_opts();
        h->U = cfg->enc_num_unit;
        h->Ud = cfg->dec_num_unit;
        h->nb = h->nbd = 0;
        h->prec = 0;
        (void)hipGetDevice(&h->device);
        rc = tae::generic_create(cfg, weights, n_weights, &h->gen);
        if (rc != TAE_OK) { delete h; return rc; }
        const int L = cfg->block_len;
        std::vector<int32_t> ident(L);
        for (int i = 0; i < L; ++i) ident[i] = i;
        hipError_t e = hipMalloc(&h->d_perm, L * sizeof(int32_t));
        if (e == hipSuccess) e = hipMalloc(&h->d_inv, L * sizeof(int32_t));
        if (e == hipSuccess) e = hipMalloc(&h->d_stats, 4 * sizeof(double));
        if (e == hipSuccess) e = hipMalloc(&h->d_flags, 4 * sizeof(uint32_t));
        if (e == hipSuccess) e = hipMemset(h->d_flags, 0, 4 * sizeof(uint32_t));
        if (e == hipSuccess) e = hipMemcpy(h->d_perm, ident.data(), L * sizeof(int32_t), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(h->d_inv, ident.data(), L * sizeof(int32_t), hipMemcpyHostToDevice);
        if (e != hipSuccess) { tae_destroy(h); return fail(TAE_EHIP, hipGetErrorString(e)); }
        rc = tae_reserve(h, cfg->max_batch > 0 ? cfg->max_batch : 1);
        if (rc != TAE_OK) { tae_destroy(h); return rc; }
        *out = h;
        return TAE_OK;
    }
    const tae_config user_cfg = *cfg;
    const float* const user_weights = weights;
    const size_t user_n_weights = n_weights;
    std::vector<float> w5;
    tae_config cfg5 = *cfg;
    if (needs_embedding(cfg)) {          // narrower stacks / kernel sizes 1, 3: run, exactly, in the next instantiated geometry
        w5 = embed_weights(cfg, weights, &cfg5);
        cfg = &cfg5;
        weights = w5.data();
        n_weights = w5.size();
    }
    tae_handle* h = new tae_handle();
    h->cfg = *cfg;
    h->nopts = default_norm_opts();
    h->noise_opts = default_noise_opts();
    h->U = cfg->enc_num_unit;
    h->Ud = cfg->dec_num_unit;
    (void)hipGetDevice(&h->device);
    if (hipDeviceGetAttribute(&h->ncu, hipDeviceAttributeMultiprocessorCount, h->device) != hipSuccess || h->ncu < 1) h->ncu = 256;
    if (const char* e = tae::debug_knob("TAE_GRU_L1")) h->gru_l1_split = !strcmp(e, "split");     // r04 form of the f16x2 GRU layer 1
    if (const char* e = tae::debug_knob("TAE_RNN_L1")) { h->rnn_l1_mode = !strcmp(e, "split") ? 1 : (!strcmp(e, "fused") ? 2 : 0); h->rnn_l1_check = !strcmp(e, "check"); }     // r05 form of the LSTM / RNN layer 1 (A/B, bit-identity test)
    if (const char* e = tae::debug_knob("TAE_GRU_L0")) h->gru_l0_mode = !strcmp(e, "block") ? 1 : (!strcmp(e, "unit") ? 2 : 0);     // A/B of the two bit-identical layer-0 kernels
    const char* fixed_nb = tae::debug_knob("TAE_FIXED_NB");
    h->fixed_nb = fixed_nb && fixed_nb[0] == '1';
    const int taps_e = cfg->enc_kernel_size, taps_d = cfg->dec_kernel_size;      // 5, 7 or 9 here (1 and 3 were embedded)
    const bool big_taps = taps_e > 5 || taps_d > 5;                               // f16x2 kernels only: no fp32 packing
    // precision of the conv kernels: config field, overridden by env TAE_PRECISION=f32|f16x2 (testing knob)
    int want_h2 = cfg->precision == TAE_PREC_F32 ? 0 : 1;
    if (const char* pe = tae::debug_knob("TAE_PRECISION")) {
        if (!strcmp(pe, "f32")) want_h2 = 0;
        else if (!strcmp(pe, "f16x2")) want_h2 = 1;
    }
    h->nb = choose_nb(h->U, cfg->block_len, &h->lds_bytes, want_h2 != 0, taps_e, 3 * cfg->enc_num_layer);
    h->nbd = choose_nb(h->Ud, cfg->block_len, &h->lds_bytes_d, want_h2 != 0, taps_d, 2 * cfg->num_iteration * cfg->dec_num_layer);
    // Testing knobs (documented in DESIGN.md): TAE_FORCE_SEGMENTED=1 selects the long-block path even
    // when whole blocks fit; TAE_SEG_T=<n> caps the centre length of a segment.
    const char* force_seg = tae::debug_knob("TAE_FORCE_SEGMENTED");
    if (force_seg && force_seg[0] == '1') h->nb = h->nbd = 0;
    if (cfg->dense) h->nb = h->nbd = 0;   // dense stacks run on the long-block kernels (one stack per launch)
    if (h->nb < 1 || h->nbd < 1) {
        h->nb = h->nbd = 0;               // one path for both sides (the exchange buffers and the workspace follow it)
        if (!choose_seg(h->U, cfg->block_len, cfg->enc_num_layer, &h->enc_T, &h->enc_T0, &h->enc_nseg, &h->enc_lds, cfg->dense != 0, want_h2 != 0, taps_e) ||
            !choose_seg(h->Ud, cfg->block_len, cfg->dec_num_layer, &h->dec_T, &h->dec_T0, &h->dec_nseg, &h->dec_lds, cfg->dense != 0, want_h2 != 0, taps_d)) {
            delete h;
            return fail(TAE_EINVAL, "too many conv layers for the segmented long-block kernels (halo exceeds the panel)");
        }
    }
    {   // positions x real channels a workgroup holds at least (one block, or the last segment of one): see kRangeMinValues
        const int le = h->nb >= 1 ? cfg->block_len : cfg->block_len - (h->enc_nseg - 1) * h->enc_T;
        const int ld = h->nbd >= 1 ? cfg->block_len : cfg->block_len - (h->dec_nseg - 1) * h->dec_T;
        h->enc_min_values = (le > 0 ? le : 1) * user_cfg.enc_num_unit;
        h->dec_min_values = (ld > 0 ? ld : 1) * user_cfg.dec_num_unit;
    }
    const Layout lo(h->U), lod(h->Ud);
    const int F = cfg->num_iter_ft;
    const char* nosup = tae::debug_knob("TAE_NO_SUPER");     // testing knob: force the padded-tile path
    h->super = (lo.sup && cfg->block_len % 4 == 0 && !(nosup && nosup[0] == '1')) ? 1 : 0;
    h->super_d = (lod.sup && cfg->block_len % 4 == 0 && !(nosup && nosup[0] == '1')) ? 1 : 0;
    h->enc_stride = (uint32_t)lo.stack_stride(cfg->enc_num_layer);
    h->dec_stride = (uint32_t)lod.stack_stride(cfg->dec_num_layer);
    std::vector<float> penc((size_t)3 * h->enc_stride, 0.0f), pdec((size_t)2 * cfg->num_iteration * h->dec_stride, 0.0f);
    h->enc_bytes = (uint32_t)(penc.size() * sizeof(float));
    h->dec_bytes = (uint32_t)(pdec.size() * sizeof(float));
    const float* src = weights;
    if (cfg->dense || big_taps) src = weights + n_weights;                 // dense stacks / kernel sizes 7, 9: f16x2 packing only (below)
    else if (cfg->enc_type == 1) src += 3 * rnn_stack_floats(100, 1, 1, (size_t)cell_gates(cfg->enc_rnn));   // ENC_interRNN: packed with the recurrent kernels' layouts below
    else for (int s = 0; s < 3; ++s) src += pack_stack(src, lo, cfg->enc_num_layer, 1, 1, penc.data() + (size_t)s * h->enc_stride);
    const float* dec_src;                   // first decoder weight in the canonical blob
    {
        tae_config enc_only = *cfg;
        enc_only.num_iteration = 0;
        dec_src = weights + num_weights(&enc_only);
    }
    if (cfg->dec_type == 1 || cfg->dense || big_taps) {
        src = weights + n_weights;          // canonical GRU weights are uploaded unchanged below
    } else {
        for (int it = 0; it < cfg->num_iteration; ++it)
            for (int half = 0; half < 2; ++half) {
                const int nout = (half == 1 && it == cfg->num_iteration - 1) ? 1 : F;
                src += pack_stack(src, lod, cfg->dec_num_layer, 2 + F, nout, pdec.data() + (size_t)(2 * it + half) * h->dec_stride);
            }
    }
    if ((size_t)(src - weights) != n_weights) {
        delete h;
        return fail(TAE_EINVAL, "internal: weight walk mismatch");
    }
    std::vector<char> penc_h, pdec_h;
    bool h2_ok = false;
    if (want_h2 && h->nb >= 1) {
        h->lds_bytes_h = tae::fused_lds_bytes_h(h->U, cfg->block_len, h->nb, taps_e, 3 * cfg->enc_num_layer);
        h->lds_bytes_hd = tae::fused_lds_bytes_h(h->Ud, cfg->block_len, h->nbd, taps_d, 2 * cfg->num_iteration * cfg->dec_num_layer);
        h2_ok = h->lds_bytes_h <= 160 * 1024 && h->lds_bytes_hd <= 160 * 1024;
    } else if (want_h2) {               // long-block path: same segment geometry, f16x2 panels
        h->enc_lds_h = tae::seg_lds_bytes_h(h->U, h->enc_T, cfg->enc_num_layer, taps_e);
        h->dec_lds_h = tae::seg_lds_bytes_h(h->Ud, h->dec_T, cfg->dec_num_layer, taps_d);
        h2_ok = h->enc_lds_h <= 160 * 1024 && h->dec_lds_h <= 160 * 1024;
    }
    if (cfg->dense) {
        if (!want_h2) { delete h; return fail(TAE_EINVAL, "DenseSameShapeConv1d stacks run on the fp16-split kernels only (TAE_PRECISION=f32 given)"); }
        // the dense geometry of choose_seg already is the f16x2 one
        h->enc_lds_h = h->enc_lds;
        h->dec_lds_h = h->dec_lds;
        h->prec = 1;
        const LayoutH lh(h->U), lhd(h->Ud);
        h->enc_stride_h = (uint32_t)dense_stack_bytes(lh, cfg->enc_num_layer);
        h->dec_stride_h = (uint32_t)dense_stack_bytes(lhd, cfg->dec_num_layer);
        penc_h.assign((size_t)3 * h->enc_stride_h, 0);
        pdec_h.assign((size_t)2 * cfg->num_iteration * h->dec_stride_h, 0);
        h->enc_bytes_h = (uint32_t)penc_h.size();
        h->dec_bytes_h = (uint32_t)pdec_h.size();
        const float* s2 = weights;
        for (int s = 0; s < 3; ++s)
            s2 += pack_stack_h_dense(s2, lh, cfg->enc_num_layer, 1, 1, penc_h.data() + (size_t)s * h->enc_stride_h, &h->enc_tails, (uint32_t)s * h->enc_stride_h);
        for (int it = 0; it < cfg->num_iteration; ++it)
            for (int half = 0; half < 2; ++half) {
                const int nout = (half == 1 && it == cfg->num_iteration - 1) ? 1 : F;
                s2 += pack_stack_h_dense(s2, lhd, cfg->dec_num_layer, 2 + F, nout, pdec_h.data() + (size_t)(2 * it + half) * h->dec_stride_h, &h->dec_tails,
                                         (uint32_t)(2 * it + half) * h->dec_stride_h);
            }
        if ((size_t)(s2 - weights) != n_weights) { delete h; return fail(TAE_EINVAL, "internal: dense weight walk mismatch"); }
    } else if (big_taps && !h2_ok) {
        delete h;
        return fail(TAE_EINVAL, "kernel sizes 7 / 9 need the fp16-split kernels (precision auto; TAE_PRECISION=f32 given, or the panels do not fit the LDS)");
    } else if (h2_ok) {
        h->prec = 1;
        if (cfg->precision == TAE_PREC_F16X1) {
            // the separately labelled reduced-precision decoder: the f16x2 handle as it is, decoder launches on the
            // one-product instantiation of the 100-wide whole-block kernel
            if (h->Ud != 100 || h->nbd < 1) { delete h; return fail(TAE_EINVAL, "TAE_PREC_F16X1 needs the 100-wide whole-block decoder kernel (block_len <= 320)"); }
            h->x1 = true;
        }
        LayoutH lh(h->U, taps_e), lhd(h->Ud, taps_d);
        lh.tail20 = h->U == 100 && taps_e == 5 && !cfg->dense;       // the plain-stack 100-wide kernels, whole-block and long-block (run_stack_h<.., T20>)
        lhd.tail20 = h->Ud == 100 && taps_d == 5 && !cfg->dense;
        h->enc_stride_h = (uint32_t)lh.stack_bytes(cfg->enc_num_layer);
        h->dec_stride_h = (uint32_t)lhd.stack_bytes(cfg->dec_num_layer);
        penc_h.assign((size_t)3 * h->enc_stride_h, 0);
        h->enc_bytes_h = (uint32_t)penc_h.size();
        const float* s2 = weights;
        if (cfg->enc_type == 0)
            for (int s = 0; s < 3; ++s)
                s2 += pack_stack_h(s2, lh, cfg->enc_num_layer, 1, 1, penc_h.data() + (size_t)s * h->enc_stride_h, &h->enc_tails, (uint32_t)s * h->enc_stride_h);
        if (cfg->dec_type == 0) {
            pdec_h.assign((size_t)2 * cfg->num_iteration * h->dec_stride_h, 0);
            h->dec_bytes_h = (uint32_t)pdec_h.size();
            for (int it = 0; it < cfg->num_iteration; ++it)
                for (int half = 0; half < 2; ++half) {
                    const int nout = (half == 1 && it == cfg->num_iteration - 1) ? 1 : F;
                    s2 += pack_stack_h(s2, lhd, cfg->dec_num_layer, 2 + F, nout, pdec_h.data() + (size_t)(2 * it + half) * h->dec_stride_h, &h->dec_tails,
                                       (uint32_t)(2 * it + half) * h->dec_stride_h);
                }
        }
    }
    const int L = cfg->block_len;
    std::vector<int32_t> ident(L);
    for (int i = 0; i < L; ++i) ident[i] = i;
#define TAE_HIP_H(expr)                                                                            \
    do {                                                                                           \
        hipError_t e__ = (expr);                                                                   \
        if (e__ != hipSuccess) { tae_destroy(h); return fail(TAE_EHIP, std::string(#expr) + ": " + hipGetErrorString(e__)); } \
    } while (0)
    TAE_HIP_H(hipMalloc(&h->d_wenc, penc.size() * sizeof(float)));
    TAE_HIP_H(hipMalloc(&h->d_wdec, pdec.size() * sizeof(float)));
    TAE_HIP_H(hipMalloc(&h->d_perm, L * sizeof(int32_t)));
    TAE_HIP_H(hipMalloc(&h->d_inv, L * sizeof(int32_t)));
    TAE_HIP_H(hipMalloc(&h->d_stats, 4 * sizeof(double)));
    TAE_HIP_H(hipMalloc(&h->d_flags, 4 * sizeof(uint32_t)));
    TAE_HIP_H(hipMemset(h->d_flags, 0, 4 * sizeof(uint32_t)));
    if (h->prec == 1) {
        TAE_HIP_H(hipMalloc(&h->d_wenc_h, penc_h.size()));
        TAE_HIP_H(hipMemcpy(h->d_wenc_h, penc_h.data(), penc_h.size(), hipMemcpyHostToDevice));
        if (!pdec_h.empty()) {
            TAE_HIP_H(hipMalloc(&h->d_wdec_h, pdec_h.size()));
            TAE_HIP_H(hipMemcpy(h->d_wdec_h, pdec_h.data(), pdec_h.size(), hipMemcpyHostToDevice));
        }
    }
    TAE_HIP_H(hipMemcpy(h->d_wenc, penc.data(), penc.size() * sizeof(float), hipMemcpyHostToDevice));
    TAE_HIP_H(hipMemcpy(h->d_wdec, pdec.data(), pdec.size() * sizeof(float), hipMemcpyHostToDevice));
    h->dec_gates = cfg->dec_type == 1 ? cell_gates(cfg->dec_rnn) : 3;
    h->enc_gates = cfg->enc_type == 1 ? cell_gates(cfg->enc_rnn) : 3;
    // ENC_interRNN (GRU cells, 2 layers) on the GRU kernels, whatever cell the decoder uses (r06: also in front of an LSTM / vanilla-RNN
    // decoder; the encoder shares the decoder's chunk workspace, not its kernels)
    auto pack_rnn_encoder = [&]() -> int {
        const std::vector<size_t> en(3, 1);
        if (h->enc_gates != 3) {      // r06: LSTM / vanilla-RNN cells of ENC_interRNN (encoders.py:242-253) on the unit-split f16x2 kernels
            if (h->prec != 1) { tae_destroy(h); return fail(TAE_EINVAL, "internal: the LSTM / RNN encoder kernels exist in the fp16-split arithmetic only"); }
            size_t bytes = 0;
            for (size_t nout : en) bytes += rnn_u_stack_bytes(nout, h->enc_gates);
            std::vector<char> pu(bytes, 0);
            repack_rnn_u(weights, pu.data(), 1, en, h->enc_gates, h->rnn_u_gimul_enc);
            TAE_HIP_H(hipMalloc(&h->d_wernn_u, pu.size()));
            TAE_HIP_H(hipMemcpy(h->d_wernn_u, pu.data(), pu.size(), hipMemcpyHostToDevice));
            return TAE_OK;
        }
        std::vector<float> pe(rnn_packed_floats(en), 0.0f);
        repack_rnn(weights, pe.data(), 100, 1, en);
        TAE_HIP_H(hipMalloc(&h->d_wernn, pe.size() * sizeof(float)));
        TAE_HIP_H(hipMemcpy(h->d_wernn, pe.data(), pe.size() * sizeof(float), hipMemcpyHostToDevice));
        if (h->prec == 1) {
            std::vector<char> peh(rnn_h_packed_bytes(en), 0);
            repack_rnn_h(weights, peh.data(), 1, en);
            TAE_HIP_H(hipMalloc(&h->d_wernn_h, peh.size()));
            TAE_HIP_H(hipMemcpy(h->d_wernn_h, peh.data(), peh.size(), hipMemcpyHostToDevice));
        }
        return TAE_OK;
    };
    if (cfg->dec_type == 1 && h->dec_gates != 3) {
        // LSTM / vanilla-RNN decoder (generic_needed left it here: CNN encoder, precision auto): unit-split f16x2 kernels only
        if (h->prec != 1) { tae_destroy(h); return fail(TAE_EINVAL, "internal: the LSTM / RNN decoder kernels exist in the fp16-split arithmetic only"); }
        const std::vector<size_t> nouts = dec_rnn_nouts((size_t)F, cfg->num_iteration);
        size_t bytes = 0;
        for (size_t nout : nouts) bytes += rnn_u_stack_bytes(nout, h->dec_gates);
        std::vector<char> pu(bytes, 0);
        repack_rnn_u(dec_src, pu.data(), 2 + (size_t)F, nouts, h->dec_gates, h->rnn_u_gimul);
        TAE_HIP_H(hipMalloc(&h->d_wrnn_u, pu.size()));
        TAE_HIP_H(hipMemcpy(h->d_wrnn_u, pu.data(), pu.size(), hipMemcpyHostToDevice));
        if (cfg->enc_type == 1) { rc = pack_rnn_encoder(); if (rc != TAE_OK) return rc; }
    } else if (cfg->dec_type == 1) {
        const size_t nrnn = (size_t)(weights + n_weights - dec_src);
        (void)nrnn;
        const std::vector<size_t> nouts = dec_rnn_nouts((size_t)F, cfg->num_iteration);
        std::vector<float> prnn(rnn_packed_floats(nouts), 0.0f);
        repack_rnn(dec_src, prnn.data(), 100, 2 + (size_t)F, nouts);
        TAE_HIP_H(hipMalloc(&h->d_wrnn, prnn.size() * sizeof(float)));
        TAE_HIP_H(hipMemcpy(h->d_wrnn, prnn.data(), prnn.size() * sizeof(float), hipMemcpyHostToDevice));
        if (h->prec == 1) {
            std::vector<char> prnn_h(rnn_h_packed_bytes(nouts), 0);
            repack_rnn_h(dec_src, prnn_h.data(), 2 + (size_t)F, nouts);
            TAE_HIP_H(hipMalloc(&h->d_wrnn_h, prnn_h.size()));
            TAE_HIP_H(hipMemcpy(h->d_wrnn_h, prnn_h.data(), prnn_h.size(), hipMemcpyHostToDevice));
        }
        if (cfg->enc_type == 1) { rc = pack_rnn_encoder(); if (rc != TAE_OK) return rc; }
    }
    TAE_HIP_H(hipMemcpy(h->d_perm, ident.data(), L * sizeof(int32_t), hipMemcpyHostToDevice));
    TAE_HIP_H(hipMemcpy(h->d_inv, ident.data(), L * sizeof(int32_t), hipMemcpyHostToDevice));
#undef TAE_HIP_H
    rc = tae_reserve(h, cfg->max_batch > 0 ? cfg->max_batch : 1);
    if (rc != TAE_OK) { tae_destroy(h); return rc; }
    if (h->prec == 1 && range_calibration_on(cfg)) {
        rc = calibrate_range(h, nullptr, nullptr, 0);
        if (rc != TAE_OK) { tae_destroy(h); return rc; }
    }
    if (h->prec == 1 && user_cfg.range_fallback) {
        // fp32 twin (fp32 MFMA kernels, or the generic fp32 kernels where those do not exist): a call that raises a range flag is
        // re-run there, and so is every later call
        tae_config fc = user_cfg;
        fc.precision = TAE_PREC_F32;
        fc.range_fallback = 0;
        rc = tae_create(&fc, user_weights, user_n_weights, &h->fb);
        if (rc != TAE_OK) { tae_destroy(h); return rc; }
        // the synthetic calibration batch above may have grown h->cap past max_batch: the twin serves every batch h accepts
        // (check_batch tests h->cap only; tae_reserve keeps the two in step from here on)
        if (h->fb->cap < h->cap) {
            rc = tae_reserve(h->fb, h->cap);
            if (rc != TAE_OK) { tae_destroy(h); return rc; }
        }
    }
    *out = h;
    return TAE_OK;
}

int tae_destroy(tae_handle* h) {
    if (!h) return TAE_OK;
    tae::generic_destroy(h->gen);
    h->gen = nullptr;
    if (h->fb) { tae_destroy(h->fb); h->fb = nullptr; }
    (void)hipFree(h->d_cal);
    (void)hipFree(h->d_wenc); (void)hipFree(h->d_wdec); (void)hipFree(h->d_perm); (void)hipFree(h->d_inv);
    (void)hipFree(h->d_xtx); (void)hipFree(h->d_rx); (void)hipFree(h->d_partials); (void)hipFree(h->d_stats);
    (void)hipFree(h->d_e0); (void)hipFree(h->d_e1);
    (void)hipFree(h->d_wrnn); (void)hipFree(h->d_gxa); (void)hipFree(h->d_gxb); (void)hipFree(h->d_gy0); (void)hipFree(h->d_gy1);
    (void)hipFree(h->d_ggi);
    (void)hipFree(h->d_wenc_h); (void)hipFree(h->d_wdec_h); (void)hipFree(h->d_flags); (void)hipFree(h->d_wrnn_h);
    (void)hipFree(h->d_wernn); (void)hipFree(h->d_wernn_h); (void)hipFree(h->d_rnn_partials); (void)hipFree(h->d_wrnn_u); (void)hipFree(h->d_wernn_u);
    (void)hipFree(h->d_eval_u); (void)hipFree(h->d_eval_noise); (void)hipFree(h->d_eval_xdec);
    delete h;
    return TAE_OK;
}

int tae_reserve(tae_handle* h, int32_t max_batch) {
    { const int rc_h = check_handle(h); if (rc_h != TAE_OK) return rc_h; }
    if (max_batch < 1) return fail(TAE_EINVAL, "max_batch must be >= 1");
    if (h->fb) { const int rc_f = tae_reserve(h->fb, max_batch); if (rc_f != TAE_OK) return rc_f; }
    if (max_batch <= h->cap) return TAE_OK;
    TAE_HIP(hipDeviceSynchronize());
    (void)hipFree(h->d_xtx); (void)hipFree(h->d_rx); (void)hipFree(h->d_partials); (void)hipFree(h->d_e0); (void)hipFree(h->d_e1);
    h->d_xtx = h->d_rx = h->d_e0 = h->d_e1 = nullptr; h->d_partials = nullptr; h->cap = 0;
    const size_t n3 = (size_t)max_batch * h->cfg.block_len * 3;
    if (h->gen) {
        const int rc_g = tae::generic_reserve(h->gen, max_batch);
        if (rc_g != TAE_OK) return rc_g;
        TAE_HIP(hipMalloc(&h->d_xtx, n3 * sizeof(float)));
        TAE_HIP(hipMalloc(&h->d_rx, n3 * sizeof(float)));
        h->cap = max_batch;
        return TAE_OK;
    }
    const size_t grid = h->nb >= 1 ? (size_t)max_batch : (size_t)3 * max_batch * h->enc_nseg;   // workgroups of the encoder at most (nb_for_batch may pick 1 block each)
    if (h->nbd < 1) {
        const size_t n8 = (size_t)max_batch * h->cfg.block_len * 8;
        TAE_HIP(hipMalloc(&h->d_e0, n8 * sizeof(float)));
        TAE_HIP(hipMalloc(&h->d_e1, n8 * sizeof(float)));
    }
    TAE_HIP(hipMalloc(&h->d_xtx, n3 * sizeof(float)));
    TAE_HIP(hipMalloc(&h->d_rx, n3 * sizeof(float)));
    TAE_HIP(hipMalloc(&h->d_partials, grid * 2 * sizeof(double)));
    if (h->cfg.dec_type == 1) {
        (void)hipFree(h->d_gxa); (void)hipFree(h->d_gxb); (void)hipFree(h->d_gy0); (void)hipFree(h->d_gy1); (void)hipFree(h->d_ggi);
        h->d_gxa = h->d_gxb = h->d_gy0 = h->d_gy1 = h->d_ggi = nullptr;
        // one full wave of recurrent workgroups = 256 CUs x 128 blocks / 2 directions; bound the workspace for long blocks
        int32_t chunk = 16384;
        while (chunk > 128 && (size_t)chunk * h->cfg.block_len > (size_t)16384 * 100) chunk -= 128;
        h->rnn_chunk = max_batch < chunk ? max_batch : chunk;
        const size_t np = (size_t)((h->rnn_chunk + 31) / 32 * 32) * h->cfg.block_len;     // whole block groups (16 per GRU workgroup, 32 per LSTM / RNN one)
        TAE_HIP(hipMalloc(&h->d_gxa, np * 8 * sizeof(float)));
        TAE_HIP(hipMalloc(&h->d_gxb, np * 8 * sizeof(float)));
        TAE_HIP(hipMalloc(&h->d_gy0, np * 200 * sizeof(float)));
        TAE_HIP(hipMalloc(&h->d_gy1, np * 200 * sizeof(float)));
        // GI (layer-1 input projections, 2.4 KB per position = 4 GB per 16 384-block chunk) exists only on the fp32 path and in the
        // r04 split form of the f16x2 path: the fused layer-1 kernel (turboae_gru_l1f.hip) never writes it
        // LSTM / RNN (r06): GI only where the split form of layer 1 runs - calls below rnn_l1_split_below(h) blocks (or every call with the
        // debug knob TAE_RNN_L1=split); the fused kernel never writes it
        // (the encoder's cell counts too: an LSTM encoder in front of a GRU decoder uses the same buffer for its own small calls)
        const bool gru_gi = (h->dec_gates == 3 || (h->cfg.enc_type == 1 && h->enc_gates == 3)) && (h->prec != 1 || h->gru_l1_split);
        const bool rnn_u = h->dec_gates != 3 || (h->cfg.enc_type == 1 && h->enc_gates != 3);
        const bool need_gi = gru_gi || (rnn_u && (h->rnn_l1_mode != 2 || h->rnn_l1_check));
        const int gmax = std::max(h->dec_gates, h->cfg.enc_type == 1 ? h->enc_gates : 1);
        const size_t gi_row = (size_t)2 * (6 * gmax + 1) * 16;          // floats per position: 2 directions x row tiles x 16 rows (GRU: 608)
        size_t np_gi = np;
        if (!gru_gi && h->rnn_l1_mode == 0 && !h->rnn_l1_check) {
            const size_t cap = (size_t)((rnn_l1_split_below(h) + 31) / 32 * 32) * h->cfg.block_len;
            if (cap < np_gi) np_gi = cap;
        }
        if (need_gi) TAE_HIP(hipMalloc(&h->d_ggi, np_gi * gi_row * sizeof(float)));
        TAE_HIP(hipMemset(h->d_gxa, 0, np * 8 * sizeof(float)));
        TAE_HIP(hipMemset(h->d_gxb, 0, np * 8 * sizeof(float)));
        // rows of padding blocks (last block group) are never written; they are read next to valid rows by the K-padding
        // over-read of the projection GEMM (x zero weights), so they must hold finite values
        TAE_HIP(hipMemset(h->d_gy0, 0, np * 200 * sizeof(float)));
        TAE_HIP(hipMemset(h->d_gy1, 0, np * 200 * sizeof(float)));
        if (need_gi) TAE_HIP(hipMemset(h->d_ggi, 0, np_gi * gi_row * sizeof(float)));
        if (h->cfg.enc_type == 1) {
            (void)hipFree(h->d_rnn_partials);
            h->d_rnn_partials = nullptr;
            const int nchunk = (max_batch + h->rnn_chunk - 1) / h->rnn_chunk;
            h->rnn_partial_slots = nchunk * 3 * tae::gru_head_grid(np);
            TAE_HIP(hipMalloc(&h->d_rnn_partials, (size_t)h->rnn_partial_slots * 2 * sizeof(double)));
        }
    }
    h->cap = max_batch;
    return TAE_OK;
}

}  // extern "C"
