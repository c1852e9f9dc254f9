// libturboae_hip.so, host side 2 of 3 - geometry and launch sequences (see turboae_host.hpp): blocks per workgroup / segment geometry,
// kernel parameter blocks, and the launch sequences of the encoder and decoder families (whole-block CNN, long-block CNN, GRU, generic).
#include "turboae_host.hpp"

namespace tae {
namespace host {

int choose_nb(int U, int L, int* lds_out, bool h2, int taps, int range_layers) {
    const int max_pos = tae::fused_max_positions();
    int nb = max_pos / L;
    auto bytes = [&](int n) { return h2 ? tae::fused_lds_bytes_h(U, L, n, taps, range_layers) : tae::fused_lds_bytes(U, L, n); };
    while (nb >= 1 && bytes(nb) > 160 * 1024) --nb;
    if (nb < 1) return 0;
    *lds_out = bytes(nb);
    return nb;
}

// Segment geometry of the long-block path.  A workgroup holds at most fused_max_positions() panel positions: T centre
// positions + H halo positions per side (+ up to 3 alignment rows); halos that would lie outside the block are not walked
// (the kernels clip the panel to [0, L)).  Segments are BALANCED (block_len 1000, H = 10: 4 x 250, 17 / 18 / 18 / 17 position
// tiles).  Measured on MI355X (tools/seg_ab.sh, 25 000 blocks of 1000): balanced 4 x 250 412-419 ms per forward; panel-filling
// segments with a short last one (310 + 297 + 297 + 96, T0 = T + H + 3) 434-441 ms although they walk 7 % fewer tile rows
// (full 20-tile workgroups run every SIMD at 5 tiles and clock lower; the short workgroup still pays the fixed prologue);
// 5 x 200 445 ms, 6 x 167 440 ms, 8 x 125 (two workgroups per CU) 545 ms.  T0 (segment 0 may own more centre positions, it
// has no left halo) is kept in the kernel interface and set to T.
bool choose_seg(int U, int L, int n_layer, int* T, int* T0, int* nseg, int* lds, bool dense, bool h2, int taps) {
    const int H = (taps / 2) * n_layer;
    auto seg_bytes = [&](int t) { return h2 ? tae::seg_lds_bytes_h(U, t, n_layer, taps) : tae::seg_lds_bytes(U, t, n_layer); };
    int tmax = tae::fused_max_positions() - 2 * H - 3;    // 3 alignment rows: panel origin floored to a multiple of 4
    if (dense) {
        // every earlier layer's output stays resident (n_layer - 1 panels): a segment is one position group (5 tiles) at most
        tmax = 80 - 2 * H - 3;
        while (tmax >= 8 && tae::seg_lds_bytes_h_dense(U, tmax, n_layer) > 160 * 1024) tmax -= 4;
        if (tmax < 8) return false;
        *nseg = (L + tmax - 1) / tmax;
        *T = (L + *nseg - 1) / *nseg;
        *T0 = *T;
        *lds = tae::seg_lds_bytes_h_dense(U, *T, n_layer);
        return true;
    }
    while (tmax >= 16 && seg_bytes(tmax) > 160 * 1024) tmax -= 16;
    if (tmax < 16) return false;
    const char* cap = tae::debug_knob("TAE_SEG_T");
    if (cap && atoi(cap) >= 1 && atoi(cap) < tmax) {      // testing knob: equal segments of at most this many centre positions
        tmax = atoi(cap);
        *nseg = (L + tmax - 1) / tmax;
        *T = (L + *nseg - 1) / *nseg;
        *T0 = *T;
        *lds = seg_bytes(*T);
        return true;
    }
    *nseg = (L + tmax - 1) / tmax;
    *T = (L + *nseg - 1) / *nseg;      // balanced segments
    *T0 = *T;
    *lds = seg_bytes(*T);
    return true;
}

// Blocks per workgroup for one call of the whole-block f16x2 kernels.  One workgroup is resident per CU and its time
// is set by the most loaded of its 4 position groups (group_span in turboae_h2.hip): measured on MI355X, about
// 0.33 + 0.135 * tiles (ms per decoder workgroup: 2 tiles 0.60, 5 tiles 1.00), i.e. proportional to 5 + 2 * tiles.
// A large batch wants the fullest workgroups (3 blocks of 100 -> 5 tiles per group); a batch that would leave CUs
// idle is cheaper spread thinner (500 blocks: 250 workgroups x 4 tiles instead of 167 x 5; <= 256 blocks: one block
// per workgroup, 2 tiles).  Results do not depend on the choice (blocks never see each other).
int nb_for_batch(const tae_handle* h, int32_t B, int nb_max) {
    if (h->fixed_nb || h->prec != 1) return nb_max;
    const int L = h->cfg.block_len;
    int best = nb_max;
    long best_cost = -1;
    for (int nb = nb_max; nb >= 1; --nb) {              // ties keep the larger nb (fewer passes over the weights)
        const long grid = ((long)B + nb - 1) / nb;
        const long rounds = (grid + h->ncu - 1) / h->ncu;
        const long ntile = ((long)nb * L + 15) / 16;
        const long cost = rounds * (5 + 2 * ((ntile + 3) / 4));
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = nb; }
    }
    return best;
}

// Grid of one call of the whole-block f16x2 kernels: full rounds of ncu workgroups with nb blocks each, and the blocks that are
// left (less than one round's worth) dealt nb_tail per workgroup with nb_tail chosen like nb_for_batch does: the last, partial
// round then costs 5 + 2 * ceil(tiles(nb_tail) / 4) instead of a full workgroup time (50 000 blocks on 256 CUs: 65 rounds of
// 256 x 3 blocks + 80 single-block workgroups instead of 27 three-block ones).  Returns the grid size.
int tail_geometry(const tae_handle* h, int32_t B, int nb, tae::FusedParams* P) {
    P->n_full = -1;
    P->nb_tail = nb;
    const int grid = (B + nb - 1) / nb;
    if (h->fixed_nb || nb <= 1 || grid <= h->ncu) return grid;       // a single round is nb_for_batch's business
    const int n_full = (B / nb) / h->ncu * h->ncu;                    // whole rounds of full workgroups
    const int rest = B - n_full * nb;                                 // < ncu * nb + nb blocks
    if (rest <= 0) return grid;
    const int L = h->cfg.block_len;
    int best = nb;
    long best_cost = -1;
    for (int t = nb; t >= 1; --t) {
        const long g = ((long)rest + t - 1) / t, rounds = (g + h->ncu - 1) / h->ncu;
        const long ntile = ((long)t * L + 15) / 16;
        const long cost = rounds * (5 + 2 * ((ntile + 3) / 4));
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = t; }
    }
    if (best == nb) return grid;
    P->n_full = n_full;
    P->nb_tail = best;
    return n_full + (rest + best - 1) / best;
}

// A handle's weights, workspace and kernel launches live on the device that was current at tae_create: a call made with another
// current device would launch there on foreign pointers (a fault, or silent peer traffic over xGMI).  One handle per GPU; a process
// that drives several GPUs makes the handle's device current before calling (hipSetDevice / torch.cuda.device).
int check_handle(tae_handle* h) {
    if (!h) return fail(TAE_EINVAL, "handle is NULL");
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess || cur != h->device)
        return fail(TAE_ESTATE, "handle belongs to device " + std::to_string(h->device) + " but the calling thread's current device is " +
                                    std::to_string(cur) + " (make the handle's device current first)");
    return TAE_OK;
}

int check_batch(tae_handle* h, int32_t B) {
    const int rc = check_handle(h);
    if (rc != TAE_OK) return rc;
    if (B < 1) return fail(TAE_EINVAL, "batch must be >= 1");
    if (B > h->cap) return fail(TAE_ESTATE, "batch exceeds reserved workspace; call tae_reserve first");
    return TAE_OK;
}

tae_noise_opts default_noise_opts() {
    tae_noise_opts o;
    o.struct_size = (int32_t)sizeof(tae_noise_opts);
    o.kind = TAE_NOISE_AWGN;
    o.vv = 5.0f; o.radar_prob = 0.05f; o.radar_power = 5.0f;      // get_args.py:53-56
    o.p_gg = 0.8f; o.p_bb = 0.8f;                                  // channels.py:60-61,86-87
    return o;
}

int check_noise_opts(const tae_noise_opts* o) {
    if (o->struct_size != (int32_t)sizeof(tae_noise_opts)) return fail(TAE_EINVAL, "tae_noise_opts.struct_size mismatch (ABI)");
    if (o->kind < TAE_NOISE_AWGN || o->kind > TAE_NOISE_FADING) return fail(TAE_EINVAL, "tae_noise_opts.kind must be one of TAE_NOISE_*");
    if (o->kind == TAE_NOISE_TDIST && !(o->vv > 2.0f)) return fail(TAE_EINVAL, "t-dist needs vv > 2 (the reference scales by sqrt((vv - 2) / vv), channels.py:41)");
    if (o->kind == TAE_NOISE_RADAR && !(o->radar_prob >= 0.0f && o->radar_prob <= 1.0f)) return fail(TAE_EINVAL, "radar_prob must be in [0, 1]");
    if ((o->kind == TAE_NOISE_GE || o->kind == TAE_NOISE_GE_AWGN) && !(o->p_gg >= 0.0f && o->p_gg <= 1.0f && o->p_bb >= 0.0f && o->p_bb <= 1.0f))
        return fail(TAE_EINVAL, "Gilbert-Elliott transition probabilities must be in [0, 1]");
    return TAE_OK;
}

// host-side derivation of the generator's constants from test_sigma (channels.py:27-31,62-63,88-89; utils.py:69-76)
int make_noise_gen(const tae_noise_opts* o, float test_sigma, tae::NoiseGen* g) {
    const bool mask = o->kind == TAE_NOISE_BEC || o->kind == TAE_NOISE_BSC || o->kind == TAE_NOISE_GE;
    if (mask && !(test_sigma >= 0.0f && test_sigma <= 1.0f)) return fail(TAE_EINVAL, "bec / bsc / ge: test_sigma is a probability in [0, 1]");
    const double sigma = pow(10.0, -(double)test_sigma / 20.0);          // snr_db2sigma
    const double snr_back = -20.0 * log10(sigma);                       // snr_sigma2db
    g->kind = o->kind;
    g->sigma = mask ? 0.0f : (float)sigma;
    g->p = mask ? test_sigma : 0.0f;
    g->s_good = (float)pow(10.0, -(snr_back + 1.0) / 20.0);
    g->s_bad = (float)pow(10.0, -(snr_back - 1.0) / 20.0);
    g->vv = o->vv; g->radar_prob = o->radar_prob; g->radar_power = o->radar_power; g->p_gg = o->p_gg; g->p_bb = o->p_bb;
    return TAE_OK;
}

tae::NormOpts default_norm_opts() {
    tae::NormOpts o;
    memset(&o, 0, sizeof(o));
    o.std = 1.0f;
    o.enc_value_limit = 1.0f;
    o.enc_quantize_level = 2.0f;
    o.rec_quantize_limit = 1.0f;
    o.rec_quantize_level = 2.0f;
    return o;
}

// Calibration array (tae_handle::d_cal, uint32 float bits): encoder part [0] unused | [1 + s * nl + l] layer maxima | then one slot per
// stack (unused: encoder inputs are +-1); decoder part at cal_dec_offset: [0] max |stack input| of the whole-block kernel |
// [1 + s * nl + l] | [1 + n_stack * nl + s] max |extrinsic value| stack s staged on the long-block path | [cal_dec_r] max |received value|.
size_t cal_dec_offset(const tae_handle* h) { return 1 + 3 * (size_t)h->cfg.enc_num_layer + 3; }
size_t cal_dec_r(const tae_handle* h) { return 1 + 2 * (size_t)h->cfg.num_iteration * ((size_t)h->cfg.dec_num_layer + 1); }     // relative to the decoder part
size_t cal_words(const tae_handle* h) { return cal_dec_offset(h) + cal_dec_r(h) + 1; }

tae::FusedParams base_params(const tae_handle* h, int32_t B, bool decoder) {
    tae::FusedParams P;
    memset(&P, 0, sizeof(P));
    P.perm = h->d_perm;
    P.inv = h->d_inv;
    P.B = B;
    P.L = h->cfg.block_len;
    P.nb = decoder ? h->nbd : h->nb;
    P.taps = decoder ? h->cfg.dec_kernel_size : h->cfg.enc_kernel_size;
    P.n_iter = h->cfg.num_iteration;
    P.F = h->cfg.num_iter_ft;
    P.extrinsic = h->cfg.extrinsic;
    P.act = h->cfg.enc_act;
    P.lds_bytes = decoder ? h->lds_bytes_d : h->lds_bytes;
    P.super = decoder ? h->super_d : h->super;
    // stack-input planes of the fp16-split kernels: the encoder's are +-1 (exponent 0), the whole-block decoder has one exponent
    const int ax = decoder && !h->dec_Ax.empty() ? h->dec_Ax[0] : 0;
    P.x_scale = ldexpf(1.0f, ax);
    P.x_inv = ldexpf(1.0f, -ax);
    P.x_low = decoder && h->calibrated && !h->calibrating ? ldexpf(h->dec_r_low, ax) : 0.0f;
    P.cal = h->calibrating ? h->d_cal + (decoder ? cal_dec_offset(h) : 0) : nullptr;
    P.cal_r = (int32_t)cal_dec_r(h);
    {   // see FusedParams::track / head2
        const std::vector<int>& K = decoder ? h->dec_kind : h->enc_kind;
        const int nl = decoder ? h->cfg.dec_num_layer : h->cfg.enc_num_layer;
        bool both = false;
        for (size_t i = 0; i < K.size(); ++i) both = both || ((int)(i % nl) == nl - 1 && K[i] == 2);
        P.track = h->calibrating ? 2 : (decoder ? 1 : (h->calibrated ? 0 : 2));
        P.head2 = both ? 1 : 0;          // production launch with both-branch heads: its own (spill-free) instantiation, not the full one
        P.prod = (decoder && h->x1) ? 1 : 3;
    }
    return P;
}

tae::SegParams seg_params(const tae_handle* h, int32_t B, bool decoder) {
    tae::SegParams P;
    memset(&P, 0, sizeof(P));
    P.perm = h->d_perm;
    P.inv = h->d_inv;
    P.B = B;
    P.L = h->cfg.block_len;
    P.F = h->cfg.num_iter_ft;
    P.extrinsic = h->cfg.extrinsic;
    P.act = h->cfg.enc_act;
    P.super = decoder ? h->super_d : h->super;
    P.taps = decoder ? h->cfg.dec_kernel_size : h->cfg.enc_kernel_size;
    P.dense = h->cfg.dense;
    for (int s = 0; s < 3; ++s) P.x_scale[s] = (!decoder && (size_t)s < h->enc_Ax.size()) ? ldexpf(1.0f, h->enc_Ax[s]) : 1.0f;
    P.x_low = 0.0f;                      // decoder: per launch (run_decoder_long), the exponent is the stack's
    P.cal = h->calibrating ? h->d_cal + (decoder ? cal_dec_offset(h) : 0) : nullptr;
    P.cal_r = (int32_t)cal_dec_r(h);
    return P;
}

int run_encoder_long(tae_handle* h, const float* u, float* xtx, double* stats, int32_t B, hipStream_t st) {
    tae::SegParams P = seg_params(h, B, false);
    P.wpack = h->d_wenc;
    P.in = u;
    P.out = xtx;
    P.partials = h->d_partials;
    P.mode = 0;
    P.T = h->enc_T;
    P.T0 = h->enc_T0;
    P.nseg = h->enc_nseg;
    P.n_layer = h->cfg.enc_num_layer;
    P.stack_stride = h->enc_stride;
    P.wpack_bytes = h->enc_bytes;
    P.lds_bytes = h->enc_lds;
    const int grid = 3 * B * h->enc_nseg;
    if (h->prec == 1) {
        P.wpack = reinterpret_cast<const float*>(h->d_wenc_h);
        P.stack_stride = h->enc_stride_h;
        P.wpack_bytes = h->enc_bytes_h;
        P.lds_bytes = h->enc_lds_h;
        P.flags = h->d_flags;
        TAE_HIP(tae::launch_seg_h(h->U, P, grid, st));
    } else
    TAE_HIP(tae::launch_seg(h->U, P, grid, st));
    TAE_HIP(tae::launch_reduce_partials(h->d_partials, grid, (double)B * h->cfg.block_len * 3.0, stats, st));
    return TAE_OK;
}

int run_decoder_long(tae_handle* h, const float* rx, float* xdec, int32_t B, hipStream_t st, float* tap_out = nullptr) {
    tae::SegParams P = seg_params(h, B, true);
    P.wpack = h->d_wdec;
    P.in = rx;
    P.out = xdec;
    P.mode = 1;
    P.T = h->dec_T;
    P.T0 = h->dec_T0;
    P.nseg = h->dec_nseg;
    P.n_layer = h->cfg.dec_num_layer;
    P.stack_stride = h->dec_stride;
    P.wpack_bytes = h->dec_bytes;
    P.lds_bytes = h->dec_lds;
    const int n_stack = 2 * h->cfg.num_iteration;
    const int grid = B * h->dec_nseg;
    if (h->prec == 1) {
        P.wpack = reinterpret_cast<const float*>(h->d_wdec_h);
        P.stack_stride = h->dec_stride_h;
        P.wpack_bytes = h->dec_bytes_h;
        P.lds_bytes = h->dec_lds_h;
        P.flags = h->d_flags;
    }
    for (int s = 0; s < n_stack; ++s) {
        P.stack = s;
        P.last = (s == n_stack - 1);
        P.x_scale[0] = (size_t)s < h->dec_Ax.size() ? ldexpf(1.0f, h->dec_Ax[s]) : 1.0f;
        P.x_low = (size_t)s < h->dec_Ax.size() && h->calibrated && !h->calibrating ? ldexpf(h->dec_r_low, h->dec_Ax[s]) : 0.0f;
        P.cal_x = 1 + n_stack * h->cfg.dec_num_layer + s;
        P.eprev = (s & 1) ? h->d_e0 : h->d_e1;
        P.ecur = (s & 1) ? h->d_e1 : h->d_e0;
        if (h->prec == 1) TAE_HIP(tae::launch_seg_h(h->Ud, P, grid, st));
        else TAE_HIP(tae::launch_seg(h->Ud, P, grid, st));
        if (tap_out && !P.last) {       // (B, L, 8) exchange rows -> compact (B, L, F)
            const size_t F = (size_t)h->cfg.num_iter_ft, rows = (size_t)B * h->cfg.block_len;
            TAE_HIP(hipMemcpy2DAsync(tap_out + (size_t)s * rows * F, F * sizeof(float), P.ecur, 8 * sizeof(float), F * sizeof(float), rows,
                                     hipMemcpyDeviceToDevice, st));
        }
    }
    return TAE_OK;
}

int run_encoder_rnn(tae_handle* h, const float* u, float* xtx, double* stats, int32_t B, hipStream_t st);
int run_encoder_rnn_u(tae_handle* h, const float* u, float* xtx, double* stats, int32_t B, hipStream_t st);

int run_encoder(tae_handle* h, const float* u, float* xtx, double* stats, int32_t B, hipStream_t st) {
    if (h->gen) return tae::generic_encode(h->gen, u, xtx, stats, h->d_perm, B, st);
    if (h->cfg.enc_type == 1) return h->enc_gates != 3 ? run_encoder_rnn_u(h, u, xtx, stats, B, st) : run_encoder_rnn(h, u, xtx, stats, B, st);
    if (h->nb < 1) return run_encoder_long(h, u, xtx, stats, B, st);
    tae::FusedParams P = base_params(h, B, false);
    P.wpack = h->d_wenc;
    P.in = u;
    P.out = xtx;
    P.partials = h->d_partials;
    P.n_layer = h->cfg.enc_num_layer;
    P.stack_stride = h->enc_stride;
    P.wpack_bytes = h->enc_bytes;
    P.nb = nb_for_batch(h, B, h->nb);
    P.n_full = -1;
    int grid = (B + P.nb - 1) / P.nb;
    if (h->prec == 1) {
        grid = tail_geometry(h, B, P.nb, &P);
        P.wpack = reinterpret_cast<const float*>(h->d_wenc_h);
        P.stack_stride = h->enc_stride_h;
        P.wpack_bytes = h->enc_bytes_h;
        P.lds_bytes = P.nb == h->nb ? h->lds_bytes_h : tae::fused_lds_bytes_h(h->U, h->cfg.block_len, P.nb, P.taps, 3 * h->cfg.enc_num_layer);
        P.flags = h->d_flags;
        TAE_HIP(tae::launch_fused_h(h->U, false, P, grid, st));
    } else
    TAE_HIP(tae::launch_fused(h->U, false, P, grid, st));
    TAE_HIP(tae::launch_reduce_partials(h->d_partials, grid, (double)B * h->cfg.block_len * 3.0, stats, st));
    return TAE_OK;
}

// layer 1 of a GRU stack as one kernel (f16x2 path): `img` = the stack's two GruL1fLayout images
// Layer 0 of an f16x2 GRU stack.  Two bit-identical kernels: one wave per 16 blocks (4.7 us per step whatever the batch - best when
// every SIMD of the chip holds two of them) or seven waves per 16 blocks, two workgroups per CU (a short step: 2.2x at 500 blocks,
// equal at a full 16 384-block chunk; tools/probes/gru_l0_ab.py).  Results do not depend on the choice (tests/test_gpu_parity.py).
hipError_t launch_gru_l0(const tae_handle* h, const tae::GruRecParams& R0, hipStream_t st) {
    const bool unit = h->gru_l0_mode == 2 || (h->gru_l0_mode == 0 && R0.B <= kGruL0UnitMaxB);
    return unit ? tae::launch_gru_rec0u(R0, st) : tae::launch_gru_rec_h(true, R0, st);
}

tae::GruL1fParams l1f_params(const tae_handle* h, const char* img, int32_t Bc) {
    tae::GruL1fParams F;
    memset(&F, 0, sizeof(F));
    F.w = img; F.w_dir_stride = (uint32_t)tae::GruL1fLayout::kDirB;
    F.y0 = reinterpret_cast<const char*>(h->d_gy0); F.hpart = h->d_gy1;
    F.B = Bc; F.L = h->cfg.block_len; F.ngroups = (Bc + 15) / 16;
    return F;
}

// ENC_interRNN.forward before power_constraint (encoders.py:281-296): three GRU stacks on the decoder's kernels
// (rec layer 0 -> projection -> rec layer 1 -> head in encoder mode), per internal chunk; every head workgroup leaves a
// partial (sum, sumsq) that reduce_partials adds in fixed order.
int run_encoder_rnn(tae_handle* h, const float* u, float* xtx, double* stats, int32_t B, hipStream_t st) {
    const int L = h->cfg.block_len, H = 100;
    int slot = 0;
    {   // every head launch writes gru_head_grid(npos) partial-sum slots: check the whole call BEFORE anything is launched
        long need = 0;
        for (int32_t c0 = 0; c0 < B; c0 += h->rnn_chunk) {
            const int32_t Bc = (B - c0 < h->rnn_chunk) ? B - c0 : h->rnn_chunk;
            const size_t npos = h->prec == 1 ? (size_t)((Bc + 15) / 16) * 16 * L : (size_t)Bc * L;
            need += 3L * tae::gru_head_grid(npos);
        }
        if (need > h->rnn_partial_slots) return fail(TAE_ESTATE, "internal: GRU-encoder partial-sum slots exceeded");
    }
    for (int32_t c0 = 0; c0 < B; c0 += h->rnn_chunk) {
        const int32_t Bc = (B - c0 < h->rnn_chunk) ? B - c0 : h->rnn_chunk;
        const size_t np = (size_t)Bc * L, npg = (size_t)((Bc + 15) / 16) * 16 * L;
        const float* w = h->d_wernn;
        const char* wb = h->d_wernn_h;
        for (int s = 0; s < 3; ++s) {
            TAE_HIP(tae::launch_gru_prep_enc(u + (size_t)c0 * L, h->d_perm, h->d_gxa, Bc, L, s == 2 ? 1 : 0, st));
            tae::GruRecParams R0, R1;
            tae::GruProjParams PP;
            memset(&PP, 0, sizeof(PP));
            tae::GruHeadParams HP;
            memset(&R0, 0, sizeof(R0)); memset(&R1, 0, sizeof(R1)); memset(&HP, 0, sizeof(HP));
            R0.x = h->d_gxa; R0.B = Bc; R0.L = L; R0.y = h->d_gy0;
            R1.gi = h->d_ggi; R1.B = Bc; R1.L = L; R1.y = h->d_gy1;
            PP.yin = h->d_gy0; PP.gi = h->d_ggi; PP.B = Bc; PP.L = L;
            const float* wl;
            if (h->prec == 1) {
                R0.w = reinterpret_cast<const float*>(wb); R0.w_dir_stride = (uint32_t)kGHRec0B;
                const char* w1 = wb + 2 * kGHRec0B;
                PP.w = reinterpret_cast<const float*>(w1); PP.npos = npg;
                R1.w = reinterpret_cast<const float*>(w1 + kGHProjB); R1.w_dir_stride = (uint32_t)kGHRec1B;
                wl = reinterpret_cast<const float*>(w1 + kGHProjB + 2 * kGHRec1B);
                R1.hpart = h->d_gy1;          // per-direction head products (the layer-1 recurrence contracts Y1 away)
                TAE_HIP(launch_gru_l0(h, R0, st));
                if (h->gru_l1_split) {
                    TAE_HIP(tae::launch_gru_proj_h(PP, st));
                    TAE_HIP(tae::launch_gru_rec_h(false, R1, st));
                } else {
                    TAE_HIP(tae::launch_gru_l1f(l1f_params(h, wb + rnn_h_l1f_offset(1), Bc), st));
                }
                wb += rnn_h_stack_bytes(1);
            } else {
                R0.w = w; R0.w_dir_stride = (uint32_t)kGL0Dir;
                const float* w1 = w + 2 * kGL0Dir;
                PP.w = w1; PP.npos = np;
                R1.w = w1 + kGProjF + kGPB; R1.w_dir_stride = (uint32_t)kGL1Dir;
                wl = w1 + kGProjF + kGPB + 2 * kGL1Dir;
                TAE_HIP(tae::launch_gru_rec(true, R0, st));
                TAE_HIP(tae::launch_gru_proj(PP, st));
                TAE_HIP(tae::launch_gru_rec(false, R1, st));
                w += rnn_packed_stack_floats(1);
            }
            HP.y = h->d_gy1; HP.w = wl; HP.b = wl + 2 * H; HP.npos = h->prec == 1 ? npg : np; HP.L = L; HP.F = 1; HP.nout = 1;
            HP.grouped = h->prec == 1 ? 1 : 0; HP.B = Bc;
            HP.enc_stack = s; HP.act = h->cfg.enc_act; HP.xtx = xtx + (size_t)c0 * L * 3;
            HP.partials = h->d_rnn_partials + (size_t)slot * 2;
            if (h->prec == 1) TAE_HIP(tae::launch_gru_head_part(HP, st));
            else TAE_HIP(tae::launch_gru_head(HP, st));
            slot += tae::gru_head_grid(HP.npos);
        }
    }
    TAE_HIP(tae::launch_reduce_partials(h->d_rnn_partials, slot, (double)B * L * 3.0, stats, st));
    return TAE_OK;
}

// One LSTM / vanilla-RNN stack on turboae_rnn_u.hip: layer 0 -> layer 1 (fused, or projection + recurrence below rnn_l1_split_below blocks;
// bit-identical) -> per-direction head products in d_gy1.  `wb`: the stack's image (repack_rnn_u), `gimul`: its two layer-1 scales.
int run_rnn_u_stack(tae_handle* h, int G, const char* wb, const float* xin, const float* gimul, int32_t Bc, size_t npg, hipStream_t st) {
    const size_t dirb = tae::RnnULayout::dir_bytes(G), projb = tae::RnnULayout::proj_bytes(G);
    tae::RnnUParams R;
    memset(&R, 0, sizeof(R));
    R.w = wb; R.w_dir_stride = (uint32_t)dirb; R.x = xin; R.y0 = reinterpret_cast<char*>(h->d_gy0);
    R.B = Bc; R.L = h->cfg.block_len; R.ncu = h->ncu;
    TAE_HIP(tae::launch_rnn_rec_u(G, true, R, st));
    R.w = wb + 2 * dirb + projb; R.x = nullptr; R.hpart = h->d_gy1;
    if (h->rnn_l1_mode == 1 || (h->rnn_l1_mode == 0 && Bc < rnn_l1_split_below(h))) {
        tae::RnnProjParams PP;
        memset(&PP, 0, sizeof(PP));
        PP.yin = h->d_gy0; PP.w = reinterpret_cast<const float*>(wb + 2 * dirb); PP.gi = h->d_ggi; PP.npos = npg;
        PP.gi_mul[0] = gimul[0]; PP.gi_mul[1] = gimul[1];
        TAE_HIP(tae::launch_rnn_proj_u(G, PP, st));
        R.gi = h->d_ggi; R.y0 = nullptr;
        TAE_HIP(tae::launch_rnn_rec_u(G, false, R, st));
    } else {
        R.wproj = wb + 2 * dirb; R.gi_mul[0] = gimul[0]; R.gi_mul[1] = gimul[1];
        TAE_HIP(tae::launch_rnn_l1f_u(G, R, st));
    }
    return TAE_OK;
}

// ENC_interRNN.forward with LSTM / vanilla-RNN cells (encoders.py:242-253,281-296) on the unit-split f16x2 kernels (r06): three stacks,
// each closed by gru_head_part in encoder mode (enc_act, x_tx column, partial sums for the power constraint)
int run_encoder_rnn_u(tae_handle* h, const float* u, float* xtx, double* stats, int32_t B, hipStream_t st) {
    const int L = h->cfg.block_len, H = 100, G = h->enc_gates;
    int slot = 0;
    {
        long need = 0;
        for (int32_t c0 = 0; c0 < B; c0 += h->rnn_chunk) {
            const int32_t Bc = (B - c0 < h->rnn_chunk) ? B - c0 : h->rnn_chunk;
            need += 3L * tae::gru_head_grid((size_t)((Bc + 15) / 16) * 16 * L);
        }
        if (need > h->rnn_partial_slots) return fail(TAE_ESTATE, "internal: recurrent-encoder partial-sum slots exceeded");
    }
    for (int32_t c0 = 0; c0 < B; c0 += h->rnn_chunk) {
        const int32_t Bc = (B - c0 < h->rnn_chunk) ? B - c0 : h->rnn_chunk;
        const size_t npg = (size_t)((Bc + 15) / 16) * 16 * L;
        const char* wb = h->d_wernn_u;
        for (int s = 0; s < 3; ++s) {
            TAE_HIP(tae::launch_gru_prep_enc(u + (size_t)c0 * L, h->d_perm, h->d_gxa, Bc, L, s == 2 ? 1 : 0, st));
            const int rc = run_rnn_u_stack(h, G, wb, h->d_gxa, &h->rnn_u_gimul_enc[2 * s], Bc, npg, st);
            if (rc != TAE_OK) return rc;
            const float* wl = reinterpret_cast<const float*>(wb + 4 * tae::RnnULayout::dir_bytes(G) + tae::RnnULayout::proj_bytes(G));
            tae::GruHeadParams HP;
            memset(&HP, 0, sizeof(HP));
            HP.y = h->d_gy1; HP.w = wl; HP.b = wl + 2 * H; HP.npos = npg; HP.L = L; HP.F = 1; HP.nout = 1;
            HP.grouped = 1; HP.B = Bc; HP.enc_stack = s; HP.act = h->cfg.enc_act; HP.xtx = xtx + (size_t)c0 * L * 3;
            HP.partials = h->d_rnn_partials + (size_t)slot * 2;
            TAE_HIP(tae::launch_gru_head_part(HP, st));
            slot += tae::gru_head_grid(HP.npos);
            wb += rnn_u_stack_bytes(1, G);
        }
    }
    TAE_HIP(tae::launch_reduce_partials(h->d_rnn_partials, slot, (double)B * L * 3.0, stats, st));
    return TAE_OK;
}

// DEC_LargeRNN.forward (decoders.py:84-149): per half-iteration rec(layer 0) -> proj -> rec(layer 1) -> head
int run_decoder_rnn(tae_handle* h, const float* rx, float* xdec, int32_t B, hipStream_t st, float* tap_out) {
    const int L = h->cfg.block_len, F = h->cfg.num_iter_ft, H = 100, n_iter = h->cfg.num_iteration;
    for (int32_t c0 = 0; c0 < B; c0 += h->rnn_chunk) {
        const int32_t Bc = (B - c0 < h->rnn_chunk) ? B - c0 : h->rnn_chunk;
        const size_t np = (size_t)Bc * L;
        TAE_HIP(tae::launch_gru_prep(rx + (size_t)c0 * L * 3, h->d_perm, h->d_gxa, h->d_gxb, Bc, L, st));
        if (h->prec == 1) {
            // f16x2 kernels: layer 0 writes Y0 as halves straight into the projection's operand layout
            const char* wb = h->d_wrnn_h;
            for (int s = 0; s < 2 * n_iter; ++s) {
                const bool odd = (s & 1) != 0, last = (s == 2 * n_iter - 1);
                const int nout = last ? 1 : F;
                const float* xin = odd ? h->d_gxb : h->d_gxa;
                tae::GruRecParams R0;
                memset(&R0, 0, sizeof(R0));
                R0.w = reinterpret_cast<const float*>(wb); R0.w_dir_stride = (uint32_t)kGHRec0B; R0.x = xin; R0.B = Bc; R0.L = L; R0.y = h->d_gy0;
                TAE_HIP(launch_gru_l0(h, R0, st));
                const char* w1 = wb + 2 * kGHRec0B;
                tae::GruProjParams PP;
            memset(&PP, 0, sizeof(PP));
                const size_t npg = (size_t)((Bc + 15) / 16) * 16 * L;       // block-group-major rows incl. the padding blocks of the last group
                PP.yin = h->d_gy0; PP.w = reinterpret_cast<const float*>(w1); PP.gi = h->d_ggi; PP.npos = npg; PP.B = Bc; PP.L = L;
                tae::GruRecParams R1;
                memset(&R1, 0, sizeof(R1));
                R1.w = reinterpret_cast<const float*>(w1 + kGHProjB); R1.w_dir_stride = (uint32_t)kGHRec1B; R1.gi = h->d_ggi; R1.B = Bc; R1.L = L; R1.hpart = h->d_gy1;
                if (h->gru_l1_split) {        // r04 form (debug knob): projection to HBM, then the block-split recurrence
                    TAE_HIP(tae::launch_gru_proj_h(PP, st));
                    TAE_HIP(tae::launch_gru_rec_h(false, R1, st));
                } else {
                    TAE_HIP(tae::launch_gru_l1f(l1f_params(h, wb + rnn_h_l1f_offset((size_t)nout), Bc), st));
                }
                const float* wl = reinterpret_cast<const float*>(w1 + kGHProjB + 2 * kGHRec1B);
                tae::GruHeadParams HP;
                memset(&HP, 0, sizeof(HP));
                HP.y = h->d_gy1; HP.w = wl; HP.b = wl + (size_t)nout * 2 * H; HP.xcur = xin;
                HP.xnext = odd ? h->d_gxa : h->d_gxb; HP.xdec = xdec + (size_t)c0 * L;
                HP.ptab = odd ? h->d_perm : h->d_inv;
                HP.npos = npg; HP.L = L; HP.F = F; HP.nout = nout; HP.extrinsic = h->cfg.extrinsic; HP.last = last ? 1 : 0;
                HP.grouped = 1; HP.B = Bc; HP.enc_stack = -1; HP.act = h->cfg.dec_act;
                HP.tap = (tap_out && !last) ? tap_out + ((size_t)s * B + c0) * L * F : nullptr;
                TAE_HIP(tae::launch_gru_head_part(HP, st));
                wb += rnn_h_stack_bytes((size_t)nout);
            }
            continue;
        }
        const float* w = h->d_wrnn;
        for (int s = 0; s < 2 * n_iter; ++s) {
            const bool odd = (s & 1) != 0, last = (s == 2 * n_iter - 1);
            const int nout = last ? 1 : F;
            const float* xin = odd ? h->d_gxb : h->d_gxa;
            tae::GruRecParams R0;
            memset(&R0, 0, sizeof(R0));
            R0.w = w; R0.w_dir_stride = (uint32_t)kGL0Dir; R0.x = xin; R0.B = Bc; R0.L = L; R0.y = h->d_gy0;
            TAE_HIP(tae::launch_gru_rec(true, R0, st));
            const float* w1 = w + 2 * kGL0Dir;
            tae::GruProjParams PP;
            memset(&PP, 0, sizeof(PP));
            PP.yin = h->d_gy0; PP.w = w1; PP.gi = h->d_ggi; PP.npos = np; PP.B = Bc; PP.L = L;
            TAE_HIP(tae::launch_gru_proj(PP, st));
            tae::GruRecParams R1;
            memset(&R1, 0, sizeof(R1));
            R1.w = w1 + kGProjF + kGPB; R1.w_dir_stride = (uint32_t)kGL1Dir; R1.gi = h->d_ggi; R1.B = Bc; R1.L = L; R1.y = h->d_gy1;
            TAE_HIP(tae::launch_gru_rec(false, R1, st));
            const float* wl = w1 + kGProjF + kGPB + 2 * kGL1Dir;
            tae::GruHeadParams HP;
            memset(&HP, 0, sizeof(HP));
            HP.y = h->d_gy1; HP.w = wl; HP.b = wl + (size_t)nout * 2 * H; HP.xcur = xin;
            HP.xnext = odd ? h->d_gxa : h->d_gxb; HP.xdec = xdec + (size_t)c0 * L;
            HP.ptab = odd ? h->d_perm : h->d_inv;     // dec1 -> interleave (row inv[t]); dec2 -> deinterleave (row p[i])
            HP.npos = np; HP.L = L; HP.F = F; HP.nout = nout; HP.extrinsic = h->cfg.extrinsic; HP.last = last ? 1 : 0;
            HP.enc_stack = -1; HP.act = h->cfg.dec_act;
            HP.tap = (tap_out && !last) ? tap_out + ((size_t)s * B + c0) * L * F : nullptr;
            TAE_HIP(tae::launch_gru_head(HP, st));
            w += rnn_packed_stack_floats((size_t)nout);
        }
    }
    return TAE_OK;
}

int rnn_l1_split_below(const tae_handle* h) { return 6 * h->ncu; }      // 2 * ceil(B / 16) workgroups < 3/4 of the CUs

// DEC_LargeRNN.forward with an LSTM / vanilla-RNN cell (decoders.py:27-32,84-149) on the unit-split f16x2 kernels (turboae_rnn_u.hip):
// per half-iteration rec(layer 0) -> layer 1 (projection + recurrence + head tile in one kernel since r06; TAE_RNN_L1=split: the r05 pair) -> gru_head_part
int run_decoder_rnn_u(tae_handle* h, const float* rx, float* xdec, int32_t B, hipStream_t st, float* tap_out) {
    const int L = h->cfg.block_len, F = h->cfg.num_iter_ft, H = 100, n_iter = h->cfg.num_iteration, G = h->dec_gates;
    const size_t dirb = tae::RnnULayout::dir_bytes(G), projb = tae::RnnULayout::proj_bytes(G);
    for (int32_t c0 = 0; c0 < B; c0 += h->rnn_chunk) {
        const int32_t Bc = (B - c0 < h->rnn_chunk) ? B - c0 : h->rnn_chunk;
        TAE_HIP(tae::launch_gru_prep(rx + (size_t)c0 * L * 3, h->d_perm, h->d_gxa, h->d_gxb, Bc, L, st));
        const char* wb = h->d_wrnn_u;
        const size_t npg = (size_t)((Bc + 15) / 16) * 16 * L;        // block-group-major rows incl. the padding blocks of the last group of 16
        for (int s = 0; s < 2 * n_iter; ++s) {
            const bool odd = (s & 1) != 0, last = (s == 2 * n_iter - 1);
            const int nout = last ? 1 : F;
            const float* xin = odd ? h->d_gxb : h->d_gxa;
            tae::RnnUParams R;
            memset(&R, 0, sizeof(R));
            R.w = wb; R.w_dir_stride = (uint32_t)dirb; R.x = xin; R.y0 = reinterpret_cast<char*>(h->d_gy0);
            R.B = Bc; R.L = L; R.ncu = h->ncu;
            TAE_HIP(tae::launch_rnn_rec_u(G, true, R, st));
            if (h->rnn_l1_mode == 1 || (h->rnn_l1_mode == 0 && Bc < rnn_l1_split_below(h))) {        // r05 form: projection GEMM to HBM (GI), then the recurrence
                tae::RnnProjParams PP;
                memset(&PP, 0, sizeof(PP));
                PP.yin = h->d_gy0; PP.w = reinterpret_cast<const float*>(wb + 2 * dirb); PP.gi = h->d_ggi; PP.npos = npg;
                PP.gi_mul[0] = h->rnn_u_gimul[2 * s]; PP.gi_mul[1] = h->rnn_u_gimul[2 * s + 1];
                TAE_HIP(tae::launch_rnn_proj_u(G, PP, st));
                R.w = wb + 2 * dirb + projb; R.x = nullptr; R.gi = h->d_ggi; R.y0 = nullptr; R.hpart = h->d_gy1;
                TAE_HIP(tae::launch_rnn_rec_u(G, false, R, st));
            } else {                      // r06: the projection inside the recurrence (rnn_l1f_u_kernel), bit-identical, GI never exists
                R.w = wb + 2 * dirb + projb; R.x = nullptr; R.gi = nullptr; R.hpart = h->d_gy1;
                if (h->rnn_l1_check) {   // debug (TAE_RNN_L1=check, -DTAE_L1F_DBG_GI builds): GI from the projection kernel beside the fused kernel
                    tae::RnnProjParams PP;
                    memset(&PP, 0, sizeof(PP));
                    PP.yin = h->d_gy0; PP.w = reinterpret_cast<const float*>(wb + 2 * dirb); PP.gi = h->d_ggi; PP.npos = npg;
                    PP.gi_mul[0] = h->rnn_u_gimul[2 * s]; PP.gi_mul[1] = h->rnn_u_gimul[2 * s + 1];
                    TAE_HIP(tae::launch_rnn_proj_u(G, PP, st));
                    R.gi = h->d_ggi;
                }
                R.wproj = wb + 2 * dirb; R.gi_mul[0] = h->rnn_u_gimul[2 * s]; R.gi_mul[1] = h->rnn_u_gimul[2 * s + 1];
                TAE_HIP(tae::launch_rnn_l1f_u(G, R, st));
            }
            const float* wl = reinterpret_cast<const float*>(wb + 4 * dirb + projb);
            tae::GruHeadParams HP;
            memset(&HP, 0, sizeof(HP));
            HP.y = h->d_gy1; HP.w = wl; HP.b = wl + (size_t)nout * 2 * H; HP.xcur = xin;
            HP.xnext = odd ? h->d_gxa : h->d_gxb; HP.xdec = xdec + (size_t)c0 * L;
            HP.ptab = odd ? h->d_perm : h->d_inv;
            HP.npos = npg; HP.L = L; HP.F = F; HP.nout = nout; HP.extrinsic = h->cfg.extrinsic; HP.last = last ? 1 : 0;
            HP.grouped = 1; HP.B = Bc; HP.enc_stack = -1; HP.act = h->cfg.dec_act;
            HP.tap = (tap_out && !last) ? tap_out + ((size_t)s * B + c0) * L * F : nullptr;
            TAE_HIP(tae::launch_gru_head_part(HP, st));
            wb += rnn_u_stack_bytes((size_t)nout, G);
        }
    }
    return TAE_OK;
}

int run_decoder(tae_handle* h, const float* rx, float* xdec, int32_t B, hipStream_t st, float* tap_out) {
    if (h->gen) {
        return tae::generic_decode(h->gen, rx, xdec, h->d_perm, h->d_inv, B, st, tap_out);
    }
    if (h->cfg.dec_type == 1) {
        if (h->dec_gates != 3) return run_decoder_rnn_u(h, rx, xdec, B, st, tap_out);
        return run_decoder_rnn(h, rx, xdec, B, st, tap_out);
    }
    if (h->nbd < 1) return run_decoder_long(h, rx, xdec, B, st, tap_out);
    tae::FusedParams P = base_params(h, B, true);
    P.tap_out = tap_out;
    P.wpack = h->d_wdec;
    P.in = rx;
    P.out = xdec;
    P.n_layer = h->cfg.dec_num_layer;
    P.stack_stride = h->dec_stride;
    P.wpack_bytes = h->dec_bytes;
    P.nb = nb_for_batch(h, B, h->nbd);
    P.n_full = -1;
    int grid = (B + P.nb - 1) / P.nb;
    if (h->prec == 1) {
        grid = tail_geometry(h, B, P.nb, &P);
        P.wpack = reinterpret_cast<const float*>(h->d_wdec_h);
        P.stack_stride = h->dec_stride_h;
        P.wpack_bytes = h->dec_bytes_h;
        P.lds_bytes = P.nb == h->nbd ? h->lds_bytes_hd : tae::fused_lds_bytes_h(h->Ud, h->cfg.block_len, P.nb, P.taps, 2 * h->cfg.num_iteration * h->cfg.dec_num_layer);
        P.flags = h->d_flags;
        TAE_HIP(tae::launch_fused_h(h->Ud, true, P, grid, st));
        return TAE_OK;
    }
    TAE_HIP(tae::launch_fused(h->Ud, true, P, grid, st));
    return TAE_OK;
}

}  // namespace host
}  // namespace tae
