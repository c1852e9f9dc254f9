"""Case generators shared by the GPU fuzz (tests/test_gpu_fuzz.py) and the build-container pin of the oracle against the real
reference (oracle/fuzz_vs_reference.py), so that both walk the same configuration space.  No tests live here: editing a test file
cannot change what the pin script draws."""
import numpy as np

from turboae_amd import TurboAEConfig, philox, weights as W


def _sigma(snr_db):           # utils.py:69-70 (snr_db2sigma), restated to keep this helper free of oracle imports
    return 10 ** (-snr_db * 1.0 / 20)


EDGE_LENS = [1, 2, 3, 4, 5, 7, 15, 16, 17, 31, 32, 33, 48, 63, 79, 80, 81, 99, 101, 106, 107, 159, 160, 161,
             318, 319, 320, 321, 322, 323, 400]


def draw_cases(n, seed):
    rng = np.random.RandomState(seed)
    cases = []
    for i in range(n):
        widths = [32, 64, 100, 100, int(rng.randint(1, 101))]                 # any width up to 100 (narrow ones run embedded)
        U = int(rng.choice(widths))
        Ud = U if rng.rand() < 0.5 else int(rng.choice(widths))               # encoder and decoder widths are independent
        L = int(rng.choice(EDGE_LENS)) if rng.rand() < 0.6 else int(rng.randint(1, 420))
        nb_guess = max(1, 320 // L)
        B = int(rng.choice([1, 2, nb_guess, nb_guess + 1, 2 * nb_guess + 1, int(rng.randint(1, 48))]))
        cases.append(dict(block_len=L, enc_num_unit=U, dec_num_unit=Ud, enc_num_layer=int(rng.randint(1, 6)),
                          dec_num_layer=int(rng.randint(1, 6)), num_iter_ft=int(rng.randint(1, 7)),
                          num_iteration=int(rng.randint(1, 4)), extrinsic=int(rng.randint(0, 2)),
                          enc_kernel_size=int(rng.choice([5, 5, 3, 1, 7, 9])), dec_kernel_size=int(rng.choice([5, 5, 3, 1, 7, 9])),
                          enc_act=str(rng.choice(["elu", "linear", "elu", "tanh", "relu", "selu", "sigmoid"])), B=B, fixed_nb=str(int(rng.randint(0, 2))), wseed=int(rng.randint(1, 1 << 30))))
    return cases


def draw_wide_cases(n, seed):
    """Channel widths 101 .. 124 on at least one side (the 124-wide instantiations of the MFMA kernels: fp16-split since r04, fp32 twins since late r05;
    precision='f32' with kernel sizes 7 / 9 takes a network to the generic kernels).  Its own generator: draw_cases' random stream is pinned by oracle/fuzz_vs_reference.py."""
    rng = np.random.RandomState(seed)
    cases = []
    for i in range(n):
        wide = lambda: int(rng.randint(101, 125))                                # noqa: E731
        U, Ud = (wide(), wide()) if i % 3 == 0 else ((wide(), int(rng.choice([32, 64, 100, int(rng.randint(1, 101))]))) if i % 3 == 1
                                                     else (int(rng.choice([32, 64, 100])), wide()))
        L = int(rng.choice([1, 7, 16, 33, 100, 106, 160, 161, 213, 214, 320, 321, 400]))
        B = int(rng.choice([1, 2, 3, 5, 9, int(rng.randint(1, 24))]))
        cases.append(dict(block_len=L, enc_num_unit=U, dec_num_unit=Ud, enc_num_layer=int(rng.randint(1, 4)), dec_num_layer=int(rng.randint(1, 4)),
                          num_iter_ft=int(rng.randint(1, 7)), num_iteration=int(rng.randint(1, 3)), extrinsic=int(rng.randint(0, 2)),
                          enc_kernel_size=int(rng.choice([5, 5, 3, 7])), dec_kernel_size=int(rng.choice([5, 5, 1, 9])),
                          enc_act=str(rng.choice(["elu", "linear", "tanh"])), B=B, fixed_nb=str(int(rng.randint(0, 2))), wseed=int(rng.randint(1, 1 << 30))))
    return cases


def draw_variant_cases(n, seed):
    """GRU decoder (CNN or GRU encoder) and DenseSameShapeConv1d stacks: fixed widths, random lengths / batches."""
    rng = np.random.RandomState(seed)
    cases = []
    for i in range(n):
        kind = ["dec_rnn", "enc_rnn", "dense"][i % 3]
        L = int(rng.choice([1, 2, 5, 15, 16, 17, 33, 64, 100, 127])) if rng.rand() < 0.6 else int(rng.randint(1, 140))
        B = int(rng.choice([1, 2, 15, 16, 17, 31, 33, int(rng.randint(1, 70))]))
        c = dict(block_len=L, num_iter_ft=int(rng.randint(1, 7)), num_iteration=int(rng.randint(1, 3)), extrinsic=int(rng.randint(0, 2)),
                 B=B, wseed=int(rng.randint(1, 1 << 30)), kind=kind)
        acts = ["linear", "elu", "tanh", "relu", "selu", "sigmoid"]
        if kind == "dec_rnn":
            c.update(decoder="TurboAE_rate3_rnn", enc_num_layer=int(rng.randint(1, 4)), dec_act=str(rng.choice(acts)),
                     enc_num_unit=int(rng.choice([32, 64, 100, int(rng.randint(1, 101))])),
                     dec_num_unit=int(rng.choice([100, 100, int(rng.randint(1, 101))])))       # GRU widths below 100 run embedded
        elif kind == "enc_rnn":
            c.update(encoder="TurboAE_rate3_rnn", decoder="TurboAE_rate3_rnn", enc_act=str(rng.choice(acts)), dec_act=str(rng.choice(acts)),
                     enc_num_unit=int(rng.choice([100, int(rng.randint(1, 101))])), dec_num_unit=int(rng.choice([100, int(rng.randint(1, 101))])))
        else:
            U = int(rng.choice([32, 64]))
            c.update(encoder="TurboAE_rate3_cnn_dense", decoder="TurboAE_rate3_cnn_dense", enc_num_unit=U, dec_num_unit=int(rng.choice([32, 64])),
                     enc_num_layer=int(rng.randint(1, 4)), dec_num_layer=int(rng.randint(1, 4)))
        cases.append(c)
    return cases


def draw_channel_cases(n, seed):
    """Random combinations of the encoder-output / channel options (power_constraint variants encoders.py:102-125, channel
    branches channel_ae.py:40-69) on small CNN networks."""
    rng = np.random.RandomState(seed)
    cases = []
    for i in range(n):
        U = int(rng.choice([32, 64, 100]))
        c = dict(block_len=int(rng.choice([16, 40, 64, 100, 101])), enc_num_unit=U, dec_num_unit=U, enc_num_layer=int(rng.randint(1, 3)),
                 dec_num_layer=int(rng.randint(1, 4)), num_iteration=int(rng.randint(1, 3)), num_iter_ft=int(rng.randint(1, 6)),
                 train_channel_mode=str(rng.choice(["block_norm", "block_norm_ste", "block_norm_ste"])),
                 enc_value_limit=float(rng.choice([1.0, 1.5])), enc_quantize_level=float(rng.choice([2.0, 4.0, 8.0])),
                 enc_truncate_limit=float(rng.choice([0.0, 0.0, 1.2, 2.0])), no_code_norm=bool(rng.rand() < 0.15),
                 channel=str(rng.choice(["awgn", "bec", "bsc", "fading", "t-dist", "ge_awgn", "radar", "ge"])),
                 rec_quantize=bool(rng.rand() < 0.4), rec_quantize_level=int(rng.choice([2, 4])),
                 B=int(rng.choice([1, 3, 8, 17])), wseed=int(rng.randint(1, 1 << 30)))
        cases.append(c)
    return cases


def channel_case_inputs(case):
    """(cfg, state_dict, u, noise, fading) of a channel case - shared with oracle/fuzz_vs_reference.py."""
    case = dict(case)
    B, wseed = case.pop("B"), case.pop("wseed")
    cfg = TurboAEConfig(**case)
    L = cfg.block_len
    sd = W.generate_state_dict(cfg, seed=wseed, gain=1.0)
    u = philox.random_bits(wseed, 0, B * L).reshape(B, L, 1)
    if cfg.channel in ("bec", "bsc", "ge"):       # 0 / 1 keep masks (channels.py:48-54), erase / flip probability 0.1
        w = philox.random_u32(wseed, philox.STREAM_NOISE, 0, B * L * 3).astype(np.float64) / 2.0 ** 32
        noise = (w >= 0.1).astype(np.float32).reshape(B, L, 3)
    else:
        noise = (np.float32(_sigma(1.0)) * philox.random_normal(wseed, 0, B * L * 3)).reshape(B, L, 3).astype(np.float32)
    fading = None
    if cfg.channel == "fading":
        a, b = philox.random_normal(wseed + 1, 0, B * L * 3), philox.random_normal(wseed + 2, 0, B * L * 3)
        fading = (np.sqrt(a.astype(np.float64) ** 2 + b.astype(np.float64) ** 2) / np.sqrt(3.14 / 2.0)).astype(np.float32).reshape(B, L, 3)
    return cfg, sd, B, u, noise, fading


def draw_generic_cases(n, seed):
    """Random configurations OUTSIDE the MFMA kernels' envelope (generic fp32 kernels, csrc/turboae_generic.hip): wide stacks, many taps,
    many features, LSTM / RNN cells, ENC_interRNN depths, RNN encoder + (dense) CNN decoder."""
    rng = np.random.RandomState(seed)
    cases = []
    for i in range(n):
        kind = ["wide", "bigk", "ft", "lstm", "rnn", "enc_rnn", "rnn_cnn"][i % 7]
        c = dict(block_len=int(rng.choice([1, 7, 31, 32, 33, 64, 90])), num_iteration=int(rng.randint(1, 3)), num_iter_ft=int(rng.randint(1, 6)),
                 extrinsic=int(rng.randint(0, 2)), enc_num_unit=int(rng.randint(4, 40)), dec_num_unit=int(rng.randint(4, 40)),
                 enc_num_layer=int(rng.randint(1, 4)), dec_num_layer=int(rng.randint(1, 4)),
                 enc_act=str(rng.choice(["elu", "linear", "tanh"])), B=int(rng.choice([1, 2, 5])), wseed=int(rng.randint(1, 1 << 30)), kind=kind)
        if kind == "wide":
            c.update(enc_num_unit=int(rng.randint(101, 200)), dec_num_unit=int(rng.randint(101, 260)))
        elif kind == "bigk":
            c.update(enc_kernel_size=int(rng.choice([11, 15, 21])), dec_kernel_size=int(rng.choice([11, 13, 63])))
        elif kind == "ft":
            c.update(num_iter_ft=int(rng.randint(7, 20)))
        elif kind in ("lstm", "rnn"):
            c.update(decoder="TurboAE_rate3_rnn", dec_rnn=kind, dec_act=str(rng.choice(["linear", "tanh", "elu"])))
        elif kind == "enc_rnn":
            c.update(encoder="TurboAE_rate3_rnn", decoder="TurboAE_rate3_rnn", enc_rnn=str(rng.choice(["gru", "lstm", "rnn"])),
                     dec_rnn=str(rng.choice(["gru", "lstm", "rnn"])), enc_num_layer=int(rng.choice([1, 3, 4])))
        else:
            c.update(encoder="TurboAE_rate3_rnn", decoder="TurboAE_rate3_cnn", enc_rnn=str(rng.choice(["gru", "lstm"])))
        cases.append(c)
    return cases


# ---- activation-range cases: the reference's fp32 conv (cnn_utils.py:36-46) is scale-invariant, so a drop-in must be
def scale_layers(sd, cfg, factors_enc=None, factors_dec=None):
    """Copy of `sd` with conv layer l of every encoder / decoder stack rescaled: weight *= f[l], bias *= f[0] * ... * f[l] - in the
    linear regime the stack computes prod(f) times what it did, through intermediate activations prod(f[:l+1]) times as large.
    With prod(f) == 1 the network stays the 'same' function (exactly so without the ELU) but walks small / large panels."""
    out = {k: np.array(v, dtype=np.float32, copy=True) for k, v in sd.items()}

    def apply(prefixes, n_layer, f):
        if f is None:
            return
        assert len(f) == n_layer
        for pre in prefixes:
            cum = 1.0
            for l in range(n_layer):
                cum *= float(f[l])
                out[f"{pre}.cnns.{l}.weight"] *= np.float32(f[l])
                out[f"{pre}.cnns.{l}.bias"] *= np.float32(cum)
    apply([f"enc.enc_cnn_{s}" for s in (1, 2, 3)], cfg.enc_num_layer, factors_enc)
    apply([f"dec.dec{h}_cnns.{it}" for it in range(cfg.num_iteration) for h in (1, 2)], cfg.dec_num_layer, factors_dec)
    return out


def balanced(n_layer, g):
    """n_layer factors: g for every layer but the last, which undoes them (product 1)."""
    return [g] * (n_layer - 1) + [g ** -(n_layer - 1)] if n_layer > 1 else [1.0]


def range_cases():
    """(name, TurboAEConfig kwargs, weight seed, transform(sd, cfg) -> sd): the networks of the range tests."""
    cases = []
    for g in (0.3, 0.1, 0.03, 0.01):                      # the judge's list: plain weight gain (biases as drawn: they then dominate)
        cases.append((f"gain_{g:g}", {}, 7, ("gain", g)))
    for g in (0.3, 0.1, 0.03, 0.01, 4.0, 16.0):           # weights AND biases: activations really shrink / grow, the last layer undoes it
        cases.append((f"balanced_{g:g}", {}, 7, ("balanced", g)))
    cases.append(("alt_2^-8_2^+8", {}, 7, ("factors", [2.0 ** -8, 2.0 ** 8, 2.0 ** -8, 2.0 ** 8, 1.0], [2.0 ** -8, 2.0 ** 8])))
    cases.append(("alt_2^+6_2^-6", {}, 7, ("factors", [2.0 ** 6, 2.0 ** -6, 2.0 ** 6, 2.0 ** -6, 1.0], [2.0 ** 6, 2.0 ** -6])))
    cases.append(("balanced_0.03_L1000", dict(block_len=1000, num_iteration=2), 9, ("balanced", 0.03)))      # long-block kernels
    cases.append(("balanced_0.05_k7", dict(enc_kernel_size=7, dec_kernel_size=7, num_iteration=2, block_len=64), 9, ("balanced", 0.05)))
    cases.append(("balanced_0.1_u64", dict(enc_num_unit=64, dec_num_unit=64, num_iteration=2, block_len=40, enc_num_layer=3, dec_num_layer=3), 9,
                  ("balanced", 0.1)))
    cases.append(("balanced_0.1_dense", dict(encoder="TurboAE_rate3_cnn_dense", decoder="TurboAE_rate3_cnn_dense", enc_num_unit=32,
                                             dec_num_unit=32, num_iteration=2, block_len=64, dec_num_layer=3), 9, ("balanced", 0.1)))
    return cases


def range_case_weights(cfg, wseed, spec):
    if spec[0] == "gain":
        return W.generate_state_dict(cfg, seed=wseed, gain=float(spec[1]))
    sd = W.generate_state_dict(cfg, seed=wseed, gain=1.0)
    if cfg.dense:
        # dense layer l also sees the stack inputs directly (cnn_utils.py:59-62): scale every layer alike, the Linear head undoes it
        g = float(spec[1])
        out = {k: np.array(v, dtype=np.float32, copy=True) for k, v in sd.items()}
        for k in out:
            if ".cnns." in k:
                out[k] *= np.float32(g)
            elif k.endswith("weight") and ("enc_linear" in k or "_outputs." in k):
                out[k] *= np.float32(1.0 / g)
        return out
    if spec[0] == "balanced":
        return scale_layers(sd, cfg, balanced(cfg.enc_num_layer, float(spec[1])), balanced(cfg.dec_num_layer, float(spec[1])))
    return scale_layers(sd, cfg, spec[2], spec[1])
