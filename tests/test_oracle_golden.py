"""The oracle restatement vs the golden vectors produced by the REAL reference
(oracle/make_golden.py).  Runs everywhere (no reference, no GPU)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import turboae_oracle as O
from turboae_amd import TurboAEConfig, philox, weights as W, rand_interleaver

GOLD = os.path.join(os.path.dirname(__file__), "golden")
with open(os.path.join(GOLD, "MANIFEST.json")) as fh:
    MANIFEST = json.load(fh)

# CPU time: L=1000 case is ~10x the others; keep all, they finish in seconds.
CASES = sorted(MANIFEST["cases"])


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_reference_fixture(name):
    meta = MANIFEST["cases"][name]
    cfg = TurboAEConfig(**meta["config"])
    sd = W.golden_state_dict(cfg, meta)
    g = np.load(os.path.join(GOLD, name + ".npz"))
    taps, state = {}, {}
    fading = torch.from_numpy(g["fading"]) if "fading" in g.files else None
    ocfg = cfg.to_dict()
    if meta.get("is_interleave", 1) == 0:          # -is_interleave 0: the identity permutation (main.py:129-131)
        ocfg["p_array"] = np.arange(cfg.block_len)
    x, c = O.channel_ae_forward(torch.from_numpy(g["u"]), torch.from_numpy(g["noise"]), O.to_torch(sd), ocfg, taps, state, fading)
    # the reference itself is not bit-deterministic across thread counts: <= 5e-7 / 6e-8
    assert np.abs(c.numpy() - g["codes"]).max() <= 2e-6
    assert np.abs(x.numpy() - g["x_dec"]).max() <= 5e-6
    assert np.abs(taps["x_tx"].numpy() - g["x_tx"]).max() <= 2e-6
    assert abs(float(taps["mean"]) - float(g["mean"])) <= 1e-6
    assert abs(float(taps["std"]) - float(g["std"])) <= 1e-6
    if "u2" in g.files:      # --precompute_norm_stats: the second call uses the running averages
        x2, c2 = O.channel_ae_forward(torch.from_numpy(g["u2"]), torch.from_numpy(g["noise2"]), O.to_torch(sd), cfg.to_dict(), None, state)
        assert np.abs(c2.numpy() - g["codes2"]).max() <= 2e-6
        assert np.abs(x2.numpy() - g["x_dec2"]).max() <= 5e-6
    if "dec_taps" in g.files:
        # per-stage taps captured from the REAL reference (hooks on dec{1,2}_outputs, oracle/make_golden.py::reference_taps):
        # `prior` after every iteration = deinterleave of the dec2 tap (decoders.py:244-249)
        p = torch.from_numpy(O.rand_interleaver(cfg.block_len, 0))
        assert g["dec_taps"].shape == (2 * cfg.num_iteration - 1, meta["B"], cfg.block_len, cfg.num_iter_ft)
        for it in range(cfg.num_iteration - 1):
            ref_prior = O.deinterleave(torch.from_numpy(g["dec_taps"][2 * it + 1]), p)
            assert float((ref_prior - taps[f"prior_{it}"]).abs().max()) <= 5e-6, it


def test_some_fixtures_carry_per_stage_taps():
    with_taps = [n for n in CASES if "dec_taps" in np.load(os.path.join(GOLD, n + ".npz")).files]
    assert len(with_taps) >= 4 and "fwd_enc2dec5_u100_L100_b4" in with_taps and "fwd_u100_L1000_b2" in with_taps


@pytest.mark.parametrize("L", [40, 64, 100, 150, 1000])
def test_interleaver_matches_reference(L):
    p = np.load(os.path.join(GOLD, f"interleaver_L{L}_seed0.npy")).astype(np.int64)
    assert np.array_equal(p, O.rand_interleaver(L, 0))
    assert np.array_equal(p, rand_interleaver(L, 0))
    assert sorted(p.tolist()) == list(range(L))


def test_interleaver_known_prefix():
    # SURVEY.md F7 [measured against the reference]
    assert rand_interleaver(100, 0)[:12].tolist() == [26, 86, 2, 55, 75, 93, 16, 73, 54, 95, 53, 92]
    assert rand_interleaver(1000, 0)[:12].tolist() == [993, 859, 298, 553, 672, 971, 27, 231, 306, 706, 496, 558]


def test_interleave_deinterleave_roundtrip():
    p = torch.from_numpy(O.rand_interleaver(100, 0))
    x = torch.randn(3, 100, 5)
    assert torch.equal(O.deinterleave(O.interleave(x, p), p), x)
    y = O.interleave(x, p)
    assert torch.equal(y[:, 7, :], x[:, int(p[7]), :])


def test_metrics_definitions():
    u = torch.tensor([[[0.], [1.], [1.], [0.]], [[1.], [1.], [0.], [0.]]])
    xh = torch.tensor([[[0.2], [0.7], [0.5], [0.1]], [[0.9], [0.6], [0.4], [0.49]]])   # 0.5 rounds to 0 (half-to-even)
    assert O.errors_ber(u, xh) == pytest.approx(1 / 8)
    assert O.errors_bler(u, xh) == pytest.approx(0.5)
    assert O.error_counts(u, xh) == (1, 1)
    assert O.snr_db2sigma(2.0) == pytest.approx(0.7943282347)


def test_power_constraint_is_global_unbiased():
    x = torch.randn(4, 10, 3)
    y, m, s = O.power_constraint(x)
    assert float(y.mean()) == pytest.approx(0.0, abs=1e-6)
    assert float(y.std()) == pytest.approx(1.0, abs=1e-5)
    n = x.numel()
    assert float(s) == pytest.approx(float(torch.sqrt(((x - x.mean()) ** 2).sum() / (n - 1))), rel=1e-6)


def _trained():
    g = np.load(os.path.join(GOLD, "trained_enc2dec5_u100.npz"))
    meta = MANIFEST["trained"]
    cfg = TurboAEConfig(**meta["config"])
    sd = W.unpack_blob(cfg, g["weights_fp16"].astype(np.float32))
    return g, meta, cfg, sd


def test_oracle_matches_reference_on_trained_weights():
    """Short-trained REAL reference model (oracle/train_fixture.py): BER-meaningful decisions."""
    from turboae_amd import philox
    g, meta, cfg, sd = _trained()
    B, L = 64, cfg.block_len        # first 64 of the fixture's 200 blocks (CPU time)
    u = philox.random_bits(meta["input_seed"], 0, meta["B"] * L).reshape(meta["B"], L, 1)
    noise = (np.float32(O.snr_db2sigma(meta["snr_db"])) *
             philox.random_normal(meta["input_seed"], 0, meta["B"] * L * 3)).reshape(meta["B"], L, 3).astype(np.float32)
    # power_constraint couples the whole batch (encoders.py:107-108): encode all 200, decode the first 64
    w = O.to_torch(sd)
    with torch.no_grad():
        p = torch.from_numpy(O.rand_interleaver(L, 0))
        codes = O.encode(torch.from_numpy(u), w, p, cfg.enc_num_layer)
        x = O.decode(codes[:B] + torch.from_numpy(noise[:B]), w, p, cfg.dec_num_layer, cfg.num_iteration, cfg.num_iter_ft)
    assert np.abs(codes.numpy()[:8] - g["codes_first8"]).max() <= 2e-6
    assert np.abs(x.numpy()[:8] - g["x_dec_first8"]).max() <= 5e-6
    hard_ref = np.unpackbits(g["hard_bits"])[: meta["B"] * L].reshape(meta["B"], L)[:B]
    assert np.array_equal((x.numpy()[:, :, 0] > 0.5).astype(np.uint8), hard_ref)
    assert meta["ber"] == pytest.approx(1.44e-2, rel=0.05)


@pytest.mark.parametrize("kind,fname", [("trained_fp32", "trained_enc2dec5_u100_fp32.npz"),
                                        ("trained_enc5dec5_fp32", "trained_enc5dec5_u100_fp32.npz"),
                                        ("trained_cnn_gru_fp32", "trained_cnn_gru_u100_fp32.npz"),
                                        ("trained_cnn_lstm_fp32", "trained_cnn_lstm_u100_fp32.npz")])
def test_oracle_matches_reference_on_full_precision_trained_weights(kind, fname):
    """Reference-trained fp32 checkpoints (weights NOT rounded to fp16; oracle/make_golden.py::trained_fp32) of BASELINE's three
    trained shapes - enc2/dec5, enc5/dec5 (configs[2]), CNN encoder + GRU decoder (configs[4]) - and of the LSTM decoder: batch 0 at each SNR."""
    from turboae_amd import philox
    g = np.load(os.path.join(GOLD, fname))
    meta = MANIFEST[kind]
    cfg = TurboAEConfig(**meta["config"])
    rnn = cfg.decoder == "TurboAE_rate3_rnn"
    sd = W.unpack_blob(cfg, g["weights_fp32"])
    B, L, n = meta["batch"], cfg.block_len, 8 if rnn else 24        # encode the whole batch (power_constraint couples it), decode the first n blocks
    w = O.to_torch(sd)
    p = torch.from_numpy(O.rand_interleaver(L, 0))
    u = philox.random_bits(meta["input_seed"], 0, B * L).reshape(B, L, 1)
    z = philox.random_normal(meta["input_seed"], 0, B * L * 3).reshape(B, L, 3)
    with torch.no_grad():
        codes = O.encode(torch.from_numpy(u), w, p, cfg.enc_num_layer)
    assert np.abs(codes.numpy() - g["codes_batch0"]).max() <= 2e-6
    for snr in meta["snrs"]:
        key = f"{snr:g}dB"
        noise = (np.float32(O.snr_db2sigma(snr)) * z).astype(np.float32)
        taps = {}
        rx = codes[:n] + torch.from_numpy(noise[:n])
        with torch.no_grad():
            if rnn:
                x = O.decode_rnn(rx, w, p, cfg.dec_num_unit, cfg.num_iteration, cfg.num_iter_ft, 1, cell=cfg.dec_rnn)
            else:
                x = O.decode(rx, w, p, cfg.dec_num_layer, cfg.num_iteration, cfg.num_iter_ft, 1, taps)
        xr = g[f"x_dec_batch0_{key}"][:n]
        assert np.abs(x.numpy() - xr).max() <= 5e-6        # r06: the recurrent oracle too (measured <= 4e-7 against the reference)
        hard_ref = np.unpackbits(g[f"hard_bits_{key}"])[: n * L].reshape(n, L)
        flips = (x.numpy()[:, :, 0] > 0.5).astype(np.uint8) != hard_ref
        assert np.all(np.abs(xr[:, :, 0][flips] - 0.5) < 1e-4) and flips.sum() <= (1 if rnn else 0)
        if snr == meta["snrs"][0] and not rnn:
            for it in range(cfg.num_iteration - 1):
                ref_prior = O.deinterleave(torch.from_numpy(g["dec_taps_first4"][2 * it + 1]), p)
                assert float((ref_prior - taps[f"prior_{it}"][:4]).abs().max()) <= 5e-6
    assert meta["ber"]["6dB"] < 0.1 * meta["ber"]["2dB"] and meta["ber"]["2dB"] < 2e-2


# ---- randomised pin: oracle == REAL reference on the configuration space the GPU fuzz walks (oracle/fuzz_vs_reference.py)
with open(os.path.join(GOLD, "oracle_fuzz_vs_reference.json")) as _fh:
    FUZZ_REF = json.load(_fh)


def test_recorded_random_configurations_matched_the_reference():
    cases = FUZZ_REF["cases"]
    assert len(cases) >= 90
    live = [c for c in cases if not c["degenerate"]]
    assert len(live) >= 0.8 * len(cases)
    for c in cases:
        if c["degenerate"]:
            assert c["same_nonfinite_pattern"], c["config"]        # std = 0: the reference and the oracle produce the same non-finite outputs
            continue
        assert c["max_abs_codes"] <= 3e-6 * c["amplify"] and c["max_abs_x_dec"] <= 2e-6 * c["amplify"], c
        assert c["decision_flips"] == 0, c
    # the draw really covered the space
    cfgs = [c["config"] for c in cases]
    assert {k for c in cfgs for k in (c.get("enc_kernel_size", 5), c.get("dec_kernel_size", 5))} == {1, 3, 5, 7, 9}
    assert {c.get("enc_act", "elu") for c in cfgs} == {"elu", "linear", "tanh", "relu", "selu", "sigmoid"}
    assert {"TurboAE_rate3_rnn", "TurboAE_rate3_cnn_dense"} <= {c.get("decoder", "TurboAE_rate3_cnn") for c in cfgs}
    assert any(c.get("encoder") == "TurboAE_rate3_rnn" for c in cfgs)
    assert min(c["block_len"] for c in cfgs) == 1 and max(c["block_len"] for c in cfgs) > 320
    assert any(c.get("extrinsic") == 0 for c in cfgs) and {c["num_iter_ft"] for c in cfgs} == {1, 2, 3, 4, 5, 6}


def test_recorded_random_channel_option_combinations_matched_the_reference():
    """40 random combinations of block_norm_ste levels / truncation / --no_code_norm / channel branch / --rec_quantize: a code symbol
    on a quantiser threshold may differ (counted); wherever a block's codes agree its decoder output must."""
    cases = FUZZ_REF["channel_cases"]
    assert len(cases) >= 40
    for c in cases:
        if c["degenerate"]:
            assert c["same_nonfinite_pattern"]
            continue
        assert c["code_symbols_off"] <= 2e-3 * c["code_symbols"] + 1e-9, c
        assert c["x_dec_mismatch_outside_those_blocks"] == 0 and c["max_abs_x_dec_on_agreeing_blocks"] <= 5e-6 * c["amplify"], c
    cfgs = [c["config"] for c in cases]
    assert {c["channel"] for c in cfgs} == {"awgn", "bec", "bsc", "fading", "t-dist", "ge_awgn", "radar", "ge"}
    assert any(c["rec_quantize"] for c in cfgs) and any(c["no_code_norm"] for c in cfgs) and any(c["enc_truncate_limit"] > 0 for c in cfgs)
    assert {c["enc_quantize_level"] for c in cfgs if c["train_channel_mode"] == "block_norm_ste"} == {2.0, 4.0, 8.0}


def test_recorded_random_generic_configurations_matched_the_reference():
    """r03: 28 random configurations outside the MFMA kernels' envelope (what the generic fp32 kernels run)"""
    cases = FUZZ_REF["generic_cases"]
    assert len(cases) >= 28
    for c in cases:
        if c["degenerate"]:
            assert c["same_nonfinite_pattern"], c["config"]
            continue
        assert c["max_abs_codes"] <= 3e-6 * c["amplify"] and c["max_abs_x_dec"] <= 2e-6 * c["amplify"], c
        assert c["decision_flips"] == 0, c
    cfgs = [c["config"] for c in cases]
    assert {"lstm", "rnn"} <= {c.get("dec_rnn", "gru") for c in cfgs} and {"lstm", "rnn", "gru"} <= {c.get("enc_rnn", "gru") for c in cfgs if c.get("encoder") == "TurboAE_rate3_rnn"}
    assert any(c.get("encoder") == "TurboAE_rate3_rnn" and c.get("decoder", "TurboAE_rate3_cnn") == "TurboAE_rate3_cnn" for c in cfgs)
    assert max(c["dec_num_unit"] for c in cfgs) > 100 and max(c["num_iter_ft"] for c in cfgs) > 6 and max(c.get("dec_kernel_size", 5) for c in cfgs) > 9


def _digest(a):
    a = np.asarray(a, dtype=np.float64).reshape(-1)
    sign = 1.0 - 2.0 * philox.random_bits(424242, 0, a.size).astype(np.float64)
    return float(a.sum()), float((a * sign).sum()), float(np.abs(a).max())


_SMALL = ([c for c in FUZZ_REF["cases"] if not c["degenerate"] and c["bits"] <= 3000][:40]
          + [c for c in FUZZ_REF.get("generic_cases", []) if not c["degenerate"] and c["bits"] <= 700][:14])


@pytest.mark.parametrize("case", _SMALL, ids=lambda c: "L{}_B{}_w{}".format(c["config"]["block_len"], c["B"], c["weight_seed"]))
def test_oracle_reproduces_the_reference_digest_of_a_random_configuration(case):
    """Re-run, without the reference: same generated weights, same Philox inputs; the oracle's outputs must land on the three-number
    digest (sum, +-1 projection, max |.|) the reference's outputs had in the build container."""
    torch.set_num_threads(4)
    cfg = TurboAEConfig(**case["config"])
    B, L, wseed = case["B"], cfg.block_len, case["weight_seed"]
    sd = W.generate_state_dict(cfg, seed=wseed, gain=1.0)
    u = philox.random_bits(wseed, 0, B * L).reshape(B, L, 1)
    noise = (np.float32(O.snr_db2sigma(1.0)) * philox.random_normal(wseed, 0, B * L * 3)).reshape(B, L, 3).astype(np.float32)
    x, c = O.channel_ae_forward(torch.from_numpy(u), torch.from_numpy(noise), O.to_torch(sd), cfg.to_dict())
    for name, got, tol in (("x_dec", x.numpy(), 2e-6), ("codes", c.numpy(), 3e-6)):
        ref = case["reference_digest"][name]
        s, pr, mx = _digest(got)
        n, e = ref["n"], tol * case["amplify"]
        assert got.size == n
        assert abs(s - ref["sum"]) <= n * e and abs(pr - ref["proj"]) <= n * e and abs(mx - ref["max_abs"]) <= e + 1e-6 * ref["max_abs"], (name, s, pr, mx, ref)
