"""turboae_amd/channels.py (the numpy mirror of the library's device generator tae_generate_noise = generate_noise restated,
channels.py:7-109) against statistics of the REFERENCE's own
generate_noise (tests/golden/channel_stats.json, written by oracle/make_channel_stats.py in the build container).

The draws cannot be compared (the reference uses the unseeded global numpy / torch streams); the distributions can:
each case draws the same shape as the fixture and compares moments, |x| quantiles, 0/1 fractions, the always-good first
position of the Gilbert-Elliott chains and the lag-1 correlation along time.  Tolerances are sampling errors of BOTH
samples (reference and ours), ~5 sigma."""
import json
import math
import os

import numpy as np
import pytest

from turboae_amd import TurboAEConfig, channels

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
with open(os.path.join(GOLD, "channel_stats.json")) as _fh:
    REF = json.load(_fh)["cases"]


def _draw(case, seed=3):
    cfg = TurboAEConfig(channel=case["channel"], **case["args"])
    x = channels.generate_noise(tuple(case["shape"]), cfg, case["test_sigma"], seed=seed)
    assert tuple(x.shape) == tuple(case["shape"]) and x.dtype == np.float32
    return x.astype(np.float64)


def _lag1(s):
    a, b = s[:, :-1, :].reshape(-1), s[:, 1:, :].reshape(-1)
    return float(np.corrcoef(a, b)[0, 1])


@pytest.mark.parametrize("name", [n for n, c in REF.items() if c["channel"] in ("bec", "bsc", "ge")])
def test_mask_channels_match_reference_statistics(name):
    case, st = REF[name], REF[name]["stats"]
    x = _draw(case)
    assert set(np.unique(x).tolist()) <= {0.0, 1.0}
    n = st["n"]
    p = st["frac_one"]
    tol = 5.0 * math.sqrt(2.0 * max(p * (1 - p), 1e-4) / n)
    assert float((x == 1.0).mean()) == pytest.approx(st["frac_one"], abs=tol)
    nf = n / case["shape"][1]
    pf = st["first_pos_mean"]
    assert float(x[:, 0, :].mean()) == pytest.approx(pf, abs=5.0 * math.sqrt(2.0 * max(pf * (1 - pf), 0.0) / nf) + 1e-12)
    assert float(x[:, 1:, :].mean()) == pytest.approx(st["rest_mean"], abs=tol)
    if st["var"] > 0 and x.std() > 0:
        assert _lag1(x) == pytest.approx(st["lag1"], abs=5.0 * math.sqrt(2.0 / n))


@pytest.mark.parametrize("name", [n for n, c in REF.items() if c["channel"] not in ("bec", "bsc", "ge")])
def test_additive_channels_match_reference_statistics(name):
    case, st = REF[name], REF[name]["stats"]
    x = _draw(case)
    n = st["n"]
    sd = math.sqrt(st["var"])
    assert float(x.mean()) == pytest.approx(st["mean"], abs=5.0 * sd * math.sqrt(2.0 / n))
    # quantiles of |x| are robust for every channel (t-dist with vv <= 4 has no finite kurtosis)
    for q, rel in (("abs_q50", 0.02), ("abs_q90", 0.02), ("abs_q99", 0.04)):
        assert float(np.quantile(np.abs(x), float(q[5:]) / 100.0)) == pytest.approx(st[q], rel=rel), q
    light_tailed = case["channel"] in ("awgn", "ge_awgn")
    if light_tailed:
        # var of the sample variance of a near-Gaussian: 2 sigma^4 / n (x kurt/3 margin)
        assert float(x.var()) == pytest.approx(st["var"], rel=5.0 * math.sqrt(2.0 * 2.0 * st["kurt"] / 3.0 / n))
        assert float(((x - x.mean()) ** 4).mean() / x.var() ** 2) == pytest.approx(st["kurt"], abs=0.12)
        nf = n / case["shape"][1]
        assert float(x[:, 0, :].var()) == pytest.approx(st["first_pos_var"], rel=5.0 * math.sqrt(2.0 * 2.0 / nf))
        assert float(x[:, 1:, :].var()) == pytest.approx(st["rest_var"], rel=5.0 * math.sqrt(2.0 * 2.0 * st["kurt"] / 3.0 / n))
    elif case["channel"] == "radar" or case["args"].get("vv", 0) > 4:
        assert float(x.var()) == pytest.approx(st["var"], rel=0.06)
    assert _lag1(x * x) == pytest.approx(st["lag1"], abs=5.0 * math.sqrt(2.0 / n) + 0.004)


def test_gilbert_elliott_chain_is_the_reference_chain():
    """channels.py:73,79 / 100,105: the next state is good w.p. p_gg from good and w.p. p_bb from BAD (the reference
    returns to good with 0.8; it does not stay bad with 0.8), so with 0.8 / 0.8 the state is good 80 % of the time,
    memoryless after the start.  The numbers asserted are the reference's (fixture), not this derivation."""
    shape = (400, 100, 3)
    good = channels._good_states(shape[0], shape[1], 3, 0, 0.8, 0.8)
    assert bool(good[:, 0, :].all())                              # every chain starts good (channels.py:64,91)
    ref = REF["ge_0p0"]["stats"]                                  # p = 0: the mask IS the state sequence
    assert float(good[:, 1:, :].mean()) == pytest.approx(ref["rest_mean"], abs=0.008)
    assert _lag1(good.astype(np.float64)) == pytest.approx(ref["lag1"], abs=0.02)
    # asymmetric chain: stationary good fraction = p_bb' / (1 - p_gg + p_bb') with p_bb' = P(bad -> good)
    g2 = channels._good_states(400, 200, 5, 0, 0.9, 0.3)
    assert float(g2[:, 50:, :].mean()) == pytest.approx(0.3 / (0.1 + 0.3), abs=0.01)


def test_fixture_is_the_reference_not_a_derivation():
    # the numbers VERDICT r01 quoted from running the reference generator: ge keep-fraction 0.857, ge_awgn variance 0.9
    assert REF["ge_0p3"]["stats"]["rest_mean"] == pytest.approx(0.8 + 0.2 * 0.3, abs=0.005)
    sg, sb = channels.snr_db2sigma(1.0), channels.snr_db2sigma(-1.0)
    assert REF["ge_awgn_0dB"]["stats"]["rest_var"] == pytest.approx(0.8 * sg * sg + 0.2 * sb * sb, rel=0.02)


def test_rayleigh_fading_constant_of_the_reference():
    h = channels.rayleigh_fading((400, 100, 3), seed=3)
    assert h.dtype == np.float32 and float(h.min()) >= 0.0
    # E[sqrt(a^2 + b^2)] = sqrt(pi / 2); the reference divides by sqrt(3.14 / 2) (channel_ae.py:53)
    assert float(h.mean()) == pytest.approx(math.sqrt(math.pi / 2) / math.sqrt(3.14 / 2), rel=0.01)


def test_generators_are_counter_based():
    """any shard of any batch reproduces the single-stream draw (keyed by seed and GLOBAL block index), for every channel"""
    for ch, sig in (("radar", 1.0), ("t-dist", 0.5), ("ge_awgn", 2.0), ("ge", 0.3), ("bec", 0.2), ("bsc", 0.1), ("awgn", 0.0), ("fading", 1.0)):
        cfg = TurboAEConfig(channel=ch, block_len=37)
        full = channels.generate_noise((9, 37, 3), cfg, sig, seed=9)
        assert np.array_equal(full, channels.generate_noise((9, 37, 3), cfg, sig, seed=9))
        assert np.array_equal(full[4:7], channels.generate_noise((3, 37, 3), cfg, sig, seed=9, first_block=4)), ch
        assert not np.array_equal(full, channels.generate_noise((9, 37, 3), cfg, sig, seed=10))
    h = channels.rayleigh_fading((9, 37, 3), seed=9)
    assert np.array_equal(h[2:5], channels.rayleigh_fading((3, 37, 3), seed=9, first_block=2))


def test_awgn_mirror_is_the_benchmark_noise_stream():
    """channel='awgn' of the mirror == the Philox noise tae_generate_inputs draws (philox.random_normal x sigma)"""
    from turboae_amd import philox
    cfg = TurboAEConfig(block_len=20)
    x = channels.generate_noise((5, 20, 3), cfg, 2.0, seed=11, first_block=3)
    ref = np.float32(channels.snr_db2sigma(2.0)) * philox.random_normal(11, 3 * 20 * 3, 5 * 20 * 3)
    assert np.array_equal(x.reshape(-1), ref.astype(np.float32))


def test_chi_square_sampler_moments():
    from turboae_amd import philox
    for vv in (3.0, 5.0, 12.5):
        x = philox.chi_square(4, 0, 400000, vv)
        assert x.min() > 0.0
        assert float(x.mean()) == pytest.approx(vv, rel=0.01)
        assert float(x.var()) == pytest.approx(2.0 * vv, rel=0.03)
