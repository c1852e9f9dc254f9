"""turboae_amd/channels.py (generate_noise restated, channels.py:7-109) - distribution checks on CPU."""
import math

import pytest
import torch

from turboae_amd import TurboAEConfig, channels


def _gen(seed=3):
    g = torch.Generator()
    g.manual_seed(seed)
    return g


def test_awgn_and_fading_noise_sigma():
    for ch in ("awgn", "fading"):
        n = channels.generate_noise((200, 100, 3), TurboAEConfig(channel=ch), 2.0, generator=_gen())
        assert n.shape == (200, 100, 3) and n.dtype == torch.float32
        assert float(n.std()) == pytest.approx(channels.snr_db2sigma(2.0), rel=0.02)
        assert abs(float(n.mean())) < 0.01


def test_t_dist_is_unit_variance_scaled_and_heavy_tailed():
    cfg = TurboAEConfig(channel="t-dist", vv=5.0)
    n = channels.generate_noise((400, 100, 3), cfg, 0.0, generator=_gen())
    # sqrt((vv - 2) / vv) * t_vv has unit variance (channels.py:41); sigma(0 dB) = 1
    assert float(n.var()) == pytest.approx(1.0, rel=0.08)
    kurt = float(((n / n.std()) ** 4).mean())
    assert kurt > 4.0          # Gaussian: 3; t_5: 9


def test_radar_mixture_variance():
    cfg = TurboAEConfig(channel="radar", radar_prob=0.05, radar_power=5.0)
    n = channels.generate_noise((400, 100, 3), cfg, 0.0, generator=_gen())
    assert float(n.var()) == pytest.approx(1.0 + 0.05 * 25.0, rel=0.08)


@pytest.mark.parametrize("ch", ["bec", "bsc"])
def test_erasure_flip_masks(ch):
    m = channels.generate_noise((300, 100, 3), TurboAEConfig(channel=ch), 0.2, generator=_gen())
    assert set(m.unique().tolist()) <= {0.0, 1.0}
    assert float(m.mean()) == pytest.approx(0.8, abs=0.01)      # 1 = kept with probability 1 - p (channels.py:51-57)


def test_gilbert_elliott_chains():
    # stationary distribution of the 2-state chain with p_gg = p_bb = 0.8 started in the good state: -> 1/2 good
    shape = (500, 100, 3)
    good = channels._markov_good_state(shape, 0.8, 0.8, _gen(), "cpu")
    assert bool(good[:, 0, :].all())                              # every chain starts good (channels.py:64,91)
    assert float(good[:, 50:, :].float().mean()) == pytest.approx(0.5, abs=0.02)
    stay = (good[:, 1:, :] == good[:, :-1, :]).float().mean()
    assert float(stay) == pytest.approx(0.8, abs=0.01)
    m = channels.generate_noise(shape, TurboAEConfig(channel="ge"), 0.3, generator=_gen())
    # good state always keeps (bsc_k = 1), bad state keeps with probability this_sigma (channels.py:87-88,95,99)
    assert float(m[:, 50:, :].mean()) == pytest.approx(0.5 + 0.5 * 0.3, abs=0.02)
    n = channels.generate_noise(shape, TurboAEConfig(channel="ge_awgn"), 0.0, generator=_gen())
    sg, sb = channels.snr_db2sigma(1.0), channels.snr_db2sigma(-1.0)
    assert float(n[:, 50:, :].var()) == pytest.approx(0.5 * (sg * sg + sb * sb), rel=0.05)


def test_rayleigh_fading_constant_of_the_reference():
    h = channels.rayleigh_fading((400, 100, 3), generator=_gen())
    assert float(h.min()) >= 0.0
    # E[sqrt(a^2 + b^2)] = sqrt(pi / 2); the reference divides by sqrt(3.14 / 2) (channel_ae.py:53)
    assert float(h.mean()) == pytest.approx(math.sqrt(math.pi / 2) / math.sqrt(3.14 / 2), rel=0.01)


def test_generators_are_reproducible():
    cfg = TurboAEConfig(channel="radar")
    a = channels.generate_noise((4, 100, 3), cfg, 1.0, generator=_gen(9))
    b = channels.generate_noise((4, 100, 3), cfg, 1.0, generator=_gen(9))
    assert torch.equal(a, b)
