"""Test-time channel noise, restating ``generate_noise(noise_shape, args, test_sigma=...)`` of the reference
(``channels.py:7-109``, the ``test_sigma != 'default'`` branches used by ``trainer.test``, ``trainer.py:167-169``) plus
the Rayleigh fading coefficients that ``Channel_AE.forward`` draws for ``-channel fading`` (``channel_ae.py:51-56``).

torch generators on the target device, so an eval sweep never crosses PCIe.  The draws are not bit-compatible with
the reference's numpy / torch-CPU streams (no test of the reference depends on them; BER is a statistic) - parity of
the *path* for every channel is pinned with explicit noise tensors in tests/golden.  ``awgn`` inputs for benchmarks
and parity come from the Philox generator in the library (``Channel_AE_HIP.generate_inputs``) instead.
"""
from __future__ import annotations

import math
from typing import Optional

import torch

from .config import TurboAEConfig

ADDITIVE = ("awgn", "t-dist", "radar", "ge_awgn", "fading")


def snr_db2sigma(snr_db: float) -> float:
    return 10.0 ** (-snr_db / 20.0)          # utils.py:69-70


def snr_sigma2db(sigma: float) -> float:
    return -20.0 * math.log10(sigma)           # utils.py:72-73


def _markov_good_state(shape, p_gg: float, p_bb: float, gen: Optional[torch.Generator], device) -> torch.Tensor:
    """Gilbert-Elliott state sequence along dim 1 (channels.py:60-82 / 87-107), independent chains per (block, code
    symbol), every chain starts good.  The transitions are the reference's as written: from the good state the next
    state is good with probability p_gg (``good = np.random.random() < p_gg``, channels.py:73,100), and from the BAD state
    the next state is good with probability p_bb as well (``good = np.random.random() < p_bb``, channels.py:79,105 - the
    comment there says "stay in bad state", the code returns to good).  With p_gg = p_bb = 0.8 the state is therefore
    good with probability 0.8 at every step after the first, independent of the previous state.  Returns a bool tensor
    (True = good)."""
    B, L, C = shape
    u = torch.rand((B, L, C), generator=gen, device=device)
    good = torch.ones((B, C), dtype=torch.bool, device=device)
    out = torch.empty((B, L, C), dtype=torch.bool, device=device)
    for t in range(L):
        out[:, t, :] = good
        good = torch.where(good, u[:, t, :] < p_gg, u[:, t, :] < p_bb)
    return out


def generate_noise(shape, cfg: TurboAEConfig, test_sigma: float, device="cpu", generator: Optional[torch.Generator] = None) -> torch.Tensor:
    """``generate_noise(noise_shape, args, test_sigma=test_sigma)`` (channels.py:27-109).  For the additive channels
    ``test_sigma`` is the SNR in dB; for bec / bsc / ge it is the erase / flip probability (channels.py:28-31)."""
    ch = cfg.channel
    g = generator
    if ch in ("bec", "bsc"):
        # np.random.choice([0, 1], p=[p, 1 - p]): 1 = symbol kept (channels.py:51-57)
        return (torch.rand(shape, generator=g, device=device) >= test_sigma).float()
    if ch == "ge":
        good = _markov_good_state(shape, 0.8, 0.8, g, device)                          # channels.py:84-107
        keep_bad = torch.rand(shape, generator=g, device=device) < test_sigma          # bad state: 1 with probability bsc_h = this_sigma
        return torch.where(good, torch.ones(shape, device=device), keep_bad.float())   # good state: bsc_k = 1.0 -> always 1
    sigma = snr_db2sigma(test_sigma)
    if ch in ("awgn", "fading"):
        return sigma * torch.randn(shape, generator=g, device=device)                  # channels.py:37-38
    if ch == "t-dist":
        # sqrt((vv - 2) / vv) * standard_t(vv) (channels.py:40-41); t = z / sqrt(chi2_vv / vv), chi2_vv = 2 * Gamma(vv / 2)
        vv = float(cfg.vv)
        z = torch.randn(shape, generator=g, device=device)
        gam = torch.distributions.Gamma(torch.tensor(vv / 2.0, device=device), torch.tensor(1.0, device=device))
        if g is not None:
            # torch.distributions ignores explicit generators: seed the global stream from it (deterministic per generator state)
            torch.manual_seed(int(torch.randint(0, 2 ** 31 - 1, (1,), generator=g, device=device).item()))
        chi2 = 2.0 * gam.sample(tuple(shape))
        return sigma * math.sqrt((vv - 2.0) / vv) * z / torch.sqrt(chi2 / vv)
    if ch == "radar":
        add_pos = (torch.rand(shape, generator=g, device=device) < cfg.radar_prob).float()           # channels.py:43-45
        corrupted = cfg.radar_power * torch.randn(shape, generator=g, device=device) * add_pos
        return sigma * torch.randn(shape, generator=g, device=device) + corrupted                     # channels.py:47-49
    if ch == "ge_awgn":
        good = _markov_good_state(shape, 0.8, 0.8, g, device)                                          # channels.py:58-82
        s_good = snr_db2sigma(snr_sigma2db(sigma) + 1.0)
        s_bad = snr_db2sigma(snr_sigma2db(sigma) - 1.0)
        scale = torch.where(good, torch.full(shape, s_good, device=device), torch.full(shape, s_bad, device=device))
        return scale * torch.randn(shape, generator=g, device=device)
    raise ValueError(f"unknown channel {ch!r}")


def rayleigh_fading(shape, device="cpu", generator: Optional[torch.Generator] = None) -> torch.Tensor:
    """fading_h of channel_ae.py:53: sqrt(randn^2 + randn^2) / sqrt(3.14 / 2) (the reference's own constant)."""
    a = torch.randn(shape, generator=generator, device=device)
    b = torch.randn(shape, generator=generator, device=device)
    return torch.sqrt(a * a + b * b) / math.sqrt(3.14 / 2.0)
