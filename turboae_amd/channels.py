"""HOST MIRROR (numpy) of the library's test-time channel-noise generator - test support and documentation of the draw, not the
product path: the eval sweep (``evaluate.test``) and ``Channel_AE_HIP.generate_noise`` draw on the DEVICE through the C ABI
(``tae_generate_noise``, ``gen_noise_kernel`` / ``gen_noise_chain_kernel`` in csrc/turboae_kernels.hip).

Both restate ``generate_noise(noise_shape, args, test_sigma=...)`` of the reference (``channels.py:7-109``, the
``test_sigma != 'default'`` branches used by ``trainer.test``, ``trainer.py:167-169``) and the Rayleigh fading coefficients that
``Channel_AE.forward`` draws for ``-channel fading`` (``channel_ae.py:51-56``).  The reference uses the unseeded global numpy /
torch streams; here every value is a function of ``(seed, global element index e = ((first_block + b) * L + t) * 3 + c)`` on
named Philox4x32-10 streams (``philox.py``), so any shard of any batch can be drawn on any rank, on the host or on the GPU:

    awgn      sigma * N(NOISE, e)                                                       channels.py:37-38
    t-dist    sigma * f32( sqrt((vv-2)/vv) * N(NOISE, e) / sqrt(chi2_vv(e) / vv) )      channels.py:40-41  (standard_t = z / sqrt(chi2/vv))
    radar     sigma * N(NOISE, e) + [U(MASK, e) < radar_prob] * f32(radar_power * N(AUX_A, e))      channels.py:43-49
    bec, bsc  1 if U(MASK, e) >= test_sigma else 0                                      channels.py:51-57
    ge        chain state good: 1; bad: 1 if U(MASK, e) < test_sigma else 0             channels.py:84-107
    ge_awgn   (good ? sigma(snr + 1 dB) : sigma(snr - 1 dB)) * N(NOISE, e)              channels.py:58-82
    fading    noise = sigma * N(NOISE, e); fading_h = f32( sqrt(N(AUX_A, e)^2 + N(AUX_B, e)^2) / sqrt(3.14 / 2) )   channel_ae.py:53
    chain     per (block, code symbol) along time, starts good; next state good with probability p_gg from the good state and
              with probability p_bb from the BAD state too (``good = np.random.random() < p_bb``, channels.py:79,105 - the comment
              there says "stay in bad state", the code returns to good): good' = U(CHAIN, e) < (good ? p_gg : p_bb)

N = Box-Muller normal evaluated in fp64, U = 24-bit uniform in (0, 1).  Parity of the *path* for every channel is pinned with
explicit noise tensors in tests/golden; the *distributions* are pinned to statistics of the reference's own generate_noise
(tests/golden/channel_stats.json) on the host (tests/test_channels_cpu.py) and on the device (tests/test_gpu_channels.py), and the
device draw is compared value by value with this mirror.
"""
from __future__ import annotations

import math

import numpy as np

from . import philox
from .config import TurboAEConfig

ADDITIVE = ("awgn", "t-dist", "radar", "ge_awgn", "fading")
MASKS = ("bec", "bsc", "ge")
P_GG, P_BB = 0.8, 0.8          # channels.py:60-61,86-87 (hard-coded in the reference)


def snr_db2sigma(snr_db: float) -> float:
    return 10.0 ** (-snr_db / 20.0)          # utils.py:69-70


def snr_sigma2db(sigma: float) -> float:
    return -20.0 * math.log10(sigma)           # utils.py:72-73


def _good_states(B: int, L: int, seed: int, e0: int, p_gg: float, p_bb: float) -> np.ndarray:
    """Gilbert-Elliott state of every element, (B, L, 3) bool (True = good): the walk of gen_noise_chain_kernel."""
    u = philox.random_unit(seed, philox.STREAM_CHAIN, e0, B * L * 3).reshape(B, L, 3)
    pg, pb = float(np.float32(p_gg)), float(np.float32(p_bb))
    good = np.ones((B, 3), dtype=bool)
    out = np.empty((B, L, 3), dtype=bool)
    for t in range(L):
        out[:, t, :] = good
        good = np.where(good, u[:, t, :] < pg, u[:, t, :] < pb)
    return out


def generate_noise(shape, cfg: TurboAEConfig, test_sigma: float, seed: int = 0, first_block: int = 0) -> np.ndarray:
    """float32 (B, L, 3) noise of ``cfg.channel`` for the blocks ``first_block .. first_block + B``.  For the additive channels
    ``test_sigma`` is the SNR in dB, for bec / bsc / ge the erase / flip probability (channels.py:28-31)."""
    B, L, C = shape
    if C != 3:
        raise ValueError("the rate-1/3 code has 3 code symbols per position")
    ch = cfg.channel
    n, e0 = B * L * 3, first_block * L * 3
    f32 = np.float32
    if ch in ("bec", "bsc"):
        return (philox.random_unit(seed, philox.STREAM_MASK, e0, n) >= float(f32(test_sigma))).astype(np.float32).reshape(shape)
    if ch == "ge":
        good = _good_states(B, L, seed, e0, P_GG, P_BB)
        keep_bad = (philox.random_unit(seed, philox.STREAM_MASK, e0, n) < float(f32(test_sigma))).reshape(shape)
        return np.where(good, 1.0, keep_bad.astype(np.float64)).astype(np.float32)
    sigma = f32(snr_db2sigma(float(f32(test_sigma))))
    z = philox.random_normal64(seed, e0, n, philox.STREAM_NOISE)
    if ch in ("awgn", "fading"):
        return (sigma * z.astype(np.float32)).reshape(shape)
    if ch == "t-dist":
        vv = float(f32(cfg.vv))
        t = z / np.sqrt(philox.chi_square(seed, e0, n, vv) / vv)
        return (sigma * (math.sqrt((vv - 2.0) / vv) * t).astype(np.float32)).reshape(shape)
    if ch == "radar":
        hit = philox.random_unit(seed, philox.STREAM_MASK, e0, n) < float(f32(cfg.radar_prob))
        burst = (float(f32(cfg.radar_power)) * philox.random_normal64(seed, e0, n, philox.STREAM_AUX_A)).astype(np.float32)
        base = sigma * z.astype(np.float32)
        return np.where(hit, base + burst, base).astype(np.float32).reshape(shape)
    if ch == "ge_awgn":
        good = _good_states(B, L, seed, e0, P_GG, P_BB)
        snr_back = snr_sigma2db(snr_db2sigma(float(f32(test_sigma))))
        s_good, s_bad = f32(snr_db2sigma(snr_back + 1.0)), f32(snr_db2sigma(snr_back - 1.0))
        return (np.where(good, s_good, s_bad).astype(np.float32) * z.astype(np.float32).reshape(shape)).astype(np.float32)
    raise ValueError(f"unknown channel {ch!r}")


def rayleigh_fading(shape, seed: int = 0, first_block: int = 0) -> np.ndarray:
    """fading_h of channel_ae.py:53: sqrt(randn^2 + randn^2) / sqrt(3.14 / 2) (the reference's own constant), float32 (B, L, 3)."""
    B, L, C = shape
    n, e0 = B * L * 3, first_block * L * 3
    a = philox.random_normal64(seed, e0, n, philox.STREAM_AUX_A)
    b = philox.random_normal64(seed, e0, n, philox.STREAM_AUX_B)
    return (np.sqrt(a * a + b * b) / math.sqrt(3.14 / 2.0)).astype(np.float32).reshape(shape)
