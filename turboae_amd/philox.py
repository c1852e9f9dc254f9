"""Counter-based Philox4x32-10 generator (numpy), bit-identical to the device code in
``csrc/philox.hpp``.

The reference draws its test inputs from unseeded global RNGs (``torch.randint`` /
``torch.randn`` at trainer.py:167-169, channels.py:35), so "identical random bits/noise" on
two machines needs a generator of our own.  Everything here is keyed by
``(seed, stream, element index)`` so any shard of any batch can be produced independently on
any rank, on the host or on the GPU.

Stream layout (shared with the device code):
    counter = (idx_lo, idx_hi, stream, 0)      key = (seed_lo, seed_hi)
    one call yields 4 x u32; element ``e`` of a stream is word ``e & 3`` of call ``e >> 2``.
"""
from __future__ import annotations

import numpy as np

_M0 = np.uint64(0xD2511F53)
_M1 = np.uint64(0xCD9E8D57)
_W0 = 0x9E3779B9
_W1 = 0xBB67AE85
_MASK32 = np.uint64(0xFFFFFFFF)

# stream ids (must match csrc/philox.hpp)
STREAM_BITS = 1
STREAM_NOISE = 2
STREAM_WEIGHTS = 3
# channel generators (tae_generate_noise / channels.py): keep and radar-position masks, Gilbert-Elliott state walk, second and third
# normal streams (radar bursts, fading_h), chi-square draws of the t distribution (attempt index in counter word 3)
STREAM_MASK = 4
STREAM_CHAIN = 5
STREAM_AUX_A = 6
STREAM_AUX_B = 7
STREAM_GAMMA = 8


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised Philox4x32-10.  All arguments broadcastable uint32 arrays; returns 4 uint32 arrays."""
    c0 = np.asarray(c0, dtype=np.uint64)
    c1 = np.asarray(c1, dtype=np.uint64)
    c2 = np.asarray(c2, dtype=np.uint64)
    c3 = np.asarray(c3, dtype=np.uint64)
    c0, c1, c2, c3 = np.broadcast_arrays(c0, c1, c2, c3)
    k0 = int(k0) & 0xFFFFFFFF
    k1 = int(k1) & 0xFFFFFFFF
    for r in range(10):
        p0 = _M0 * c0
        p1 = _M1 * c2
        hi0, lo0 = p0 >> np.uint64(32), p0 & _MASK32
        hi1, lo1 = p1 >> np.uint64(32), p1 & _MASK32
        n0 = hi1 ^ c1 ^ np.uint64(k0)
        n2 = hi0 ^ c3 ^ np.uint64(k1)
        c0, c1, c2, c3 = n0, lo1, n2, lo0
        k0 = (k0 + _W0) & 0xFFFFFFFF
        k1 = (k1 + _W1) & 0xFFFFFFFF
    return (c0.astype(np.uint32), c1.astype(np.uint32), c2.astype(np.uint32), c3.astype(np.uint32))


def random_u32(seed: int, stream: int, start: int, count: int) -> np.ndarray:
    """``count`` consecutive u32 words of stream ``stream`` starting at element ``start``."""
    if count <= 0:
        return np.zeros((0,), dtype=np.uint32)
    first_call = start >> 2
    last_call = (start + count - 1) >> 2
    idx = np.arange(first_call, last_call + 1, dtype=np.uint64)
    w = philox4x32_10(idx & _MASK32, idx >> np.uint64(32), np.uint64(stream), np.uint64(0),
                      seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    words = np.stack(w, axis=1).reshape(-1)
    off = start - (first_call << 2)
    return words[off:off + count]


def random_bits(seed: int, start: int, count: int, stream: int = STREAM_BITS) -> np.ndarray:
    """Bernoulli(0.5) bits as float32 {0,1}: the low bit of each u32 word."""
    return (random_u32(seed, stream, start, count) & np.uint32(1)).astype(np.float32)


def _u32_to_unit_open(x: np.ndarray) -> np.ndarray:
    """u32 -> float64 in (0,1): ((x >> 8) + 0.5) * 2^-24 (exact in fp64 and in fp32)."""
    return ((x >> np.uint32(8)).astype(np.float64) + 0.5) * (1.0 / 16777216.0)


def random_normal(seed: int, start: int, count: int, stream: int = STREAM_NOISE) -> np.ndarray:
    """Standard normals as float32 via Box-Muller evaluated in fp64 (then rounded once).

    Normal ``e`` uses words ``2*(e>>1)`` and ``2*(e>>1)+1`` of the stream: even ``e`` takes the
    cosine branch, odd ``e`` the sine branch.  fp64 evaluation keeps host and device results
    equal except for measure-~1e-8 rounding ties.
    """
    if count <= 0:
        return np.zeros((0,), dtype=np.float32)
    p0 = start >> 1
    p1 = (start + count - 1) >> 1
    w = random_u32(seed, stream, 2 * p0, 2 * (p1 - p0 + 1)).reshape(-1, 2)
    u1 = _u32_to_unit_open(w[:, 0])
    u2 = _u32_to_unit_open(w[:, 1])
    r = np.sqrt(-2.0 * np.log(u1))
    th = 2.0 * np.pi * u2
    z = np.stack([r * np.cos(th), r * np.sin(th)], axis=1).reshape(-1)
    off = start - 2 * p0
    return z[off:off + count].astype(np.float32)


def random_normal64(seed: int, start: int, count: int, stream: int = STREAM_NOISE) -> np.ndarray:
    """The same normals before the rounding to float32 (the channel generators combine them in fp64 and round once)."""
    if count <= 0:
        return np.zeros((0,), dtype=np.float64)
    p0 = start >> 1
    p1 = (start + count - 1) >> 1
    w = random_u32(seed, stream, 2 * p0, 2 * (p1 - p0 + 1)).reshape(-1, 2)
    r = np.sqrt(-2.0 * np.log(_u32_to_unit_open(w[:, 0])))
    th = 2.0 * np.pi * _u32_to_unit_open(w[:, 1])
    z = np.stack([r * np.cos(th), r * np.sin(th)], axis=1).reshape(-1)
    off = start - 2 * p0
    return z[off:off + count]


def random_unit(seed: int, stream: int, start: int, count: int) -> np.ndarray:
    """Uniforms in (0, 1) as float64 with 24 random bits each: word e of the stream -> ((w >> 8) + 0.5) * 2^-24."""
    return _u32_to_unit_open(random_u32(seed, stream, start, count))


def chi_square(seed: int, start: int, count: int, vv: float) -> np.ndarray:
    """chi-square(vv) = 2 * Gamma(vv / 2) by Marsaglia-Tsang (vv > 2, so the shape is >= 1), float64.  Attempt k of element e draws
    the Philox counter (e_lo, e_hi, STREAM_GAMMA, k): a Box-Muller normal (cosine branch) from words 0, 1 and a uniform from word 2;
    the first accepted attempt (of at most 32) counts.  Mirrors chi_square_at in csrc/turboae_kernels.hip."""
    e = np.arange(start, start + count, dtype=np.uint64)
    d = 0.5 * float(vv) - 1.0 / 3.0
    c = 1.0 / np.sqrt(9.0 * d)
    out = np.full(count, 2.0 * d, dtype=np.float64)
    pending = np.arange(count)
    for k in range(32):
        if pending.size == 0:
            break
        ee = e[pending]
        w = philox4x32_10(ee & _MASK32, ee >> np.uint64(32), np.uint64(STREAM_GAMMA), np.uint64(k), seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
        x = np.sqrt(-2.0 * np.log(_u32_to_unit_open(w[0]))) * np.cos(2.0 * np.pi * _u32_to_unit_open(w[1]))
        t = 1.0 + c * x
        v = t * t * t
        with np.errstate(invalid="ignore", divide="ignore"):
            ok = (v > 0.0) & (np.log(_u32_to_unit_open(w[2])) < 0.5 * x * x + d - d * v + d * np.log(np.where(v > 0.0, v, 1.0)))
        out[pending[ok]] = 2.0 * d * v[ok]
        pending = pending[~ok]
    return out


def random_uniform_pm1(seed: int, start: int, count: int, stream: int = STREAM_WEIGHTS) -> np.ndarray:
    """Uniform in (-1,1) as float32 (used by the portable weight generator; host only)."""
    u = _u32_to_unit_open(random_u32(seed, stream, start, count))
    return (2.0 * u - 1.0).astype(np.float32)
