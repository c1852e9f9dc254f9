"""Host-side fp16 hi/lo split used by the weight packer of the fp16-split kernels (no GPU needed: the C ABI exposes it)."""
import ctypes as C

import numpy as np

from turboae_amd import _lib


def _split(x, scale=1.0):
    lib = _lib.load()
    x = np.ascontiguousarray(x, dtype=np.float32)
    hi = np.empty(x.size, dtype=np.uint16)
    lo = np.empty(x.size, dtype=np.uint16)
    _lib.check(lib.tae_debug_split_f16(x.ctypes.data_as(C.c_void_p), x.size, C.c_float(scale), hi.ctypes.data_as(C.c_void_p),
                                       lo.ctypes.data_as(C.c_void_p)))
    return hi.view(np.float16), lo.view(np.float16)


def test_hi_is_round_to_nearest_even_like_numpy():
    rng = np.random.default_rng(1)
    x = np.concatenate([rng.standard_normal(20000).astype(np.float32) * np.float32(10.0) ** rng.integers(-9, 5, 20000).astype(np.float32),
                        np.array([0.0, -0.0, 1.0, 65504.0, -65504.0, 65519.9, 6.1e-5, 5.97e-8, 2.98e-8, 2.9802322e-8, 1e-10,
                                  2049.0, 2051.0, 0.333251953125 + 2.0 ** -13], dtype=np.float32)])
    hi, lo = _split(x)
    with np.errstate(over="ignore"):
        ref = x.astype(np.float16)                  # numpy: round to nearest even, denormals kept
    assert np.array_equal(hi.view(np.uint16), ref.view(np.uint16))


def test_hi_plus_lo_represents_fp32_to_22_bits():
    rng = np.random.default_rng(2)
    x = (rng.uniform(-1, 1, 50000) * 0.0775).astype(np.float32)        # the conv weights' range
    hi, lo = _split(x, scale=2.0 ** 17)                                  # max |w| * 2^17 in [2^13, 2^14)
    rec = (hi.astype(np.float64) + lo.astype(np.float64)) / 2.0 ** 17
    err = np.abs(rec - x.astype(np.float64))
    assert err.max() <= 2.0 ** -22 * 0.0775 * 1.01       # relative 2^-22 of the largest weight
    # lo is the correctly rounded residual
    res = (x.astype(np.float64) * 2.0 ** 17 - hi.astype(np.float64)).astype(np.float32)
    assert np.array_equal(lo.view(np.uint16), res.astype(np.float16).view(np.uint16))


def test_small_values_keep_an_absolute_floor():
    x = np.float32(10.0) ** np.arange(-12, 0, dtype=np.float32)
    hi, lo = _split(x)
    rec = hi.astype(np.float64) + lo.astype(np.float64)
    assert np.all(np.abs(rec - x.astype(np.float64)) <= 2.0 ** -25 + 2.0 ** -22 * x)
