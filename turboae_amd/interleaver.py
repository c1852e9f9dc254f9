"""Fixed random interleaver of the reference.

``RandInterlv(length, seed).p_array`` (commpy/channelcoding/interleavers.py:77-82) is
``numpy.random.mtrand.RandomState(seed).permutation(arange(length))``; Channel_AE.forward re-creates
it with seed 0 on every call (channel_ae.py:32-36) and main.py:123-127 uses the same seed.  The
legacy MT19937 ``RandomState`` stream is frozen by numpy's compatibility policy, so calling numpy
here is the definition, not an approximation; the L=100 and L=1000 arrays are additionally pinned
as golden fixtures (tests/golden/interleaver_*.npy).
"""
from __future__ import annotations

import numpy as np


def rand_interleaver(block_len: int, seed: int = 0) -> np.ndarray:
    return np.random.mtrand.RandomState(seed).permutation(np.arange(block_len)).astype(np.int32)


def inverse(p: np.ndarray) -> np.ndarray:
    """reverse_p_array of DeInterleaver (interleavers.py:29-33): inv[p[i]] = i."""
    inv = np.empty_like(p)
    inv[p] = np.arange(p.size, dtype=p.dtype)
    return inv
