"""turboae_amd - MI355X-native (gfx950) TurboAE rate-1/3 CNN inference hot path.

Scope: ``Channel_AE.forward`` = ENC_interCNN -> power_constraint -> AWGN ->
DEC_LargeCNN of yihanjiang/turboae as hand-written HIP kernels behind a C ABI
(include/turboae_hip.h), plus the host-side mirror of the reference interface.
"""
from .config import TurboAEConfig
from .interleaver import rand_interleaver
from . import weights, philox

__all__ = ["TurboAEConfig", "rand_interleaver", "weights", "philox", "Channel_AE_HIP"]


def __getattr__(name):
    # torch / ctypes are only needed for the GPU object; keep `import turboae_amd` light.
    if name == "Channel_AE_HIP":
        from .channel_ae import Channel_AE_HIP
        return Channel_AE_HIP
    raise AttributeError(name)
