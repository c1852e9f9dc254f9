"""ctypes binding of libturboae_hip.so (C ABI: include/turboae_hip.h).

There is deliberately NO fallback: if the shared library is missing or a call fails, an exception
is raised.  The product path never routes through ``oracle/`` or any CPU implementation.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TAE_LIB", os.path.join(_HERE, "lib", "libturboae_hip.so"))   # TAE_LIB: kernel-variant experiments

TAE_ABI_VERSION = 12


class TaeConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "struct_size", "block_len", "enc_num_layer", "enc_num_unit", "enc_kernel_size",
        "dec_num_layer", "dec_num_unit", "dec_kernel_size", "num_iteration", "num_iter_ft",
        "extrinsic", "enc_act", "max_batch", "dec_type", "enc_type", "dense", "precision", "dec_act", "enc_rnn", "dec_rnn",
        "range_calibration", "range_fallback")]


class TaeChannelOpts(C.Structure):
    _fields_ = [("struct_size", C.c_int32), ("norm_mode", C.c_int32), ("mean", C.c_float), ("std", C.c_float),
                ("ste", C.c_int32), ("enc_value_limit", C.c_float), ("enc_quantize_level", C.c_float),
                ("enc_truncate_limit", C.c_float), ("channel", C.c_int32), ("rec_quantize", C.c_int32),
                ("rec_quantize_limit", C.c_float), ("rec_quantize_level", C.c_float)]


class TaeNoiseOpts(C.Structure):
    _fields_ = [("struct_size", C.c_int32), ("kind", C.c_int32), ("vv", C.c_float), ("radar_prob", C.c_float),
                ("radar_power", C.c_float), ("p_gg", C.c_float), ("p_bb", C.c_float)]


RANGE_HIGH, RANGE_LOW, RANGE_FELL_BACK = 1, 2, 4        # TAE_RANGE_* bits of tae_range_status

NOISE_KIND = {"awgn": 0, "t-dist": 1, "radar": 2, "ge_awgn": 3, "bec": 4, "bsc": 5, "ge": 6, "fading": 7}      # TAE_NOISE_*


# name -> (restype, argtypes); every symbol declared in include/turboae_hip.h
_P = C.c_void_p
SIGNATURES = {
    "tae_abi_version": (C.c_int, []),
    "tae_last_error": (C.c_char_p, []),
    "tae_num_weights": (C.c_size_t, [C.POINTER(TaeConfig)]),
    "tae_create": (C.c_int, [C.POINTER(TaeConfig), _P, C.c_size_t, C.POINTER(_P)]),
    "tae_destroy": (C.c_int, [_P]),
    "tae_reserve": (C.c_int, [_P, C.c_int32]),
    "tae_set_interleaver": (C.c_int, [_P, _P, C.c_int32]),
    "tae_set_channel_opts": (C.c_int, [_P, C.POINTER(TaeChannelOpts)]),
    "tae_forward": (C.c_int, [_P, _P, _P, _P, _P, C.c_int32, _P]),
    "tae_encode": (C.c_int, [_P, _P, _P, C.c_int32, _P]),
    "tae_encode_prenorm": (C.c_int, [_P, _P, _P, _P, C.c_int32, _P]),
    "tae_normalize": (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_int32, _P]),
    "tae_decode": (C.c_int, [_P, _P, _P, C.c_int32, _P]),
    "tae_decode_taps": (C.c_int, [_P, _P, _P, _P, C.c_int32, _P]),
    "tae_count_errors": (C.c_int, [_P, _P, _P, C.c_int32, _P, _P]),
    "tae_generate_inputs": (C.c_int, [_P, _P, _P, C.c_int32, C.c_int64, C.c_uint64, C.c_uint64, C.c_float, _P]),
    "tae_generate_noise": (C.c_int, [_P, C.POINTER(TaeNoiseOpts), C.c_float, _P, _P, C.c_int32, C.c_int64, C.c_uint64, _P]),
    "tae_set_noise_opts": (C.c_int, [_P, C.POINTER(TaeNoiseOpts)]),
    "tae_probe_mfma_f16": (C.c_int, [C.c_int32, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "tae_eval_snr": (C.c_int, [_P, C.c_float, C.c_int32, C.c_int32, C.c_int64, C.c_uint64, C.c_uint64, _P, _P]),
    "tae_kernel_info": (C.c_int, [_P, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "tae_overrides": (C.c_int, [_P, C.c_char_p, C.c_int32]),
    "tae_kernel_variants": (C.c_int, [_P, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "tae_range_status": (C.c_int, [_P, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "tae_calibrate_range": (C.c_int, [_P, _P, _P, C.c_int32]),
    "tae_range_info": (C.c_int, [_P, C.POINTER(C.c_int32), C.POINTER(C.c_int32), _P, C.c_int32, C.POINTER(C.c_int32)]),
    "tae_debug_split_f16": (C.c_int, [_P, C.c_size_t, C.c_float, _P, _P]),
}

_lib: Optional[C.CDLL] = None


class TurboAEError(RuntimeError):
    pass


def load() -> C.CDLL:
    """Load the shared library (once).  Raises if it has not been built (``__graft_entry__.build()``
    or ``make -C turboae_amd/csrc``)."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch-ROCm wheels bundle their own libamdhip64; it must be the first HIP runtime mapped into
    # the process so that this library's NEEDED libamdhip64.so.7 resolves to the same copy (two HIP
    # runtimes in one process see no devices).  torch is the host plumbing here anyway.
    import torch  # noqa: F401
    if not os.path.isfile(LIB_PATH):
        raise TurboAEError(f"{LIB_PATH} not found: build it with `make -C turboae_amd/csrc` "
                           "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    if lib.tae_abi_version() != TAE_ABI_VERSION:
        raise TurboAEError("libturboae_hip.so ABI version mismatch")
    _lib = lib
    return lib


def check(rc: int) -> None:
    if rc != 0:
        msg = load().tae_last_error()
        raise TurboAEError(f"libturboae_hip error {rc}: {msg.decode() if msg else '?'}")
