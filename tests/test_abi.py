"""The C-ABI library loads and exports every symbol include/turboae_hip.h declares.
No compute calls here (no GPU in the CPU test tier)."""
import ctypes as C
import os
import re

import pytest

from turboae_amd import _lib, TurboAEConfig, weights as W

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    with open(os.path.join(ROOT, "include", "turboae_hip.h")) as fh:
        text = fh.read()
    return sorted(set(re.findall(r"TAE_API\s+[\w\s\*]+?\b(tae_\w+)\s*\(", text)))


def test_header_declares_expected_entry_points():
    syms = header_symbols()
    assert "tae_forward" in syms and "tae_decode" in syms and "tae_create" in syms
    assert set(syms) == set(_lib.SIGNATURES), set(syms) ^ set(_lib.SIGNATURES)


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    for name in header_symbols():
        assert hasattr(lib, name), name
    assert lib.tae_abi_version() == _lib.TAE_ABI_VERSION


def test_num_weights_matches_python_side():
    lib = _lib.load()
    for over in (dict(), dict(enc_num_layer=5), dict(enc_num_unit=32, dec_num_unit=32, num_iteration=2, num_iter_ft=3),
                 dict(decoder="TurboAE_rate3_rnn"), dict(decoder="TurboAE_rate3_rnn", num_iteration=2, num_iter_ft=3)):
        cfg = TurboAEConfig(**over)
        c = _lib.TaeConfig(C.sizeof(_lib.TaeConfig), cfg.block_len, cfg.enc_num_layer, cfg.enc_num_unit, 5,
                           cfg.dec_num_layer, cfg.dec_num_unit, 5, cfg.num_iteration, cfg.num_iter_ft, 1, 0, 1,
                           1 if cfg.decoder == "TurboAE_rate3_rnn" else 0)
        assert lib.tae_num_weights(C.byref(c)) == W.num_params(cfg)


def test_bad_config_is_rejected_with_message():
    lib = _lib.load()
    c = _lib.TaeConfig(C.sizeof(_lib.TaeConfig), 100, 2, 48, 5, 5, 48, 5, 6, 5, 1, 0, 1, 0)
    assert lib.tae_num_weights(C.byref(c)) == 0
    assert b"channel width" in lib.tae_last_error()
    c = _lib.TaeConfig(4, 100, 2, 100, 5, 5, 100, 5, 6, 5, 1, 0, 1, 0)
    assert lib.tae_num_weights(C.byref(c)) == 0
    assert b"struct_size" in lib.tae_last_error()


def test_product_path_has_no_cpu_fallback():
    """Without a GPU the product object must fail loudly, never fall back to the oracle."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from turboae_amd import Channel_AE_HIP
    cfg = TurboAEConfig(enc_num_unit=32, dec_num_unit=32)
    with pytest.raises(_lib.TurboAEError):
        Channel_AE_HIP(cfg, W.generate_state_dict(cfg, 1))


def test_product_package_does_not_import_oracle():
    pkg = os.path.join(ROOT, "turboae_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                with open(os.path.join(dirpath, f)) as fh:
                    text = fh.read()
                assert "import oracle" not in text and "from oracle" not in text, f
