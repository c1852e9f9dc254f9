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
                 dict(decoder="TurboAE_rate3_rnn"), dict(decoder="TurboAE_rate3_rnn", num_iteration=2, num_iter_ft=3),
                 dict(enc_num_unit=64, dec_num_unit=32), dict(enc_num_unit=32, decoder="TurboAE_rate3_rnn"),
                 dict(enc_kernel_size=3, dec_kernel_size=1)):
        cfg = TurboAEConfig(**over)
        c = _lib.TaeConfig(C.sizeof(_lib.TaeConfig), cfg.block_len, cfg.enc_num_layer, cfg.enc_num_unit, cfg.enc_kernel_size,
                           cfg.dec_num_layer, cfg.dec_num_unit, cfg.dec_kernel_size, cfg.num_iteration, cfg.num_iter_ft, 1, 0, 1,
                           1 if cfg.decoder == "TurboAE_rate3_rnn" else 0)
        assert lib.tae_num_weights(C.byref(c)) == W.num_params(cfg)


def test_bad_config_is_rejected_with_message():
    lib = _lib.load()
    c = _lib.TaeConfig(C.sizeof(_lib.TaeConfig), 100, 2, 2000, 5, 5, 2000, 5, 6, 5, 1, 0, 1, 0)
    assert lib.tae_num_weights(C.byref(c)) == 0
    assert b"must be in 1..1024" in lib.tae_last_error()
    c = _lib.TaeConfig(C.sizeof(_lib.TaeConfig), 100, 2, 100, 4, 5, 100, 5, 6, 5, 1, 0, 1, 0)      # even kernel size
    assert lib.tae_num_weights(C.byref(c)) == 0
    assert b"kernel_size" in lib.tae_last_error()
    # widths above 100 are inside the generic fp32 kernels' envelope (r03)
    c = _lib.TaeConfig(C.sizeof(_lib.TaeConfig), 100, 2, 128, 5, 5, 128, 5, 6, 5, 1, 0, 1, 0)
    assert lib.tae_num_weights(C.byref(c)) == W.num_params(TurboAEConfig(enc_num_unit=128, dec_num_unit=128))
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


def test_header_is_plain_c_and_a_c_host_links(tmp_path):
    """include/turboae_hip.h must be consumable from C (the boundary other host languages bind): compile it as strict C99,
    link a C program against the library and call the entry points that need no GPU."""
    import shutil
    import subprocess
    cc = shutil.which("gcc") or shutil.which("cc")
    if cc is None:
        pytest.skip("no C compiler")
    _lib.load()
    libdir = os.path.join(ROOT, "turboae_amd", "lib")
    src = tmp_path / "abi.c"
    src.write_text('#include <stdio.h>\n#include <string.h>\n#include "turboae_hip.h"\n'
                   "int main(void) {\n"
                   "    tae_config c; memset(&c, 0, sizeof c); c.struct_size = (int32_t)sizeof c;\n"
                   "    c.block_len = 100; c.enc_num_layer = 2; c.enc_num_unit = 100; c.enc_kernel_size = 5; c.dec_num_layer = 5;\n"
                   "    c.dec_num_unit = 100; c.dec_kernel_size = 5; c.num_iteration = 6; c.num_iter_ft = 5; c.extrinsic = 1; c.max_batch = 1;\n"
                   '    printf("%d %zu\\n", tae_abi_version(), tae_num_weights(&c));\n'
                   "    c.enc_num_unit = 2000; c.dec_num_unit = 2000;\n"
                   '    { size_t n = tae_num_weights(&c); printf("%zu %s\\n", n, tae_last_error()); }\n'
                   "    return tae_abi_version() == TAE_ABI_VERSION ? 0 : 1;\n}\n")
    exe = tmp_path / "abi"
    subprocess.check_call([cc, "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe),
                           "-L", libdir, "-lturboae_hip", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    first, second = out.stdout.splitlines()
    assert first.split() == [str(_lib.TAE_ABI_VERSION), str(W.num_params(TurboAEConfig()))]
    assert second.startswith("0 ") and "must be in 1..1024" in second


def test_config_struct_is_the_same_everywhere():
    """tae_config in the header, the ctypes mirror and the binding stub printed in INTEGRATION.md list the same int32 fields."""
    with open(os.path.join(ROOT, "include", "turboae_hip.h")) as fh:
        body = re.search(r"typedef struct tae_config \{(.*?)\} tae_config;", fh.read(), re.S).group(1)
    header = re.findall(r"^\s*int32_t\s+(\w+);", body, re.M)
    mirror = [n for n, _ in _lib.TaeConfig._fields_]
    with open(os.path.join(ROOT, "INTEGRATION.md")) as fh:
        stub = re.search(r"_fields_ = \[\(n, C\.c_int32\) for n in \((.*?)\)\]", fh.read(), re.S).group(1)
    assert header == mirror == re.findall(r'"(\w+)"', stub)
    assert C.sizeof(_lib.TaeConfig) == 4 * len(header)


def _c_config(cfg):
    act = {"elu": 0, "linear": 1, "tanh": 2, "relu": 3, "selu": 4, "sigmoid": 5}
    rnn = {"gru": 0, "lstm": 1, "rnn": 2}
    return _lib.TaeConfig(C.sizeof(_lib.TaeConfig), cfg.block_len, cfg.enc_num_layer, cfg.enc_num_unit, cfg.enc_kernel_size,
                          cfg.dec_num_layer, cfg.dec_num_unit, cfg.dec_kernel_size, cfg.num_iteration, cfg.num_iter_ft, cfg.extrinsic,
                          act[cfg.enc_act], 1, 1 if cfg.decoder == "TurboAE_rate3_rnn" else 0, 1 if cfg.encoder == "TurboAE_rate3_rnn" else 0,
                          1 if cfg.dense else 0, 1 if cfg.precision == "f32" else 0, act[cfg.dec_act], rnn[cfg.enc_rnn], rnn[cfg.dec_rnn])


def test_python_and_library_agree_on_every_configuration():
    """400 random configurations over the whole accepted space (MFMA kernels and generic fp32 kernels) and beyond it: whatever
    TurboAEConfig.validate accepts the library sizes identically (same canonical blob: same choice of kernel family, same dense /
    gate-count rules), whatever it rejects the library rejects."""
    import numpy as np
    lib = _lib.load()
    rng = np.random.RandomState(2024)
    encs, decs = ["TurboAE_rate3_cnn", "TurboAE_rate3_cnn_dense", "TurboAE_rate3_rnn"], ["TurboAE_rate3_cnn", "TurboAE_rate3_cnn_dense", "TurboAE_rate3_rnn"]
    n_ok = n_generic = n_bad = 0
    for _ in range(400):
        cfg = TurboAEConfig(block_len=int(rng.randint(1, 200)), enc_num_layer=int(rng.randint(1, 5)), dec_num_layer=int(rng.randint(1, 6)),
                            enc_num_unit=int(rng.choice([8, 32, 64, 100, 33, 150, 1024, 1500])), dec_num_unit=int(rng.choice([8, 32, 64, 100, 77, 130, 2000])),
                            enc_kernel_size=int(rng.choice([1, 3, 5, 7, 9, 11, 4, 63, 65])), dec_kernel_size=int(rng.choice([1, 3, 5, 5, 7, 9, 13, 6])),
                            num_iteration=int(rng.randint(1, 4)), num_iter_ft=int(rng.choice([1, 3, 5, 6, 7, 20, 64, 70])),
                            encoder=str(rng.choice(encs)), decoder=str(rng.choice(decs)), enc_rnn=str(rng.choice(["gru", "lstm", "rnn"])),
                            dec_rnn=str(rng.choice(["gru", "gru", "lstm", "rnn"])), precision=str(rng.choice(["auto", "f32"])))
        if cfg.decoder == "TurboAE_rate3_cnn_dense" and cfg.encoder == "TurboAE_rate3_cnn":
            continue        # a NAME the C struct cannot express (dec_type is cnn / rnn; dense follows the encoder as in decoders.py:173-176)
        try:
            cfg.validate()
            ok = True
        except ValueError:
            ok = False
        n = lib.tae_num_weights(C.byref(_c_config(cfg)))
        if ok:
            assert n == W.num_params(cfg), (cfg, n, W.num_params(cfg))
            n_ok += 1
            n_generic += int(cfg.generic)
        else:
            assert n == 0, (cfg, lib.tae_last_error())
            n_bad += 1
    assert n_ok > 150 and n_generic > 80 and n_bad > 50, (n_ok, n_generic, n_bad)
