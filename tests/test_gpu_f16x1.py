"""precision='f16x1' (TAE_PREC_F16X1): the OPTIONAL, separately labelled one-product decoder.  It carries NO parity claim -
these tests pin what it is: the f16x2 handle with the decoder's contraction on the hi halves only, deterministic, never selected by
'auto', soft outputs within fp16-rounding distance of the fp32-grade path, and loud rejections everywhere it is not instantiated."""
import os

import numpy as np
import pytest
import torch

from turboae_amd import TurboAEConfig, philox, weights as W
from oracle import turboae_oracle as O

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _trained():
    cfg = TurboAEConfig()
    return cfg, W.unpack_blob(cfg, np.load(os.path.join(GOLD, "trained_enc2dec5_u100_fp32.npz"))["weights_fp32"])


def test_one_product_decoder_is_close_but_not_fp32_grade(gpu_device):
    from dataclasses import replace
    from turboae_amd import Channel_AE_HIP
    cfg, sd = _trained()
    B = 3000
    ref = Channel_AE_HIP(cfg, sd, device=gpu_device, max_batch=B)
    x1 = Channel_AE_HIP(replace(cfg, precision="f16x1"), sd, device=gpu_device, max_batch=B)
    assert ref.range_status() == ("f16x2", False) and x1.range_status() == ("f16x1", False)
    u, noise = ref.generate_inputs(B, 2.0, seed=5)
    codes = ref.enc(u)
    assert torch.equal(x1.enc(u), codes)                       # the encoder is the f16x2 one
    rx = codes + noise
    a, b = ref.dec(rx), x1.dec(rx)
    assert torch.equal(b, x1.dec(rx))                          # deterministic
    for lo, hi in ((0, 1), (5, 9), (B - 2, B)):                # batch-independent like every other path
        assert torch.equal(x1.dec(rx[lo:hi].contiguous()), b[lo:hi])
    d = float((a - b).abs().max())
    assert 1e-5 < d < 0.2, d                                   # NOT fp32-grade (fp16 operands: ~1e-3 per layer), and not garbage either
    flips = int(((a > 0.5) != (b > 0.5)).sum())
    nerr_a, nerr_b = int(((a > 0.5) != (u > 0.5)).sum()), int(((b > 0.5) != (u > 0.5)).sum())
    print(f"f16x1 vs f16x2 on {B * 100} bits @ 2 dB: max |dx| {d:.3e}, {flips} decision flips, bit errors {nerr_b} vs {nerr_a}")
    assert flips <= 0.002 * B * 100 and abs(nerr_a - nerr_b) <= 0.1 * nerr_a + 20
    assert x1.range_status() == ("f16x1", False)
    # the tap / calibration instantiation of such a handle is the fp32-grade one: taps come out, x_dec of THAT launch equals the f16x2 path's
    xd_t, taps = x1.decode_taps(rx[:4].contiguous())
    assert float((xd_t - a[:4]).abs().max()) <= 1e-6 and bool(torch.isfinite(taps).all())


def test_f16x1_is_rejected_where_it_is_not_instantiated(gpu_device):
    from turboae_amd import Channel_AE_HIP
    for over in (dict(decoder="TurboAE_rate3_rnn"), dict(dec_num_unit=32), dict(block_len=1000), dict(dec_kernel_size=7),
                 dict(encoder="TurboAE_rate3_cnn_dense", decoder="TurboAE_rate3_cnn_dense"), dict(dec_num_unit=124, enc_num_unit=124)):
        with pytest.raises(ValueError):
            TurboAEConfig(precision="f16x1", **over).validate()
    # the library checks for itself (a C caller has no TurboAEConfig): dec_num_unit = 32 through the raw ABI
    import ctypes as C
    from turboae_amd import _lib
    lib = _lib.load()
    cfg = TurboAEConfig(dec_num_unit=32)
    c = _lib.TaeConfig(C.sizeof(_lib.TaeConfig), cfg.block_len, cfg.enc_num_layer, cfg.enc_num_unit, cfg.enc_kernel_size, cfg.dec_num_layer,
                       cfg.dec_num_unit, cfg.dec_kernel_size, cfg.num_iteration, cfg.num_iter_ft, cfg.extrinsic, 0, 4, 0, 0, 0, 2, 1, 0, 0, 0, 0)
    blob = W.pack_blob(cfg, W.generate_state_dict(cfg, seed=1, gain=1.0))
    h = C.c_void_p()
    with torch.cuda.device(gpu_device):
        rc = lib.tae_create(C.byref(c), blob.ctypes.data_as(C.c_void_p), blob.size, C.byref(h))
    assert rc != 0 and b"F16X1" in lib.tae_last_error()
