"""HIP path on reference-trained networks in FULL fp32 precision (oracle/make_golden.py::trained_fp32) - one fixture per trained
BASELINE configuration: trained_enc2dec5_u100_fp32.npz (configs[0] / [1] / [3]), trained_enc5dec5_u100_fp32.npz (configs[2]) and
trained_cnn_gru_u100_fp32.npz (configs[4]: CNN encoder + GRU decoder), plus trained_cnn_lstm_u100_fp32.npz (-dec_rnn lstm: precision
auto runs the unit-split f16x2 kernels of turboae_rnn_u.hip, f32 the generic kernels), all at a BER of about 6e-3 @ 2 dB (main.py:162-174 loads
such a checkpoint before trainer.test).  Unlike trained_enc2dec5_u100.npz the weights are not rounded to fp16, so the
fp16-split kernels' lo halves carry real bits for every weight and the per-layer power-of-two scales (pack_stack_h) see
a trained network's dynamic range.  4 x 500 blocks per SNR point (200 000 bits: BER resolution 5e-6) at 2 / 4 / 6 dB -
6 dB sits two decades below the 2 dB point, 8 dB (20 x 500 blocks = 10^6 bits) at the 1e-5 level - against the REAL reference's hard decisions, its x_dec /
codes for batch 0, and its per-stage decoder taps."""
import json
import os

import numpy as np
import pytest
import torch

from turboae_amd import TurboAEConfig, philox, weights as W
from oracle import turboae_oracle as O

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
with open(os.path.join(GOLD, "MANIFEST.json")) as _fh:
    _MAN = json.load(_fh)
KINDS = {"trained_fp32": "trained_enc2dec5_u100_fp32.npz", "trained_enc5dec5_fp32": "trained_enc5dec5_u100_fp32.npz",
         "trained_cnn_gru_fp32": "trained_cnn_gru_u100_fp32.npz", "trained_cnn_lstm_fp32": "trained_cnn_lstm_u100_fp32.npz"}
META = _MAN["trained_fp32"]            # batch geometry / seeds / SNR points are the same for all of them

ATOL_CODES = 1e-5
ATOL_XDEC = 2e-5
from _tol import ATOL_XDEC_RNN, note


@pytest.fixture(scope="module", params=list(KINDS), ids=["enc2dec5", "enc5dec5", "cnn_gru", "cnn_lstm"])
def fixture_data(request):
    g = np.load(os.path.join(GOLD, KINDS[request.param]))
    meta = _MAN[request.param]
    cfg = TurboAEConfig(**meta["config"])
    return g, cfg, W.unpack_blob(cfg, g["weights_fp32"]), meta


def _inputs(i, snr):
    B, L, seed = META["batch"], 100, META["input_seed"]
    u = philox.random_bits(seed, i * B * L, B * L).reshape(B, L, 1)
    noise = (np.float32(O.snr_db2sigma(snr)) * philox.random_normal(seed, i * B * L * 3, B * L * 3)).reshape(B, L, 3).astype(np.float32)
    return u, noise


def test_fixture_weights_are_full_precision(fixture_data):
    g, cfg, sd, meta = fixture_data
    w = g["weights_fp32"]
    assert w.dtype == np.float32 and w.size == W.num_params(cfg)
    # not representable in fp16: the lo halves of the f16x2 split are exercised by (nearly) every weight
    assert (w.astype(np.float16).astype(np.float32) != w).mean() > 0.95
    # trained, not the generator: per-layer dynamic ranges differ by more than a factor of two
    mx = [np.abs(v).max() for k, v in sd.items() if "weight" in k and v.ndim >= 2 and v.shape[0] > 8]
    assert max(mx) / min(mx) > 2.0
    assert meta["ber"]["2dB"] < 2e-2                   # an operating point, not a coin toss


@pytest.mark.parametrize("precision", ["auto", "f32"])
def test_decisions_and_ber_match_reference_across_snr_points(gpu_device, fixture_data, precision):
    from dataclasses import replace
    from turboae_amd import Channel_AE_HIP
    g, cfg, sd, meta = fixture_data
    rnn = cfg.decoder == "TurboAE_rate3_rnn"
    atol_x = ATOL_XDEC_RNN if rnn else ATOL_XDEC       # the recurrent decoders' bound everywhere (tests/_tol.py: measured)
    B, L = META["batch"], 100
    model = Channel_AE_HIP(replace(cfg, precision=precision), sd, device=gpu_device, max_batch=B)
    for snr in META["snrs"]:
        key = f"{snr:g}dB"
        NB = META["n_batches"][key]
        hard_ref = np.unpackbits(g[f"hard_bits_{key}"])[: NB * B * L].reshape(NB, B, L)
        flips_total, ber_batches = 0, []
        for i in range(NB):
            u, noise = _inputs(i, snr)
            xd, codes = model(torch.from_numpy(u).to(gpu_device), torch.from_numpy(noise).to(gpu_device))
            xd, codes = xd.cpu().numpy(), codes.cpu().numpy()
            hard = (xd[:, :, 0] > 0.5).astype(np.uint8)
            flips = hard != hard_ref[i]
            flips_total += int(flips.sum())
            errs = int((hard != u[:, :, 0].astype(np.uint8)).sum())
            assert abs(errs - meta["bit_errors"][key][i]) <= 2, (key, i, errs)
            ber_batches.append(errs / (B * L))
            if i == 0:
                xr = g[f"x_dec_batch0_{key}"]
                if rnn:
                    note(f"trained:{cfg.dec_rnn}:{key}:{precision}", np.abs(xd - xr).max())
                assert np.abs(xd - xr).max() <= atol_x, (key, np.abs(xd - xr).max())
                # a decision may only differ where the reference's own soft output is within fp32 noise of 1/2
                assert np.all(np.abs(xr[:, :, 0][flips] - 0.5) < 1e-4)
                if snr == META["snrs"][0]:
                    assert np.abs(codes - g["codes_batch0"]).max() <= ATOL_CODES
        assert flips_total <= 3, (key, flips_total)
        # mean of batch means, as trainer.test reports it (trainer.py:176-177,215-216)
        assert abs(float(np.mean(ber_batches)) - meta["ber"][key]) <= 1.5e-5, key
    mode, ovf = model.range_status()
    assert mode == ("f16x2" if precision == "auto" else "f32") and not ovf
    # the low-BER points really are low: 6 dB more than a decade under 2 dB, 8 dB (10^6 bits) under 1e-4
    assert meta["ber"]["6dB"] < 0.1 * meta["ber"]["2dB"] and meta["ber"]["8dB"] < 1e-4


@pytest.mark.parametrize("precision", ["auto", "f32"])
def test_stage_taps_on_trained_weights(gpu_device, fixture_data, precision):
    from dataclasses import replace
    from turboae_amd import Channel_AE_HIP
    g, cfg, sd, meta = fixture_data
    assert "dec_taps_first4" in g.files      # r06: the recurrent fixtures carry them too (oracle/make_golden.py --trained-taps)
    _, noise = _inputs(0, META["snrs"][0])
    rx = torch.from_numpy(g["codes_batch0"][:4] + noise[:4]).to(gpu_device)
    model = Channel_AE_HIP(replace(cfg, precision=precision), sd, device=gpu_device, max_batch=4)
    xd, taps = model.decode_taps(rx)
    ref = g["dec_taps_first4"]
    taps = taps.cpu().numpy()
    for s in range(ref.shape[0]):
        d = np.abs(taps[s] - ref[s]).max()
        assert d <= 2e-5 * max(1.0, np.abs(ref[s]).max()), (s, d)
    rnn = cfg.decoder == "TurboAE_rate3_rnn"
    assert np.abs(xd.cpu().numpy() - g[f"x_dec_batch0_{META['snrs'][0]:g}dB"][:4]).max() <= (ATOL_XDEC_RNN if rnn else ATOL_XDEC)


def test_precisions_against_float64_oracle_on_trained_weights(gpu_device, fixture_data):
    """f16x2 on full-precision TRAINED weights is no further from a float64 evaluation of the same network than the
    exact-fp32 MFMA kernels (the claim DESIGN.md 3.7 makes, here on weights whose lo halves are not zero)."""
    from dataclasses import replace
    from turboae_amd import Channel_AE_HIP
    g, cfg, sd, meta = fixture_data
    B = 16
    u, noise = _inputs(0, 2.0)
    u, noise = u[:B], noise[:B]
    sd64 = {k: torch.from_numpy(np.asarray(v, dtype=np.float64)) for k, v in sd.items()}
    x64, c64 = O.channel_ae_forward(torch.from_numpy(u).double(), torch.from_numpy(noise).double(), sd64, cfg.to_dict())
    err = {}
    for prec in ("auto", "f32"):
        model = Channel_AE_HIP(replace(cfg, precision=prec), sd, device=gpu_device, max_batch=B)
        xd, codes = model(torch.from_numpy(u).to(gpu_device), torch.from_numpy(noise).to(gpu_device))
        err[prec] = (float((codes.cpu().double() - c64).abs().max()), float((xd.cpu().double() - x64).abs().max()))
    print("trained fp32 weights, max |err| vs float64 oracle (codes, x_dec):", err)
    for k in (0, 1):
        assert err["auto"][k] <= 2.0 * err["f32"][k] + 5e-7, err
    assert err["auto"][0] <= ATOL_CODES and err["auto"][1] <= (ATOL_XDEC_RNN if cfg.decoder == "TurboAE_rate3_rnn" else ATOL_XDEC)
