"""Randomised shapes through the C ABI: the two independent kernel families (fp16-split and fp32 MFMA) against each other
and against the CPU oracle.  Seeded, so every run draws the same configurations; block lengths cluster around the tile
(16), workgroup (320 positions) and long-block boundaries, batches around the blocks-per-workgroup boundaries."""
import os

import numpy as np
import pytest
import torch

from turboae_amd import TurboAEConfig, philox, weights as W
from oracle import turboae_oracle as O
from _fuzz_cases import draw_cases, draw_variant_cases, draw_channel_cases, channel_case_inputs, draw_wide_cases

pytestmark = pytest.mark.gpu

# TAE_FUZZ_CASES / TAE_FUZZ_SEED widen or move the search for an ad-hoc soak run
CASES = draw_cases(int(os.environ.get("TAE_FUZZ_CASES", "48")), int(os.environ.get("TAE_FUZZ_SEED", "20240607")))


@pytest.mark.parametrize("case", CASES, ids=lambda c: "U{enc_num_unit}x{dec_num_unit}_k{enc_kernel_size}{dec_kernel_size}_L{block_len}_B{B}_e{enc_num_layer}d{dec_num_layer}_F{num_iter_ft}_it{num_iteration}_nb{fixed_nb}".format(**c))
def test_random_shape_matches_oracle_in_both_precisions(gpu_device, monkeypatch, case):
    _check_shape_case(gpu_device, monkeypatch, case)


WIDE_CASES = draw_wide_cases(int(os.environ.get("TAE_FUZZ_CASES", "18")), int(os.environ.get("TAE_FUZZ_SEED", "12404")))


@pytest.mark.parametrize("case", WIDE_CASES, ids=lambda c: "U{enc_num_unit}x{dec_num_unit}_k{enc_kernel_size}{dec_kernel_size}_L{block_len}_B{B}_e{enc_num_layer}d{dec_num_layer}_F{num_iter_ft}_it{num_iteration}".format(**c))
def test_random_wide_shape_matches_oracle(gpu_device, monkeypatch, case):
    """Widths 101 .. 124: the 124-wide instantiations of the fp16-split and (late r05) fp32 MFMA kernels, whole-block and long-block paths
    (precision='f32' with kernel sizes 7 / 9: the generic kernels)."""
    _check_shape_case(gpu_device, monkeypatch, case)


def _check_shape_case(gpu_device, monkeypatch, case):
    from turboae_amd import Channel_AE_HIP
    case = dict(case)
    B, fixed_nb, wseed = case.pop("B"), case.pop("fixed_nb"), case.pop("wseed")
    monkeypatch.setenv("TAE_DEBUG_KNOBS", "1")      # the library ignores its debug knobs without it
    monkeypatch.setenv("TAE_FIXED_NB", fixed_nb)
    cfg = TurboAEConfig(**case)
    L = cfg.block_len
    sd = W.generate_state_dict(cfg, seed=wseed, gain=1.0)
    u = philox.random_bits(wseed, 0, B * L).reshape(B, L, 1)
    noise = (np.float32(O.snr_db2sigma(1.0)) * philox.random_normal(wseed, 0, B * L * 3)).reshape(B, L, 3).astype(np.float32)
    ut, nt = torch.from_numpy(u).to(gpu_device), torch.from_numpy(noise).to(gpu_device)
    taps = {}
    xo, co = O.channel_ae_forward(torch.from_numpy(u), torch.from_numpy(noise), O.to_torch(sd), cfg.to_dict(), taps)
    xo, co = xo.numpy(), co.numpy()
    if not (np.isfinite(xo).all() and np.isfinite(co).all()):
        pytest.skip("degenerate draw: the encoder output is constant (e.g. relu of all-negative values), the power constraint divides by 0")
    # the power constraint divides by std(x_tx): a draw whose encoder output barely varies amplifies fp32 noise by 1 / std
    amplify = max(1.0, 0.25 / float(taps["std"]))
    out = {}
    precs = ("auto", "f32") if max(cfg.enc_kernel_size, cfg.dec_kernel_size) <= 5 else ("auto",)     # kernel sizes 7, 9: f16x2 only
    for prec in precs:
        model = Channel_AE_HIP(TurboAEConfig(precision=prec, **case), sd, device=gpu_device, max_batch=B)
        xd, codes = model(ut, nt)
        model.check_range()
        out[prec] = (xd.cpu().numpy(), codes.cpu().numpy())
    for prec, (xd, codes) in out.items():
        assert np.isfinite(xd).all() and np.isfinite(codes).all(), prec
        assert xd.shape == (B, L, 1) and codes.shape == (B, L, 3)
        # a one-position block normalises a constant per stream: the statistics, not the kernels, decide; keep it loose there
        tol_c, tol_x = (1e-5 * amplify, 2e-5 * amplify) if B * L >= 8 else (1e-4 * amplify, 1e-4 * amplify)
        assert np.abs(codes - co).max() <= tol_c, (prec, np.abs(codes - co).max())
        assert np.abs(xd - xo).max() <= tol_x, (prec, np.abs(xd - xo).max())
    if "f32" in out:
        assert np.abs(out["auto"][0] - out["f32"][0]).max() <= 2e-5 * amplify


VARIANT_CASES = draw_variant_cases(int(os.environ.get("TAE_FUZZ_CASES", "15")), int(os.environ.get("TAE_FUZZ_SEED", "77001")))


@pytest.mark.parametrize("case", VARIANT_CASES, ids=lambda c: "{kind}_U{0}x{1}_L{block_len}_B{B}_F{num_iter_ft}_it{num_iteration}".format(c.get("enc_num_unit", 100), c.get("dec_num_unit", 100), **c))
def test_random_shape_variants_match_oracle(gpu_device, case):
    from turboae_amd import Channel_AE_HIP
    case = dict(case)
    B, wseed, kind = case.pop("B"), case.pop("wseed"), case.pop("kind")
    cfg = TurboAEConfig(**case)
    L = cfg.block_len
    sd = W.generate_state_dict(cfg, seed=wseed, gain=1.0)
    u = philox.random_bits(wseed, 0, B * L).reshape(B, L, 1)
    noise = (np.float32(O.snr_db2sigma(1.0)) * philox.random_normal(wseed, 0, B * L * 3)).reshape(B, L, 3).astype(np.float32)
    ut, nt = torch.from_numpy(u).to(gpu_device), torch.from_numpy(noise).to(gpu_device)
    taps = {}
    xo, co = O.channel_ae_forward(torch.from_numpy(u), torch.from_numpy(noise), O.to_torch(sd), cfg.to_dict(), taps)
    xo, co = xo.numpy(), co.numpy()
    if not (np.isfinite(xo).all() and np.isfinite(co).all()):
        pytest.skip("degenerate draw: constant encoder output, the power constraint divides by 0")
    # as above: an encoder output that barely varies (relu / sigmoid of a random GRU: std 2e-3 has been drawn) is divided by its std
    amplify = max(1.0, 0.25 / float(taps["std"]))
    for prec in (("auto",) if kind == "dense" else ("auto", "f32")):      # dense stacks exist in f16x2 only
        model = Channel_AE_HIP(TurboAEConfig(precision=prec, **case), sd, device=gpu_device, max_batch=B)
        xd, codes = model(ut, nt)
        model.check_range()
        xd, codes = xd.cpu().numpy(), codes.cpu().numpy()
        assert np.isfinite(xd).all() and np.isfinite(codes).all(), prec
        tol_c, tol_x = (2e-5 * amplify, 6e-5 * amplify) if B * L >= 8 else (2e-4 * amplify, 2e-4 * amplify)   # GRU recurrences: the golden-vector tests' x_dec tolerance
        assert np.abs(codes - co).max() <= tol_c, (prec, np.abs(codes - co).max())
        assert np.abs(xd - xo).max() <= tol_x, (prec, np.abs(xd - xo).max())


CHANNEL_CASES = draw_channel_cases(int(os.environ.get("TAE_FUZZ_CASES", "24")), int(os.environ.get("TAE_FUZZ_SEED", "55021")))


@pytest.mark.parametrize("case", CHANNEL_CASES, ids=lambda c: "{channel}_{train_channel_mode}_q{enc_quantize_level}_t{enc_truncate_limit}_rq{rec_quantize}_L{block_len}_B{B}".format(**c))
def test_random_channel_options_match_oracle_stage_by_stage(gpu_device, case):
    """Quantisers make the forward discontinuous, so the stages are checked one at a time, each fed with the GPU's own previous
    stage: encoder output; power constraint + quantiser + truncation; channel + receive quantiser; decoder."""
    from turboae_amd import Channel_AE_HIP
    cfg, sd, B, u, noise, fading = channel_case_inputs(case)
    dev = gpu_device
    w = O.to_torch(sd)
    p = torch.from_numpy(O.rand_interleaver(cfg.block_len, 0))
    model = Channel_AE_HIP(cfg, sd, device=dev, max_batch=B)
    ut, nt = torch.from_numpy(u).to(dev), torch.from_numpy(noise).to(dev)
    ft = None if fading is None else torch.from_numpy(fading).to(dev)
    x_tx, stats = model.encode_prenorm(ut)
    codes, rx = model.normalize(x_tx, stats, nt, fading=ft)
    x_dec = model.dec(rx)
    xd_fused, codes_fused = model(ut, nt, ft)
    assert torch.equal(codes_fused, codes) and torch.equal(xd_fused, x_dec)              # fused call == split calls
    model.check_range()
    with torch.no_grad():
        xo = O.encode_prenorm(torch.from_numpy(u), w, p, cfg.enc_num_layer, cfg.enc_act, O.is_dense(cfg))
        if float(xo.std()) == 0.0 and not cfg.no_code_norm:
            pytest.skip("degenerate draw: constant encoder output")
        assert float((x_tx.cpu() - xo).abs().max()) <= 1e-5
        co, _, std = O.power_constraint(x_tx.cpu(), cfg.to_dict(), {})
        amplify = 1.0 if cfg.no_code_norm else max(1.0, 0.25 / float(std))
        bad = (codes.cpu() - co).abs() > 1e-5 * amplify
        assert float(bad.float().mean()) <= 2e-3, float(bad.float().mean())            # a value on a quantiser threshold may take the other level
        ro = O.apply_channel(codes.cpu(), torch.from_numpy(noise), cfg.to_dict(), None if fading is None else torch.from_numpy(fading))
        bad = (rx.cpu() - ro).abs() > 1e-5
        assert float(bad.float().mean()) <= (2e-3 if cfg.rec_quantize else 0.0)
        xd_o = O.decode(rx.cpu(), w, p, cfg.dec_num_layer, cfg.num_iteration, cfg.num_iter_ft, cfg.extrinsic, None, O.is_dense(cfg))
    assert float((x_dec.cpu() - xd_o).abs().max()) <= 2e-5
