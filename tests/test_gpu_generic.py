"""Configurations outside the MFMA kernels' envelope, through the C ABI on the library's generic fp32 kernels
(csrc/turboae_generic.hip): golden vectors from the REAL reference for LSTM / vanilla-RNN cells, ENC_interRNN with 1 and 3 layers,
an RNN encoder in front of the (then dense) CNN decoder, channel widths above 100, num_iter_ft above 6 and kernel sizes above 9;
`precision='f32'` for the variants whose MFMA kernels exist in the fp16-split arithmetic only (dense stacks, kernel sizes 7 / 9)
against the same reference vectors; random generic configurations against the oracle; the range fall-back."""
import json
import os

import numpy as np
import pytest
import torch

from turboae_amd import TurboAEConfig, philox, weights as W
from oracle import turboae_oracle as O
from _fuzz_cases import draw_generic_cases
from _tol import ATOL_XDEC_RNN, note

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden")
with open(os.path.join(GOLD, "MANIFEST.json")) as fh:
    MANIFEST = json.load(fh)

GEN = [n for n in sorted(MANIFEST["cases"]) if n.startswith("gen_")]


def _run(gpu_device, cfg, meta, name):
    from turboae_amd import Channel_AE_HIP
    sd = W.golden_state_dict(cfg, meta)
    g = np.load(os.path.join(GOLD, name + ".npz"))
    model = Channel_AE_HIP(cfg, sd, device=gpu_device, max_batch=meta["B"])
    xd, codes = model(torch.from_numpy(g["u"]).to(gpu_device), torch.from_numpy(g["noise"]).to(gpu_device))
    return model, g, xd.cpu().numpy(), codes.cpu().numpy()


@pytest.mark.parametrize("name", GEN)
def test_generic_configurations_match_reference_golden(gpu_device, name):
    assert len(GEN) >= 9
    from dataclasses import replace
    meta = MANIFEST["cases"][name]
    cfg = TurboAEConfig(**meta["config"])
    if not cfg.generic:      # r05: LSTM / RNN decoders behind the CNN encoder have unit-split f16x2 kernels; precision = f32 keeps them here
        assert name in ("gen_dec_lstm", "gen_dec_rnn_tanh")
        cfg = replace(cfg, precision="f32")
    assert cfg.generic
    model, g, xd, codes = _run(gpu_device, cfg, meta, name)
    assert model.range_status() == ("f32", False) and model.kernel_info() == (0, 0)
    assert np.abs(codes - g["codes"]).max() <= 1e-5, np.abs(codes - g["codes"]).max()
    d = np.abs(xd - g["x_dec"]).max()
    assert d <= 2e-5, d
    flips = (xd > 0.5) != (g["x_dec"] > 0.5)
    assert np.all(np.abs(g["logits"][flips]) < 2e-4)
    # the split entry points agree with the fused forward on this path too
    u = torch.from_numpy(g["u"]).to(gpu_device)
    assert torch.equal(model.enc(u).cpu(), torch.from_numpy(codes))
    assert torch.equal(model.dec(torch.from_numpy(codes + g["noise"]).to(gpu_device)).cpu(), torch.from_numpy(xd))


@pytest.mark.parametrize("form", ["auto", "fused"])
@pytest.mark.parametrize("name", ["gen_dec_lstm", "gen_dec_rnn_tanh", "fwd_lstm_u100_L100_b3_it3", "fwd_rnntanh_u100_L64_b3_it2",
                                  "fwd_encrnn_declstm_u100_L64_b3_it2", "fwd_rnn_encrnn_declstm_e64_d48_L40_b3", "fwd_rnn_enclstm_decgru_u100_L64_b3_it2"])
def test_lstm_and_rnn_decoders_on_the_unit_split_f16x2_kernels(gpu_device, monkeypatch, name, form):
    """`-dec_rnn lstm | rnn` (decoders.py:27-32) behind the CNN encoder run on turboae_rnn_u.hip in the default
    arithmetic - the reference's golden vectors, the GRU tests' tolerances; the generic fp32 kernels stay the second implementation
    (precision = f32) and agree with it.  r06: the goldens' small batches run layer 1 as projection + recurrence; `fused` forces
    rnn_l1f_u_kernel (what full batches run) onto the same reference vectors."""
    from dataclasses import replace
    meta = MANIFEST["cases"][name]
    cfg = TurboAEConfig(**meta["config"])
    assert not cfg.generic and cfg.decoder == "TurboAE_rate3_rnn" and (cfg.dec_rnn in ("lstm", "rnn") or cfg.enc_rnn in ("lstm", "rnn"))
    if form == "fused":
        monkeypatch.setenv("TAE_DEBUG_KNOBS", "1")
        monkeypatch.setenv("TAE_RNN_L1", "fused")
    else:
        monkeypatch.delenv("TAE_RNN_L1", raising=False)
    model, g, xd, codes = _run(gpu_device, cfg, meta, name)
    if form == "fused":
        assert "TAE_RNN_L1=fused" in model.overrides()
    assert model.range_status() == ("f16x2", False)
    assert np.abs(codes - g["codes"]).max() <= 1e-5
    d = note(f"golden_u:{name}:{form}", np.abs(xd - g["x_dec"]).max())
    assert d <= ATOL_XDEC_RNN, d
    flips = (xd > 0.5) != (g["x_dec"] > 0.5)
    assert np.all(np.abs(g["logits"][flips]) < 2e-4)
    _, _, xd32, codes32 = _run(gpu_device, replace(cfg, precision="f32"), meta, name)
    assert np.abs(xd - xd32).max() <= ATOL_XDEC_RNN and np.abs(codes - codes32).max() <= 1e-5


@pytest.mark.parametrize("cell,B,L,U,F", [("lstm", 37, 100, 100, 5), ("rnn", 70, 33, 100, 5), ("lstm", 5, 7, 100, 5),
                                          ("lstm", 1, 1, 100, 5), ("lstm", 3, 321, 100, 5), ("rnn", 33, 1000, 100, 5),
                                          ("lstm", 20, 50, 37, 3), ("rnn", 40, 64, 64, 6)])
def test_lstm_rnn_unit_split_kernels_batch_independence_and_oracle(gpu_device, cell, B, L, U, F):
    """LSTM / RNN decoders on the unit-split kernels at the width they are built for (100) and narrower cells embedded in it: against
    the float64-free oracle, ragged batches (partial groups of 32 blocks), one block, one position, long blocks, other num_iter_ft,
    sub-batches bit for bit, determinism."""
    from turboae_amd import Channel_AE_HIP
    cfg = TurboAEConfig(decoder="TurboAE_rate3_rnn", dec_rnn=cell, block_len=L, num_iteration=2, dec_num_unit=U, num_iter_ft=F)
    assert not cfg.generic
    sd = W.generate_state_dict(cfg, seed=500 + L, gain=1.0)
    u = philox.random_bits(9, 0, B * L).reshape(B, L, 1)
    noise = (np.float32(O.snr_db2sigma(1.0)) * philox.random_normal(9, 0, B * L * 3)).reshape(B, L, 3).astype(np.float32)
    xo, co = O.channel_ae_forward(torch.from_numpy(u), torch.from_numpy(noise), O.to_torch(sd), cfg.to_dict(), {})
    model = Channel_AE_HIP(cfg, sd, device=gpu_device, max_batch=B)
    ud, nd = torch.from_numpy(u).to(gpu_device), torch.from_numpy(noise).to(gpu_device)
    xd, codes = model(ud, nd)
    assert model.range_status() == ("f16x2", False)
    assert np.abs(codes.cpu().numpy() - co.numpy()).max() <= 1e-5
    d = np.abs(xd.cpu().numpy() - xo.numpy()).max()
    assert note(f"oracle_u:{cell}:B{B}_L{L}_U{U}_F{F}", d) <= ATOL_XDEC_RNN, d
    rx = codes + nd
    assert torch.equal(model.dec(rx), xd)                                  # run to run
    for lo, hi in ((0, 1), (B // 2, min(B, B // 2 + 3)), (max(0, B - 2), B)):
        assert torch.equal(model.dec(rx[lo:hi].contiguous()), xd[lo:hi]), (lo, hi)


@pytest.mark.parametrize("cell,B,L,U", [("lstm", 1, 100, 100), ("lstm", 37, 100, 100), ("rnn", 70, 33, 100), ("lstm", 5, 7, 100), ("lstm", 33, 1, 64),
                                         ("rnn", 16, 2, 100), ("lstm", 16, 3, 100), ("lstm", 500, 100, 100), ("rnn", 2100, 40, 80), ("lstm", 2049, 101, 100),
                                         ("lstm", 16400, 9, 100), ("lstm", 1650, 1000, 100), ("rnn", 3000, 321, 100)])
def test_rnn_layer1_forms_are_bit_identical(gpu_device, monkeypatch, cell, B, L, U):
    """Layer 1 of the LSTM / vanilla-RNN decoder stacks exists in two forms: rnn_l1f_u_kernel (r06: the input projection inside the
    recurrence, four steps at a time, GI never written) and the r05 pair rnn_proj_u -> rnn_rec_u<layer 1>.  The library picks by
    batch size (the pair below 6 blocks per CU, the fused kernel from there on: it is 23 % faster at 16 384 blocks and slower below
    ~1 200), so results may not depend on the choice: every accumulator sees the same products in the same order, and the two forms
    must agree bit for bit - block lengths that are and are not multiples of the chunk, ragged batches, both cells, two chunks."""
    from turboae_amd import Channel_AE_HIP
    cfg = TurboAEConfig(decoder="TurboAE_rate3_rnn", dec_rnn=cell, block_len=L, dec_num_unit=U, num_iteration=2)
    sd = W.generate_state_dict(cfg, seed=900 + L + B, gain=1.0)
    u = philox.random_bits(19, 0, B * L).reshape(B, L, 1)
    noise = (np.float32(O.snr_db2sigma(1.0)) * philox.random_normal(19, 0, B * L * 3)).reshape(B, L, 3).astype(np.float32)
    ud, nd = torch.from_numpy(u).to(gpu_device), torch.from_numpy(noise).to(gpu_device)
    monkeypatch.delenv("TAE_RNN_L1", raising=False)
    auto = Channel_AE_HIP(cfg, sd, device=gpu_device, max_batch=B)
    xa, ca = [t.clone() for t in auto(ud, nd)]
    out = {}
    monkeypatch.setenv("TAE_DEBUG_KNOBS", "1")      # the library ignores its debug knobs without it
    for form in ("split", "fused"):
        monkeypatch.setenv("TAE_RNN_L1", form)
        m = Channel_AE_HIP(cfg, sd, device=gpu_device, max_batch=B)
        assert f"TAE_RNN_L1={form}" in m.overrides()
        out[form] = [t.clone() for t in m(ud, nd)]
        assert m.range_status() == ("f16x2", False)
    assert torch.equal(out["split"][0], out["fused"][0]) and torch.equal(out["split"][1], out["fused"][1])
    assert torch.equal(xa, out["fused"][0]) and torch.equal(ca, out["fused"][1])
    if B * L <= 5000:
        xo, _ = O.channel_ae_forward(torch.from_numpy(u), torch.from_numpy(noise), O.to_torch(sd), cfg.to_dict(), {})
        assert np.abs(xa.cpu().numpy() - xo.numpy()).max() <= ATOL_XDEC_RNN


@pytest.mark.parametrize("enc_rnn,dec_rnn", [("lstm", "gru"), ("rnn", "lstm"), ("lstm", "lstm")])
def test_lstm_rnn_encoder_cells_on_the_tuned_kernels(gpu_device, monkeypatch, enc_rnn, dec_rnn):
    """r06: -enc_rnn lstm | rnn (ENC_interRNN, encoders.py:242-253, 2 layers) on turboae_rnn_u.hip, in front of any recurrent decoder: a batch
    large enough for the fused layer 1 (1 600 blocks >= 6 per CU) against the CPU oracle, and the split / fused forms bit-identical
    (the encoder shares the decoder's chunk workspace: GI is sized for the wider of the two cells)."""
    from turboae_amd import Channel_AE_HIP
    cfg = TurboAEConfig(encoder="TurboAE_rate3_rnn", decoder="TurboAE_rate3_rnn", enc_rnn=enc_rnn, dec_rnn=dec_rnn, block_len=24, num_iteration=1,
                        enc_num_unit=100 if enc_rnn == "lstm" else 72)
    assert not cfg.generic
    B, L = 1600, cfg.block_len
    sd = W.generate_state_dict(cfg, seed=321, gain=1.0)
    u = philox.random_bits(29, 0, B * L).reshape(B, L, 1)
    noise = (np.float32(O.snr_db2sigma(2.0)) * philox.random_normal(29, 0, B * L * 3)).reshape(B, L, 3).astype(np.float32)
    ud, nd = torch.from_numpy(u).to(gpu_device), torch.from_numpy(noise).to(gpu_device)
    out = {}
    monkeypatch.setenv("TAE_DEBUG_KNOBS", "1")
    for form in ("split", "fused"):
        monkeypatch.setenv("TAE_RNN_L1", form)
        m = Channel_AE_HIP(cfg, sd, device=gpu_device, max_batch=B)
        out[form] = [t.clone() for t in m(ud, nd)]
        assert m.range_status() == ("f16x2", False)
    assert torch.equal(out["split"][0], out["fused"][0]) and torch.equal(out["split"][1], out["fused"][1])
    monkeypatch.delenv("TAE_RNN_L1")
    m = Channel_AE_HIP(cfg, sd, device=gpu_device, max_batch=B)
    xd, codes = m(ud, nd)
    assert torch.equal(xd, out["fused"][0]) and torch.equal(codes, out["fused"][1])
    xo, co = O.channel_ae_forward(torch.from_numpy(u), torch.from_numpy(noise), O.to_torch(sd), cfg.to_dict(), {})
    assert np.abs(codes.cpu().numpy() - co.numpy()).max() <= 1e-5
    assert note(f"oracle_enc:{enc_rnn}:{dec_rnn}", np.abs(xd.cpu().numpy() - xo.numpy()).max()) <= ATOL_XDEC_RNN


def test_lstm_eval_sweep_as_hipgraphs_on_the_fused_layer1(gpu_device):
    """evaluate.test (trainer.test restated) on a reference-trained LSTM decoder at a batch the fused layer-1 kernel serves (2 048 blocks
    per call >= 6 per CU): one hipGraph per SNR point gives the eager sweep's counts, eval_snr (the C sweep entry) agrees, and the BER at
    2 dB sits where the reference measured this network."""
    from turboae_amd import Channel_AE_HIP, evaluate
    cfg = TurboAEConfig(decoder="TurboAE_rate3_rnn", dec_rnn="lstm")
    sd = W.unpack_blob(cfg, np.load(os.path.join(GOLD, "trained_cnn_lstm_u100_fp32.npz"))["weights_fp32"])
    B = 2048
    model = Channel_AE_HIP(cfg, sd, device=gpu_device, max_batch=B)
    kw = dict(snr_test_start=1.0, snr_test_end=3.0, snr_points=3, num_block=2 * B, batch_size=B, seed=11, verbose=False)
    res = evaluate.test(model, **kw)
    gr = evaluate.test(model, hip_graph=True, **kw)
    assert gr["bit_errors"] == res["bit_errors"] and gr["block_errors"] == res["block_errors"]
    c = model.eval_snr(res["snrs"][1], B, 2, seed=11, first_block=2 * B).sum(dim=0).cpu().tolist()
    assert c == [res["bit_errors"][1], res["block_errors"][1]]
    ref = MANIFEST["trained_cnn_lstm_fp32"]["ber"]["2dB"]
    assert abs(res["ber"][1] - ref) <= 0.15 * ref, (res["ber"][1], ref)
    model.check_range()


@pytest.mark.parametrize("name", ["fwd_dense_u100_L100_b3_it2", "fwd_dense_k3_k1_u32_L64", "var_kernel_e7_d9", "var_kernel_e9_d7_L500"])
def test_precision_f32_for_dense_stacks_and_kernel_sizes_7_9(gpu_device, name):
    """the second arithmetic for the variants the fp32 MFMA kernels do not cover: the same reference vectors, fp32 end to end"""
    from dataclasses import replace
    meta = MANIFEST["cases"][name]
    cfg = replace(TurboAEConfig(**meta["config"]), precision="f32")
    assert cfg.generic
    model, g, xd, codes = _run(gpu_device, cfg, meta, name)
    assert model.range_status()[0] == "f32"
    assert np.abs(codes - g["codes"]).max() <= 1e-5
    assert np.abs(xd - g["x_dec"]).max() <= 2e-5
    # and the fp16-split kernels' result for the same network is within the same bound of it
    auto = replace(cfg, precision="auto")
    assert not auto.generic
    _, _, xd2, codes2 = _run(gpu_device, auto, meta, name)
    assert np.abs(codes - codes2).max() <= 1e-5 and np.abs(xd - xd2).max() <= 2e-5


# TAE_FUZZ_CASES_GENERIC / TAE_FUZZ_SEED_GENERIC widen or move the search for a soak run (tools/gpu_soak.sh)
@pytest.mark.parametrize("case", draw_generic_cases(int(os.environ.get("TAE_FUZZ_CASES_GENERIC", "21")), int(os.environ.get("TAE_FUZZ_SEED_GENERIC", "90210"))), ids=lambda c: "{kind}_L{block_len}_B{B}_e{enc_num_unit}x{enc_num_layer}_d{dec_num_unit}x{dec_num_layer}_F{num_iter_ft}".format(**c))
def test_random_generic_configurations_match_oracle(gpu_device, case):
    from turboae_amd import Channel_AE_HIP
    case = dict(case)
    B, wseed, kind = case.pop("B"), case.pop("wseed"), case.pop("kind")
    cfg = TurboAEConfig(**case)
    if not cfg.generic:            # a "wide" draw with both widths <= 124 (MFMA territory in both arithmetics) or an LSTM / RNN decoder
        from dataclasses import replace     # behind the CNN encoder (unit-split f16x2 kernels): precision='f32' keeps the recurrent ones generic
        cfg = replace(cfg, precision="f32")
    assert cfg.generic or kind == "wide"
    L = cfg.block_len
    sd = W.generate_state_dict(cfg, seed=wseed, gain=1.0)
    u = philox.random_bits(wseed, 0, B * L).reshape(B, L, 1)
    noise = (np.float32(O.snr_db2sigma(1.0)) * philox.random_normal(wseed, 0, B * L * 3)).reshape(B, L, 3).astype(np.float32)
    taps = {}
    xo, co = O.channel_ae_forward(torch.from_numpy(u), torch.from_numpy(noise), O.to_torch(sd), cfg.to_dict(), taps)
    xo, co = xo.numpy(), co.numpy()
    if not (np.isfinite(xo).all() and np.isfinite(co).all()):
        pytest.skip("degenerate draw: constant encoder output")
    amplify = max(1.0, 0.25 / float(taps["std"]))
    model = Channel_AE_HIP(cfg, sd, device=gpu_device, max_batch=B)
    xd, codes = model(torch.from_numpy(u).to(gpu_device), torch.from_numpy(noise).to(gpu_device))
    xd, codes = xd.cpu().numpy(), codes.cpu().numpy()
    tol_c, tol_x = (2e-5 * amplify, 6e-5 * amplify) if B * L >= 8 else (2e-4 * amplify, 2e-4 * amplify)
    assert np.abs(codes - co).max() <= tol_c, np.abs(codes - co).max()
    assert np.abs(xd - xo).max() <= tol_x, np.abs(xd - xo).max()


# r04: the generic recurrence runs on the fp32 matrix cores for H <= 128, H % 4 == 0 (gen_rnn_mfma_kernel<G, KS, 1, XK>, KS = 8 / 16 /
# 25 / 32 k-steps, XK = 2 for first layers: input projection fused), 16 blocks per workgroup; other widths on the vector-ALU kernel
# (1 / 4 / 8 blocks per workgroup by batch).  One case per instantiation and edge: widths that are no multiple of 4 (vector ALU),
# batches that leave a ragged last workgroup, the last width of the MFMA kernel and the first of the vector-ALU one.
@pytest.mark.parametrize("cell,H,B,L", [("lstm", 27, 5, 12), ("lstm", 28, 5, 12), ("gru", 64, 37, 10), ("rnn", 100, 16, 9), ("lstm", 100, 33, 16), ("gru", 101, 3, 7),
                                         ("lstm", 128, 17, 6), ("rnn", 124, 50, 5), ("gru", 120, 18, 4), ("rnn", 126, 50, 5), ("lstm", 130, 9, 6),
                                         ("gru", 200, 4, 5), ("rnn", 7, 1, 1), ("rnn", 4, 2, 3)])
def test_generic_recurrence_kernels_every_instantiation(gpu_device, monkeypatch, cell, H, B, L):
    from turboae_amd import Channel_AE_HIP
    cfg = TurboAEConfig(decoder="TurboAE_rate3_rnn", dec_rnn=cell, dec_num_unit=H, enc_num_unit=20, block_len=L, num_iteration=2)
    if not cfg.generic:            # 2-layer GRUs up to 100 units have their own MFMA kernels: the testing knob puts them on these
        monkeypatch.setenv("TAE_DEBUG_KNOBS", "1")      # the library ignores its debug knobs without it
        monkeypatch.setenv("TAE_FORCE_GENERIC", "1")
    wseed = 1000 + H
    sd = W.generate_state_dict(cfg, seed=wseed, gain=1.0)
    u = philox.random_bits(wseed, 0, B * L).reshape(B, L, 1)
    noise = (np.float32(O.snr_db2sigma(1.0)) * philox.random_normal(wseed, 0, B * L * 3)).reshape(B, L, 3).astype(np.float32)
    xo, co = O.channel_ae_forward(torch.from_numpy(u), torch.from_numpy(noise), O.to_torch(sd), cfg.to_dict(), {})
    model = Channel_AE_HIP(cfg, sd, device=gpu_device, max_batch=B)
    assert model.kernel_info() == (0, 0)
    xd, codes = model(torch.from_numpy(u).to(gpu_device), torch.from_numpy(noise).to(gpu_device))
    assert np.abs(codes.cpu().numpy() - co.numpy()).max() <= 2e-5
    d = np.abs(xd.cpu().numpy() - xo.numpy()).max()
    assert d <= 6e-5, d


def test_generic_recurrence_does_not_depend_on_the_batch(gpu_device):
    """The vector-ALU recurrence (H > 128) shares W_hh between 1 / 4 / 8 blocks of a workgroup depending on the batch, the MFMA one
    (H <= 128) between 16: a block's result must not depend on which batch it arrived in (same FMA chain / same MFMA operands)."""
    from turboae_amd import Channel_AE_HIP
    for H, L in ((130, 3), (100, 4)):
        cfg = TurboAEConfig(decoder="TurboAE_rate3_rnn", dec_rnn="lstm", dec_num_unit=H, enc_num_unit=8, enc_num_layer=1, block_len=L, num_iteration=1)
        model = Channel_AE_HIP(cfg, W.generate_state_dict(cfg, seed=5, gain=1.0), device=gpu_device, max_batch=4100)
        rx = torch.from_numpy(philox.random_normal(6, 0, 4100 * L * 3).reshape(4100, L, 3).astype(np.float32)).to(gpu_device)
        big = model.dec(rx)                        # 8 blocks per workgroup (H = 130)
        for nb in (1029, 5):                       # 4 blocks per workgroup, 1 block per workgroup
            assert torch.equal(model.dec(rx[:nb].contiguous()), big[:nb]), (H, nb)


@pytest.mark.parametrize("decoder", ["TurboAE_rate3_cnn", "TurboAE_rate3_rnn"])
def test_three_independent_implementations_agree_on_the_trained_network(gpu_device, monkeypatch, decoder):
    """TAE_FORCE_GENERIC=1 runs a standard configuration on the generic kernels (fp32 MFMA, one launch per layer): a third implementation next to the fp16-split
    and the fp32 MFMA kernels.  Reference-trained weights (CNN decoder) / random weights (GRU decoder), 1 000 blocks = 100 000 bits."""
    from dataclasses import replace
    from turboae_amd import Channel_AE_HIP
    cfg = TurboAEConfig(decoder=decoder, num_iteration=6 if decoder.endswith("cnn") else 2)
    if decoder.endswith("cnn"):
        sd = W.unpack_blob(TurboAEConfig(), np.load(os.path.join(GOLD, "trained_enc2dec5_u100_fp32.npz"))["weights_fp32"])
    else:
        sd = W.generate_state_dict(cfg, seed=77, gain=1.0)
    B = 1000
    auto = Channel_AE_HIP(cfg, sd, device=gpu_device, max_batch=B)
    f32 = Channel_AE_HIP(replace(cfg, precision="f32"), sd, device=gpu_device, max_batch=B)
    monkeypatch.setenv("TAE_DEBUG_KNOBS", "1")      # the library ignores its debug knobs without it
    monkeypatch.setenv("TAE_FORCE_GENERIC", "1")
    gen = Channel_AE_HIP(replace(cfg, precision="f32"), sd, device=gpu_device, max_batch=B)
    monkeypatch.delenv("TAE_FORCE_GENERIC")
    assert auto.kernel_info()[0] > 0 and gen.kernel_info() == (0, 0) and gen.range_status()[0] == "f32" and auto.range_status()[0] == "f16x2"
    u, noise = auto.generate_inputs(B, 2.0, seed=99)
    xg, cg = gen(u, noise)
    tol_x = 2e-5 if decoder.endswith("cnn") else 6e-5
    for name, m in (("f16x2", auto), ("f32 MFMA", f32)):
        x, c = m(u, noise)
        assert float((c - cg).abs().max()) <= 1e-5, name
        d = float((x - xg).abs().max())
        assert d <= tol_x, (name, d)
        flips = (x > 0.5) != (xg > 0.5)
        assert int(flips.sum()) <= 1 and bool(((xg[flips] - 0.5).abs() < 1e-4).all()), (name, int(flips.sum()))
    if decoder.endswith("cnn"):
        ber = float(((xg > 0.5) != (u > 0.5)).float().mean())
        assert 4e-3 < ber < 1e-2, ber            # the trained network's operating point at 2 dB (reference: 6.4e-3)


@pytest.mark.parametrize("precision", ["f32", "auto"])
def test_generic_path_runs_the_eval_sweep_and_every_entry_point(gpu_device, precision):
    """evaluate.test, tae_eval_snr and the hipGraph form on an LSTM decoder (all launches are plain kernels on the caller's stream): on
    the generic fp32 kernels (precision f32) and on the unit-split f16x2 kernels (auto)"""
    from turboae_amd import Channel_AE_HIP, evaluate
    cfg = TurboAEConfig(decoder="TurboAE_rate3_rnn", dec_rnn="lstm", enc_num_unit=16, dec_num_unit=12, num_iteration=1, block_len=20, precision=precision)
    assert cfg.generic == (precision == "f32")
    model = Channel_AE_HIP(cfg, W.generate_state_dict(cfg, seed=3, gain=1.0), device=gpu_device, max_batch=40)
    res = evaluate.test(model, snr_test_start=0.0, snr_test_end=2.0, snr_points=2, num_block=80, batch_size=40, seed=4, verbose=False)
    gr = evaluate.test(model, snr_test_start=0.0, snr_test_end=2.0, snr_points=2, num_block=80, batch_size=40, seed=4, verbose=False, hip_graph=True)
    assert gr["bit_errors"] == res["bit_errors"] and gr["block_errors"] == res["block_errors"]
    for si, snr in enumerate(res["snrs"]):
        c = model.eval_snr(snr, 40, 2, seed=4, first_block=si * 80).sum(dim=0).cpu().tolist()
        assert c == [res["bit_errors"][si], res["block_errors"][si]]
    # r06: every DEC_LargeRNN handle exports its per-stage taps (num_iteration = 1: one stack pair, one tap)
    rx = torch.randn(2, 20, 3, device=gpu_device)
    xd, taps = model.decode_taps(rx)
    assert taps.shape == (1, 2, 20, cfg.num_iter_ft) and bool(torch.isfinite(taps).all()) and float(taps.abs().max()) > 0
    assert torch.equal(xd, model.dec(rx))


def test_range_fallback_reruns_on_fp32_kernels(gpu_device):
    """Activations beyond the fp16 range make the fp16-split results invalid (sticky flag; uncalibrated arithmetic here, so that
    the ceiling is reached at all).  With range_fallback=True the LIBRARY notices (tae_config.range_fallback), runs the call again
    on its fp32 twin and keeps serving from there: a correct result and a flag that says so."""
    from turboae_amd import Channel_AE_HIP
    cfg = TurboAEConfig(enc_num_unit=32, dec_num_unit=32, num_iteration=2, block_len=40, range_calibration=False)
    sd = W.generate_state_dict(cfg, seed=11, gain=1.0)
    big = {k: (v * np.float32(40.0) if ".cnns." in k and k.endswith("weight") and k.startswith("dec.") else v) for k, v in sd.items()}
    B = 4
    u = philox.random_bits(5, 0, B * 40).reshape(B, 40, 1)
    noise = (np.float32(0.8) * philox.random_normal(5, 0, B * 40 * 3)).reshape(B, 40, 3).astype(np.float32)
    xo, co = O.channel_ae_forward(torch.from_numpy(u), torch.from_numpy(noise), O.to_torch(big), cfg.to_dict())
    ut, nt = torch.from_numpy(u).to(gpu_device), torch.from_numpy(noise).to(gpu_device)
    plain = Channel_AE_HIP(cfg, big, device=gpu_device, max_batch=B)
    plain(ut, nt)
    assert plain.range_status() == ("f16x2", True)                 # the fp16-split kernels report, the caller has to act
    model = Channel_AE_HIP(cfg, big, device=gpu_device, max_batch=B, range_fallback=True)
    xd, codes = model(ut, nt)
    assert model.fell_back and model.range_status() == ("f16x2", False)      # the handle's own arithmetic; fell_back says fp32 served
    rel = float(np.abs(xd.cpu().numpy() - xo.numpy()).max())
    assert rel <= 1e-3, rel                                        # logits of order 1e5: compare the saturated outputs
    assert np.array_equal(xd.cpu().numpy() > 0.5, xo.numpy() > 0.5)
    assert np.abs(codes.cpu().numpy() - co.numpy()).max() <= 1e-5
    xd2, _ = model(ut, nt)                                          # stays on the fp32 kernels
    assert torch.equal(xd2, xd)


def test_generic_path_takes_batches_beyond_the_grid_y_limit(gpu_device):
    """gen_conv_kernel rides the block index in grid.y (HIP: <= 65535).  70 000 tiny blocks on a configuration only the
    generic kernels run (width 125): the call succeeds, the blocks past the limit equal a small call on the same blocks bit for bit
    (blocks never interact), and a subsample matches the oracle."""
    from turboae_amd import Channel_AE_HIP
    cfg = TurboAEConfig(block_len=6, enc_num_unit=125, dec_num_unit=125, enc_num_layer=1, dec_num_layer=1, num_iteration=1)
    assert cfg.generic
    sd = W.generate_state_dict(cfg, seed=31, gain=1.0)
    B = 70000
    model = Channel_AE_HIP(cfg, sd, device=gpu_device, max_batch=B)
    u, noise = model.generate_inputs(B, 2.0, seed=3)
    x_dec, codes = model(u, noise)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(x_dec).all())
    rx = codes + noise
    for lo, hi in ((0, 9), (65530, 65545), (B - 5, B)):
        assert torch.equal(model.dec(rx[lo:hi].contiguous()), x_dec[lo:hi]), (lo, hi)
    idx = torch.tensor([0, 65534, 65535, 65536, B - 1], device=gpu_device)
    p = torch.from_numpy(O.rand_interleaver(cfg.block_len, 0))
    with torch.no_grad():
        xo = O.decode(rx[idx].cpu(), O.to_torch(sd), p, cfg.dec_num_layer, cfg.num_iteration, cfg.num_iter_ft)
    assert float((x_dec[idx].cpu() - xo).abs().max()) <= 2e-5
