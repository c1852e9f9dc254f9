"""BASELINE.json's FULL sizes under pytest (the driver runs `-m gpu`): configs[1] 50 000 x 100, configs[2] enc5/dec5
100 000 x 100, configs[3] per-GPU shape 25 000 x 1000.  The oracle cannot run these sizes in seconds, so each case checks
  * an ORACLE SUBSAMPLE: blocks strided across the whole batch (first, last, every k-th) - encoder output before the
    power constraint and decoder output against oracle/turboae_oracle.py on exactly those blocks (blocks only couple
    through the batch statistics, which are taken from the full-size run);
  * size-independent properties: codes have mean 0 / unbiased std 1 over the whole batch (encoders.py:107-116);
    decoding any sub-range reproduces the big call bit for bit (head / middle / tail: any 32-bit index overflow breaks the
    tail); the encode -> AWGN -> decode round trip of the trained network recovers the bits at high SNR; error counts
    equal a torch recount.
Bounded to well under a minute on an MI355X box."""
import os

import numpy as np
import pytest
import torch

from turboae_amd import TurboAEConfig, weights as W
from oracle import turboae_oracle as O
from _tol import ATOL_XDEC_RNN, note

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _trained_sd(cfg=None, fname="trained_enc2dec5_u100_fp32.npz"):
    return W.unpack_blob(cfg or TurboAEConfig(), np.load(os.path.join(GOLD, fname))["weights_fp32"])


def _manifest(key):
    import json
    with open(os.path.join(GOLD, "MANIFEST.json")) as fh:
        return json.load(fh)[key]


def _check_full_size(dev, cfg, sd, B, snr, n_sub, trained):
    from turboae_amd import Channel_AE_HIP
    L = cfg.block_len
    model = Channel_AE_HIP(cfg, sd, device=dev, max_batch=B)
    u, noise = model.generate_inputs(B, snr, seed=77)
    x_dec, codes = model(u, noise)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(x_dec).all()) and bool(torch.isfinite(codes).all())
    m, s = float(codes.double().mean()), float(codes.double().std())
    assert abs(m) < 1e-6 and abs(s - 1.0) < 1e-6, (m, s)                       # power constraint over the WHOLE batch
    # split form reproduces the fused call; its pre-normalisation output feeds the oracle subsample
    x_tx, stats = model.encode_prenorm(u)
    codes2, rx = model.normalize(x_tx, stats, noise)
    assert torch.equal(codes2, codes)
    assert float(stats[2]) == float(B) * L * 3
    # decoder sub-ranges bit for bit (blocks never interact; index arithmetic at the far end of the buffers)
    for lo, hi in ((0, 7), (B // 2 - 3, B // 2 + 5), (B - 9, B)):
        assert torch.equal(model.dec(rx[lo:hi].contiguous()), x_dec[lo:hi]), (lo, hi)
    # error counts: library kernel vs a torch recount
    counts = model.count_errors(x_dec, u).cpu().tolist()
    err = (x_dec > 0.5) != (u > 0.5)
    assert counts == [int(err.sum()), int(err.any(dim=1).sum())]
    # oracle subsample
    idx = np.unique(np.concatenate([np.linspace(0, B - 1, n_sub).astype(np.int64), [0, 1, B - 2, B - 1]]))
    ti = torch.from_numpy(idx).to(dev)
    w = O.to_torch(sd)
    p = torch.from_numpy(O.rand_interleaver(L, 0))
    with torch.no_grad():
        xo = O.encode_prenorm(u[ti].cpu(), w, p, cfg.enc_num_layer)
        assert float((x_tx[ti].cpu() - xo).abs().max()) <= 1e-5
        xd_o = O.decode(rx[ti].cpu(), w, p, cfg.dec_num_layer, cfg.num_iteration, cfg.num_iter_ft)
    d = float((x_dec[ti].cpu() - xd_o).abs().max())
    assert d <= 2e-5, d
    model.check_range()
    ber = counts[0] / (float(B) * L)
    if trained:
        # round trip: the trained network recovers (almost) every bit at this SNR (reference BER 3.6e-4 at 6 dB, < 2e-5 at 8 dB)
        assert ber < 1e-4, ber
    return ber


def test_configs1_50000_blocks_of_100(gpu_device):
    cfg = TurboAEConfig()
    ber = _check_full_size(gpu_device, cfg, _trained_sd(), 50000, 8.0, 120, trained=True)
    print("configs[1] 50 000 x 100 @ 8 dB: BER", ber)


def test_configs1_small_last_layers_run_the_both_branch_twin_at_full_size(gpu_device):
    """a network whose last conv layers stay below 1/4 needs both expm1 branches in its Linear heads.  Through r04
    every launch of such a network ran the calibration instantiation (dec_kernel_h<..., true>: 200 bytes of scratch, 108 spilled
    registers); since r05 it has a production twin (dec_kernel_h<100, 5, false, true> / enc_kernel_h<100, 5, 0, true>: 252 / 221 VGPRs,
    no scratch).  The trained network with every last layer scaled by 2^-5 (Linear heads by 2^5): the handle reports the twin on
    both sides and 50 000 blocks pass the full-size checks against the oracle (what a launch costs: bench.py)."""
    from turboae_amd import Channel_AE_HIP
    cfg = TurboAEConfig()
    sd = W.scale_last_layers(_trained_sd(), cfg, 2.0 ** -5)
    B = 50000
    ber = _check_full_size(gpu_device, cfg, sd, B, 8.0, 60, trained=False)
    print("configs[1] shape, last layers x 2^-5, 50 000 x 100 @ 8 dB: BER", ber)
    small = Channel_AE_HIP(cfg, sd, device=gpu_device, max_batch=B)
    plain = Channel_AE_HIP(cfg, _trained_sd(), device=gpu_device, max_batch=B)
    assert small.kernel_variants() == (True, True) and plain.kernel_variants() == (False, False)
    # timing of the twin against the plain kernel is bench.py's business (`cfg1_head2_decoder_over_plain`): no wall-clock assertion
    # lives under tests/ (VERDICT r05 item 4); _check_full_size above ran the twin against the oracle subsample


def test_configs1_operating_point_2dB(gpu_device):
    """The benchmark's own operating point at full size: BER of 5e6 bits at 2 dB sits where the reference measured this network
    (MANIFEST trained_fp32: 1.47e-2 on 2e5 bits)."""
    import json
    from turboae_amd import Channel_AE_HIP
    cfg = TurboAEConfig()
    B = 50000
    model = Channel_AE_HIP(cfg, _trained_sd(), device=gpu_device, max_batch=B)
    u, noise = model.generate_inputs(B, 2.0, seed=20190001)
    x_dec, _ = model(u, noise)
    counts = model.count_errors(x_dec, u).cpu().tolist()
    ber = counts[0] / (B * 100.0)
    with open(os.path.join(GOLD, "MANIFEST.json")) as fh:
        ref = json.load(fh)["trained_fp32"]["ber"]["2dB"]
    assert abs(ber - ref) <= 0.1 * ref, (ber, ref)


def test_configs2_enc5_dec5_100000_blocks(gpu_device):
    """BASELINE configs[2] on its reference-trained fixture (tests/golden/trained_enc5dec5_u100_fp32.npz): the 8 dB round trip
    recovers (almost) every bit, and 10^7 bits at 2 dB sit at the BER the reference measured for this network on 2e5 bits."""
    from turboae_amd import Channel_AE_HIP
    cfg = TurboAEConfig(enc_num_layer=5)
    sd = _trained_sd(cfg, "trained_enc5dec5_u100_fp32.npz")
    _check_full_size(gpu_device, cfg, sd, 100000, 8.0, 60, trained=True)
    model = Channel_AE_HIP(cfg, sd, device=gpu_device, max_batch=100000)
    u, noise = model.generate_inputs(100000, 2.0, seed=20190001)
    x_dec, _ = model(u, noise)
    ber = model.count_errors(x_dec, u).cpu().tolist()[0] / 1e7
    ref = _manifest("trained_enc5dec5_fp32")["ber"]["2dB"]
    print("configs[2] 100 000 x 100 @ 2 dB: BER", ber, "reference", ref)
    assert abs(ber - ref) <= 0.1 * ref, (ber, ref)
    model.check_range()


def test_configs3_per_gpu_shape_25000_blocks_of_1000(gpu_device):
    # conv weights do not depend on the block length: the L = 100-trained network on 1000-bit blocks (long-block kernels)
    cfg = TurboAEConfig(block_len=1000)
    # (no round-trip BER bound here: the network was trained with the 100-position interleaver)
    ber = _check_full_size(gpu_device, cfg, _trained_sd(), 25000, 8.0, 12, trained=False)
    print("configs[3] 25 000 x 1000 @ 8 dB: BER", ber)


@pytest.mark.parametrize("cell", ["gru", "lstm"])
def test_configs4_gru_decoder_16384_blocks(gpu_device, cell):
    """BASELINE configs[4] (DeepTurbo GRU decoder) at one full wave of recurrent workgroups: 16 384 blocks (256 CUs x 8 waves x
    16 blocks / 2 directions), the GRU path's internal chunk size - so this also runs the chunk boundary when 16 400 are given.
    cell = lstm: the same shape with -dec_rnn lstm (decoders.py:27-32) on the unit-split kernels of turboae_rnn_u.hip."""
    from turboae_amd import Channel_AE_HIP
    cfg = TurboAEConfig(decoder="TurboAE_rate3_rnn", dec_rnn=cell)
    sd = _trained_sd(cfg, f"trained_cnn_{cell}_u100_fp32.npz")       # reference-trained recurrent decoder behind the trained enc2 encoder
    B, L = 16400, cfg.block_len
    model = Channel_AE_HIP(cfg, sd, device=gpu_device, max_batch=B)
    u, noise = model.generate_inputs(B, 2.0, seed=78)
    x_dec, codes = model(u, noise)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(x_dec).all())
    m, s = float(codes.double().mean()), float(codes.double().std())
    assert abs(m) < 1e-6 and abs(s - 1.0) < 1e-6
    rx = codes + noise
    for lo, hi in ((0, 5), (16380, 16390), (B - 7, B)):          # inside chunk 0, across the chunk boundary, the tail of chunk 1
        assert torch.equal(model.dec(rx[lo:hi].contiguous()), x_dec[lo:hi]), (lo, hi)
    idx = np.array([0, 1, 8191, 16383, 16384, B - 1])
    w = O.to_torch(sd)
    p = torch.from_numpy(O.rand_interleaver(L, 0))
    with torch.no_grad():
        xd_o = O.decode_rnn(rx[torch.from_numpy(idx).to(gpu_device)].cpu(), w, p, cfg.dec_num_unit, cfg.num_iteration, cfg.num_iter_ft, cell=cell)
    d = float((x_dec[torch.from_numpy(idx).to(gpu_device)].cpu() - xd_o).abs().max())
    assert note(f"fullsize:{cell}:16400", d) <= ATOL_XDEC_RNN, d          # the recurrent golden tests' tolerance (tests/_tol.py)
    counts = model.count_errors(x_dec, u).cpu().tolist()
    err = (x_dec > 0.5) != (u > 0.5)
    assert counts == [int(err.sum()), int(err.any(dim=1).sum())]
    # an operating point: 1.64e6 bits at 2 dB against the BER the reference measured for this network on 2e5 bits
    ber, ref = counts[0] / (float(B) * L), _manifest(f"trained_cnn_{cell}_fp32")["ber"]["2dB"]
    print(f"configs[4] ({cell}) 16 400 x 100 @ 2 dB: BER", ber, "reference", ref)
    assert abs(ber - ref) <= 0.1 * ref, (ber, ref)
    model.check_range()


def test_configs1_twelve_point_ber_sweep(gpu_device):
    """BASELINE configs[1] as it is quoted: the BER sweep -1.5 .. 4 dB in 12 points with 50 000-block batches, reference-trained
    network, through evaluate.test (trainer.test restated) in both arithmetics, eager and as one hipGraph per SNR point.
      * the sweep's error counts equal a forward + count of the same Philox blocks made by hand (per point, exactly);
      * the fp16-split and the fp32-MFMA sweeps differ by a handful of decisions out of 5e6 per point;
      * an ORACLE SUBSAMPLE per point (blocks strided over the batch; batch statistics from the full-size run) has the GPU's
        hard decisions;
      * BER falls monotonically and sits at the reference-measured values of this network where the fixture has them.
    TAE_SWEEP_OUT=<file>: also write the table (profiles/r02_sweep_cfg1.json comes from this)."""
    import json
    import time
    from dataclasses import replace
    from turboae_amd import Channel_AE_HIP, evaluate
    from turboae_amd.distributed import mean_std_from_stats
    cfg, sd = TurboAEConfig(), _trained_sd()
    B, L, SEED, NP = 50000, 100, 20190928, 12
    sweep = dict(snr_test_start=-1.5, snr_test_end=4.0, snr_points=NP, num_block=B, batch_size=B, seed=SEED, verbose=False,
                 enc_power_epilogue=False)
    res, secs = {}, {}
    for name, prec, graph in (("f16x2", "auto", False), ("f16x2_hipgraph", "auto", True), ("f32", "f32", False)):
        model = Channel_AE_HIP(replace(cfg, precision=prec), sd, device=gpu_device, max_batch=B)
        evaluate.test(model, **{**sweep, "snr_points": 1, "snr_test_end": -1.5}, hip_graph=graph)      # warm-up (workspace, graph pools)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res[name] = evaluate.test(model, **sweep, hip_graph=graph)
        torch.cuda.synchronize()
        secs[name] = time.perf_counter() - t0
        model.check_range()
    a, g, f = res["f16x2"], res["f16x2_hipgraph"], res["f32"]
    assert g["bit_errors"] == a["bit_errors"] and g["block_errors"] == a["block_errors"]
    gaps = [abs(x - y) for x, y in zip(a["bit_errors"], f["bit_errors"])]
    assert max(gaps) <= 8, gaps                                             # of 5e6 decisions per point
    assert all(x > y for x, y in zip(a["ber"], a["ber"][1:])), a["ber"]
    with open(os.path.join(GOLD, "MANIFEST.json")) as fh:
        ref = json.load(fh)["trained_fp32"]["ber"]
    for snr, key in ((2.0, "2dB"), (4.0, "4dB")):                           # the reference's own BER of this network (2e5 bits each)
        mine = a["ber"][a["snrs"].index(snr)]
        assert abs(mine - ref[key]) <= 5.0 * (ref[key] * 20.0 / 2e5) ** 0.5, (snr, mine, ref[key])      # ~20 errors per bad block
    # by hand + oracle subsample, point by point
    model = Channel_AE_HIP(cfg, sd, device=gpu_device, max_batch=B)
    w, p = O.to_torch(sd), torch.from_numpy(O.rand_interleaver(L, 0))
    idx = torch.from_numpy(np.unique(np.concatenate([np.linspace(0, B - 1, 160).astype(np.int64), [1, B - 2]])))
    flips, worst = [], 0.0
    for si, snr in enumerate(a["snrs"]):
        u, noise = model.generate_inputs(B, snr, seed=SEED, first_block=si * B)
        x_tx, stats = model.encode_prenorm(u)
        _, rx = model.normalize(x_tx, stats, noise, want_codes=False)
        x_dec = model.dec(rx)
        assert model.count_errors(x_dec, u).cpu().tolist() == [a["bit_errors"][si], a["block_errors"][si]], snr
        mean, std = mean_std_from_stats(stats)
        ti = idx.to(gpu_device)
        with torch.no_grad():
            xo = O.encode_prenorm(u[ti].cpu(), w, p, cfg.enc_num_layer)
            rxo = (xo - np.float32(mean)) / np.float32(std) + noise[ti].cpu()
            xd_o = O.decode(rxo, w, p, cfg.dec_num_layer, cfg.num_iteration, cfg.num_iter_ft)
        worst = max(worst, float((x_dec[ti].cpu() - xd_o).abs().max()))
        diff = (x_dec[ti].cpu() > 0.5) != (xd_o > 0.5)
        flips.append(int(diff.sum()))
        assert not bool((diff & ((xd_o - 0.5).abs() > 1e-4)).any()), snr      # only a bit within fp32 noise of the threshold may differ
    assert worst <= 2e-5 and sum(flips) <= 2, (worst, flips)
    out = os.environ.get("TAE_SWEEP_OUT")
    if out:
        bits = float(NP) * B * L
        table = {"workload": "BASELINE configs[1]: enc2/dec5 reference-trained (tests/golden/trained_enc2dec5_u100_fp32.npz), block_len 100, "
                             "12 SNR points -1.5 .. 4 dB, one 50 000-block batch per point, evaluate.test (device Philox inputs, seed %d)" % SEED,
                 "snr_db": a["snrs"],
                 "f16x2": {k: a[k] for k in ("ber", "bler", "bit_errors", "block_errors")},
                 "f32": {k: f[k] for k in ("ber", "bler", "bit_errors", "block_errors")},
                 "bit_error_count_gap_f16x2_vs_f32": gaps,
                 "oracle_subsample": {"blocks_per_point": int(idx.numel()), "decision_flips_per_point": flips, "max_abs_x_dec": worst},
                 "sweep_seconds": secs, "info_bits_per_second": {k: bits / v for k, v in secs.items()},
                 "reference_ber_of_this_network": ref, "device": torch.cuda.get_device_name(0)}
        with open(out, "w") as fh:
            json.dump(table, fh, indent=1)
