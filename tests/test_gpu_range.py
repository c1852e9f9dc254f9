"""Scale-invariance of the default (fp16-split) arithmetic - VERDICT r03 item 1.

The reference convolves in fp32 (cnn_utils.py:36-46: F.conv1d on fp32 tensors): 24 significant bits at ANY magnitude.  An fp16
hi/lo pair has them only inside a window, so the library stores every panel times a calibrated per-layer power of two
(include/turboae_hip.h: tae_config.range_calibration, tae_calibrate_range) and watches both ends of the window at run time.
These tests walk networks whose activations are small, large, or alternate between both, against the oracle in FLOAT64, with the
same bound as test_precision_modes_against_fp64_oracle and NO tolerance widening; the same networks fail it without the calibration.
"""
import numpy as np
import pytest
import torch

from turboae_amd import TurboAEConfig, philox, weights as W, _lib
from oracle import turboae_oracle as O
from tests._fuzz_cases import range_cases, range_case_weights

pytestmark = pytest.mark.gpu

CASES = range_cases()


def _inputs(B, L, seed=61, snr_db=2.0):
    u = philox.random_bits(seed, 0, B * L).reshape(B, L, 1)
    noise = (np.float32(O.snr_db2sigma(snr_db)) * philox.random_normal(seed, 0, B * L * 3)).reshape(B, L, 3).astype(np.float32)
    return u, noise


def _oracle64(cfg, sd, u, noise):
    sd64 = {k: torch.from_numpy(np.asarray(v, dtype=np.float64)) for k, v in sd.items()}
    x64, c64 = O.channel_ae_forward(torch.from_numpy(u).double(), torch.from_numpy(noise).double(), sd64, cfg.to_dict())
    return x64, c64


def _errors(cfg_kw, sd, u, noise, x64, c64, dev, **extra):
    from turboae_amd import Channel_AE_HIP
    err, words = {}, {}
    ud, nd = torch.from_numpy(u).to(dev), torch.from_numpy(noise).to(dev)
    for prec in ("auto", "f32"):
        cfg = TurboAEConfig(precision=prec, **cfg_kw, **extra)
        if prec == "f32" and (cfg.dense or max(cfg.enc_kernel_size, cfg.dec_kernel_size) > 5):
            pass                                   # the generic fp32 kernels take these
        model = Channel_AE_HIP(cfg, sd, device=dev, max_batch=u.shape[0])
        xd, codes = model(ud, nd)
        words[prec] = model._eng.range_word()
        err[prec] = (float((codes.cpu().double() - c64).abs().max()), float((xd.cpu().double() - x64).abs().max()))
    return err, words


@pytest.mark.parametrize("name,cfg_kw,wseed,spec", CASES, ids=[c[0] for c in CASES])
def test_fp16_split_is_fp32_grade_at_any_activation_scale(gpu_device, name, cfg_kw, wseed, spec):
    base = TurboAEConfig(**cfg_kw)
    sd = range_case_weights(base, wseed, spec)
    B = 4 if base.block_len >= 1000 else 12
    u, noise = _inputs(B, base.block_len)
    x64, c64 = _oracle64(base, sd, u, noise)
    err, words = _errors(cfg_kw, sd, u, noise, x64, c64, gpu_device)
    print(name, "max |err| vs fp64 oracle (codes, x_dec):", err, words)
    assert words["auto"] == ("f16x2", 0), words            # inside the window on both sides
    for k in (0, 1):
        assert err["auto"][k] <= 2.0 * err["f32"][k] + 5e-7, (name, err)


@pytest.mark.parametrize("gain", [1.0, 0.3, 0.1, 0.03])
def test_gru_decoder_at_small_gains(gpu_device, gain):
    """The GRU decoder's fp16-split operands are its hidden states, bounded by tanh: |h| < 1 whatever the weights, so they are
    stored unscaled (2^-25 absolute = 2^-25 of their natural maximum).  Small weight gains (the biases keep PyTorch's scale) must
    stay inside the same bound against the float64 oracle as the CNN path."""
    cfg_kw = dict(decoder="TurboAE_rate3_rnn", num_iteration=2, block_len=40)
    base = TurboAEConfig(**cfg_kw)
    sd = W.generate_state_dict(base, seed=11, gain=gain)
    u, noise = _inputs(16, base.block_len)
    x64, c64 = _oracle64(base, sd, u, noise)
    err, words = _errors(cfg_kw, sd, u, noise, x64, c64, gpu_device)
    print("gru gain", gain, err, words)
    assert words["auto"] == ("f16x2", 0)
    for k in (0, 1):
        assert err["auto"][k] <= 2.0 * err["f32"][k] + 5e-7, (gain, err)


@pytest.mark.parametrize("name", ["balanced_0.01", "balanced_16", "alt_2^-8_2^+8"])
def test_uncalibrated_arithmetic_fails_these_networks(gpu_device, name):
    """The r03 arithmetic (all exponents 0 = range_calibration off) on the same networks: either the range word says so or the error
    bound is missed by a wide margin - i.e. the cases above do test the low and the high side."""
    _, cfg_kw, wseed, spec = [c for c in CASES if c[0] == name][0]
    base = TurboAEConfig(**cfg_kw)
    sd = range_case_weights(base, wseed, spec)
    u, noise = _inputs(12, base.block_len)
    x64, c64 = _oracle64(base, sd, u, noise)
    err, words = _errors(cfg_kw, sd, u, noise, x64, c64, gpu_device, range_calibration=False)
    print(name, err, words)
    bad_bound = any(not (err["auto"][k] <= 2.0 * err["f32"][k] + 5e-7) for k in (0, 1))
    assert bad_bound or (words["auto"][1] & _lib.RANGE_HIGH)


def test_exponents_follow_the_network_scale(gpu_device):
    """Scaling layer l's weights and biases by 2^k moves that layer's measured maximum by 2^k and its exponent by -k (ELU is close to
    linear at the small end), and nothing is measured twice: two passes."""
    from turboae_amd import Channel_AE_HIP
    cfg = TurboAEConfig(num_iteration=1)
    sd = W.generate_state_dict(cfg, seed=7, gain=1.0)
    from tests._fuzz_cases import scale_layers
    a = Channel_AE_HIP(cfg, sd, device=gpu_device, max_batch=8)
    b = Channel_AE_HIP(cfg, scale_layers(sd, cfg, None, [2.0 ** -6, 1.0, 1.0, 1.0, 2.0 ** 6]), device=gpu_device, max_batch=8)
    ea, da, pa = a._eng.range_info()
    eb, db, pb = b._eng.range_info()
    n_stack, nl = 2, cfg.dec_num_layer
    assert len(da) == n_stack + n_stack * nl and ea == eb and pa <= 3 and pb <= 3
    A, Bx = np.array(da[n_stack:]).reshape(n_stack, nl), np.array(db[n_stack:]).reshape(n_stack, nl)
    # the first stack sees identical inputs: its panels 0..3 sit 2^-6 lower (within one binade: ELU saturates the negative side)
    assert np.all(np.abs((Bx[0, :4] - A[0, :4]) - 6) <= 1), (A, Bx)
    assert np.all(A[:, 4] == 0) and np.all(Bx[:, 4] == 0)            # the last layer feeds the Linear head in fp32


def test_range_word_reports_both_ends_and_c_abi_fallback(gpu_device):
    """Data far outside what the handle was calibrated on: received values 2^12 larger raise TAE_RANGE_HIGH, 2^-12 smaller
    TAE_RANGE_LOW (tae_decode, the entry point a caller with its own channel uses); calibrating on that data clears both; a
    range_fallback handle re-runs the flagged call in fp32 inside the library and says so."""
    from turboae_amd import Channel_AE_HIP
    cfg = TurboAEConfig(num_iteration=2)
    sd = W.generate_state_dict(cfg, seed=7, gain=1.0)
    B = 6
    u, noise = _inputs(B, cfg.block_len)
    ud, nd = torch.from_numpy(u).to(gpu_device), torch.from_numpy(noise).to(gpu_device)
    model = Channel_AE_HIP(cfg, sd, device=gpu_device, max_batch=B)
    xd, codes = model(ud, nd)
    assert model._eng.range_word() == ("f16x2", 0)
    rx = codes + nd
    for factor, bit in ((2.0 ** 14, _lib.RANGE_HIGH), (2.0 ** -14, _lib.RANGE_LOW)):
        model.dec(rx * factor)
        mode, bits = model._eng.range_word()
        assert mode == "f16x2" and bits & bit, (factor, bits)
        with pytest.raises(_lib.TurboAEError):
            model.dec(rx * factor)
            model.check_range()
    # calibrating on the caller's own data moves the window there
    sd_o = O.to_torch(sd)
    small = Channel_AE_HIP(cfg, sd, device=gpu_device, max_batch=B)
    small.calibrate_range(ud, nd * 2.0 ** -14)              # codes + tiny noise is still O(1): stays clean
    small(ud, nd * 2.0 ** -14)
    assert small._eng.range_word() == ("f16x2", 0)
    # C-ABI fall-back: same flagged decode, re-run on the fp32 twin
    fb = Channel_AE_HIP(cfg, sd, device=gpu_device, max_batch=B, range_fallback=True)
    x1 = fb.dec(rx * 2.0 ** 14)
    assert fb.fell_back
    mode, bits = fb._eng.range_word()
    assert bits & _lib.RANGE_FELL_BACK
    exact = Channel_AE_HIP(TurboAEConfig(num_iteration=2, precision="f32"), sd, device=gpu_device, max_batch=B)
    x2 = exact.dec(rx * 2.0 ** 14)
    assert torch.equal(x1, x2)                                 # the fp32 kernels' own result
    x3 = fb.dec(rx)                                            # ... which serve every later call
    assert torch.equal(x3, exact.dec(rx))


def test_calibration_does_not_change_in_window_results(gpu_device):
    """Every exponent is a power of two: on an O(1) network the calibrated and the uncalibrated arithmetic agree to the last few
    ulps of fp32 (they differ only where a lo half meets the absolute floor), and two handles calibrate identically."""
    from turboae_amd import Channel_AE_HIP
    cfg = TurboAEConfig()
    sd = W.generate_state_dict(cfg, seed=7, gain=1.0)
    u, noise = _inputs(9, cfg.block_len)
    ud, nd = torch.from_numpy(u).to(gpu_device), torch.from_numpy(noise).to(gpu_device)
    a = Channel_AE_HIP(cfg, sd, device=gpu_device, max_batch=9)
    b = Channel_AE_HIP(cfg, sd, device=gpu_device, max_batch=500)
    raw = Channel_AE_HIP(TurboAEConfig(range_calibration=False), sd, device=gpu_device, max_batch=9)
    xa, ca = a(ud, nd)
    xb, cb = b(ud, nd)
    xr, cr = raw(ud, nd)
    assert a._eng.range_info() == b._eng.range_info()
    assert torch.equal(xa, xb) and torch.equal(ca, cb)
    assert float((ca - cr).abs().max()) <= 2e-6 and float((xa - xr).abs().max()) <= 2e-6
    assert raw._eng.range_info()[:2] == ([], [])


def test_fallback_twin_serves_every_batch_the_handle_accepts(gpu_device):
    """tae_create calibrates on a synthetic batch of up to 768 blocks, which grows the handle's capacity past a
    small max_batch; a C caller's B in (max_batch, 768] then passes check_batch, and a flagged call used to run on an fp32 twin whose
    workspace was sized for max_batch only.  The raw C ABI call below (no tae_reserve) must be served by the twin, equal to an fp32
    handle that was sized for the batch."""
    import ctypes as C
    from turboae_amd import Channel_AE_HIP
    from turboae_amd.channel_ae import _ptr, _stream
    cfg = TurboAEConfig(num_iteration=2)
    sd = W.generate_state_dict(cfg, seed=7, gain=1.0)
    B = 500
    u, noise = _inputs(B, cfg.block_len)
    ud, nd = torch.from_numpy(u).to(gpu_device), torch.from_numpy(noise).to(gpu_device)
    exact = Channel_AE_HIP(TurboAEConfig(num_iteration=2, precision="f32"), sd, device=gpu_device, max_batch=B)
    rx = (exact.enc(ud) + nd) * 2.0 ** 14                      # far above the calibrated window: every call is flagged
    fb = Channel_AE_HIP(cfg, sd, device=gpu_device, max_batch=100, range_fallback=True)
    e = fb._eng
    x1 = e._out(B, 1)
    with torch.cuda.device(gpu_device):
        _lib.check(e.lib.tae_decode(e.h, _ptr(rx), _ptr(x1), B, _stream()))      # B = 500 > max_batch = 100, accepted (cap = 768)
    torch.cuda.synchronize()
    assert fb.fell_back
    assert torch.equal(x1, exact.dec(rx))


def test_index_less_cuda_device_spelling(gpu_device):
    """device='cuda' (no index) is the common spelling; the engine resolves it to the current device once, so
    calibrate_range's device check (tensors report cuda:0) accepts what forward accepts."""
    from turboae_amd import Channel_AE_HIP
    cfg = TurboAEConfig(num_iteration=1, enc_num_unit=32, dec_num_unit=32, block_len=32)
    sd = W.generate_state_dict(cfg, seed=9, gain=1.0)
    with torch.cuda.device(gpu_device):
        for spelling in ("cuda", torch.device("cuda")):
            m = Channel_AE_HIP(cfg, sd, device=spelling, max_batch=4)
            assert m._eng.device == torch.device("cuda", torch.cuda.current_device())
            u, noise = _inputs(4, cfg.block_len)
            ud, nd = torch.from_numpy(u).to("cuda"), torch.from_numpy(noise).to("cuda")
            ref = m(ud, nd)
            m.calibrate_range(ud, nd)
            out = m(ud, nd)
            assert torch.equal(out[0], ref[0])


def test_calibrate_range_takes_what_forward_takes(gpu_device):
    """calibrate_range validates and routes its arguments exactly as forward does - (B, L, 1) punctured noise is
    expanded, channel='fading' needs (and lays out) the coefficients, mismatched or half-given arguments raise instead of reading
    out of bounds, and with is_variable_block_len the engine of the given length is the one calibrated."""
    from turboae_amd import Channel_AE_HIP
    cfg = TurboAEConfig(num_iteration=2)
    sd = W.generate_state_dict(cfg, seed=7, gain=1.0)
    B, L = 6, cfg.block_len
    u, noise = _inputs(B, L)
    ud, nd = torch.from_numpy(u).to(gpu_device), torch.from_numpy(noise).to(gpu_device)
    m = Channel_AE_HIP(cfg, sd, device=gpu_device, max_batch=B, is_variable_block_len=True)
    ref = m(ud, nd)
    with pytest.raises(ValueError):
        m.calibrate_range(ud)                                  # noise missing
    with pytest.raises(ValueError):
        m.calibrate_range(None, nd)
    with pytest.raises(ValueError):
        m.calibrate_range(ud, nd[:3])                          # batch mismatch
    with pytest.raises(ValueError):
        m.calibrate_range(ud, nd, fading=nd)                   # not a fading channel
    m.calibrate_range(ud, nd[:, :, :1])                        # punctured pass: (B, L, 1), broadcast over the code symbols
    m.calibrate_range(ud, nd)
    m.calibrate_range()                                        # synthetic batch again
    out = m(ud, nd)
    assert torch.equal(out[0], ref[0]) and torch.equal(out[1], ref[1])      # O(1) network: same power-of-two window
    # another block length: its own engine is calibrated, on its own shapes
    L2 = 64
    u2, n2 = _inputs(B, L2, seed=62)
    u2d, n2d = torch.from_numpy(u2).to(gpu_device), torch.from_numpy(n2).to(gpu_device)
    m.calibrate_range(u2d, n2d)
    assert L2 in m._by_len and m._by_len[L2].range_word() == ("f16x2", 0)
    m(u2d, n2d)
    assert m._by_len[L2].range_word() == ("f16x2", 0)
    # fading: the library reads [fading | noise]; calibrate_range builds that layout like forward
    cf = TurboAEConfig(num_iteration=2, channel="fading")
    mf = Channel_AE_HIP(cf, sd, device=gpu_device, max_batch=B)
    nz, fh = mf.generate_noise(B, 2.0, seed=5)
    with pytest.raises(ValueError):
        mf.calibrate_range(ud, nz)                             # coefficients missing
    mf.calibrate_range(ud, nz, fading=fh)
    mf(ud, nz, fh)
    assert mf._eng.range_word() == ("f16x2", 0)


def test_user_calibration_survives_interleaver_and_channel_changes(gpu_device):
    """tae_set_interleaver / tae_set_channel_opts used to re-run the synthetic calibration every time - with
    -is_same_interleaver 0 on every forward - and silently replaced a calibration made on the caller's data.  Now: the synthetic
    calibration is repeated once, with the first installed permutation; later permutations keep it; a user calibration stays until
    the caller replaces it."""
    from turboae_amd import Channel_AE_HIP
    from turboae_amd.interleaver import rand_interleaver
    cfg = TurboAEConfig(num_iteration=2)
    sd = W.generate_state_dict(cfg, seed=7, gain=1.0)
    B, L = 6, cfg.block_len
    u, noise = _inputs(B, L)
    ud, nd = torch.from_numpy(u).to(gpu_device), torch.from_numpy(noise).to(gpu_device)
    m = Channel_AE_HIP(cfg, sd, device=gpu_device, max_batch=B)
    m(ud, nd)                                             # installs the seed-0 permutation: the one synthetic re-measurement
    synthetic = m._eng.range_info()
    m._eng.set_interleaver(rand_interleaver(L, 5))        # another permutation: kept
    assert m._eng.range_info() == synthetic
    big = nd * 64.0                                       # the caller's channel is 36 dB noisier than the synthetic batch's
    m.calibrate_range(ud, big)
    user = m._eng.range_info()
    assert user[1] != synthetic[1]                        # the decoder's exponents moved to the caller's data
    m._eng.set_interleaver(rand_interleaver(L, 6))
    assert m._eng.range_info() == user
    m(ud, big)
    assert m._eng.range_word() == ("f16x2", 0)            # in-window on that data; the synthetic window would have raised HIGH
    m.calibrate_range()                                   # back to the synthetic batch on request
    assert m._eng.range_info()[1] == synthetic[1]


def test_range_rows_beyond_one_per_thread(gpu_device):
    """range_finish gave one thread to each (stack, layer) row and stopped at the workgroup's 512 threads; a decoder
    with 2 x 52 iterations x 5 layers = 520 rows left the last ones unchecked and uncalibrated (exponent 0).  Small-activation
    network: every panel row, the last stacks' included, gets a non-zero exponent, and the result is fp32-grade."""
    from turboae_amd import Channel_AE_HIP
    from tests._fuzz_cases import scale_layers, balanced
    cfg = TurboAEConfig(num_iteration=52, enc_num_unit=32, dec_num_unit=32, block_len=12)
    sd = scale_layers(W.generate_state_dict(cfg, seed=11, gain=1.0), cfg, balanced(cfg.enc_num_layer, 0.05), balanced(cfg.dec_num_layer, 0.05))
    B = 3
    u, noise = _inputs(B, cfg.block_len)
    m = Channel_AE_HIP(cfg, sd, device=gpu_device, max_batch=B)
    enc_a, dec_a, _ = m._eng.range_info()
    n_stack, nl = 2 * cfg.num_iteration, cfg.dec_num_layer
    panels = dec_a[len(dec_a) - n_stack * nl:]            # one exponent per (stack, layer) behind the stack-input exponents
    assert len(panels) == 520
    for s in range(n_stack):
        assert all(panels[s * nl + l] != 0 for l in range(nl - 1)), (s, panels[s * nl:(s + 1) * nl])       # layers 0..3 hold ~0.05^k: scaled up
    xd, codes = m(torch.from_numpy(u).to(gpu_device), torch.from_numpy(noise).to(gpu_device))
    assert m._eng.range_word() == ("f16x2", 0)
    x64, c64 = _oracle64(cfg, sd, u, noise)
    assert float((codes.cpu().double() - c64).abs().max()) <= 5e-6 and float((xd.cpu().double() - x64).abs().max()) <= 2e-5
