"""The library's test-time noise generator (tae_generate_noise: HIP kernels, what evaluate.test and tae_eval_snr draw from for every
non-AWGN channel) against (1) the statistics of the REFERENCE's own generate_noise (tests/golden/channel_stats.json): the same checks
as tests/test_channels_cpu.py - the Gilbert-Elliott chains included - and (2) value by value against the numpy mirror of the draw
(turboae_amd/channels.py): masks and chain states exactly, real-valued noise to one fp32 rounding."""
import math

import numpy as np
import pytest
import torch

from turboae_amd import TurboAEConfig, channels
from tests.test_channels_cpu import REF, _lag1

pytestmark = pytest.mark.gpu


from turboae_amd import weights as W

_MODELS = {}


def _model(dev, L, channel, args):
    """a small network (the generator only needs the handle's block length and channel options)"""
    from turboae_amd import Channel_AE_HIP
    key = (L, channel, tuple(sorted(args.items())))
    if key not in _MODELS:
        cfg = TurboAEConfig(block_len=L, enc_num_unit=32, dec_num_unit=32, enc_num_layer=1, dec_num_layer=1, num_iteration=1,
                            channel=channel, **args)
        _MODELS[key] = Channel_AE_HIP(cfg, W.generate_state_dict(cfg, seed=1, gain=1.0), device=dev, max_batch=8)
    return _MODELS[key]


def _draw(case, dev, seed=3):
    B, L, _ = case["shape"]
    model = _model(dev, L, case["channel"], case["args"])
    x, fading = model.generate_noise(B, case["test_sigma"], seed=seed)
    assert x.device.type == "cuda" and x.dtype == torch.float32 and tuple(x.shape) == tuple(case["shape"])
    assert (fading is not None) == (case["channel"] == "fading")
    return x.double().cpu().numpy()


MIRROR = [("awgn", 1.0, {}), ("t-dist", 0.5, {}), ("t-dist", 2.0, {"vv": 3.0}), ("radar", 1.0, {}), ("radar", 3.0, {"radar_prob": 0.2, "radar_power": 2.0}),
          ("ge_awgn", 2.0, {}), ("bec", 0.2, {}), ("bsc", 0.1, {}), ("ge", 0.3, {}), ("ge", 0.0, {}), ("fading", 1.0, {})]


@pytest.mark.parametrize("channel,sig,args", MIRROR, ids=lambda v: str(v))
def test_device_draw_equals_the_numpy_mirror(gpu_device, channel, sig, args):
    """same (seed, global block index) -> same values on the device and in turboae_amd/channels.py; a shard drawn with first_block
    equals the slice of the full draw (what lets ranks shard a batch)"""
    B, L = 37, 101
    model = _model(gpu_device, L, channel, args)
    x, fading = model.generate_noise(B, sig, seed=12345, first_block=7)
    host = channels.generate_noise((B, L, 3), model.cfg, sig, seed=12345, first_block=7)
    xd = x.cpu().numpy()
    if channel in ("bec", "bsc", "ge"):
        assert np.array_equal(xd, host)
    else:
        # fp64 libm on host and device may differ in the last place before the single rounding to fp32: allow 2 ulp
        assert np.all(np.abs(xd - host) <= 2.4e-7 * np.maximum(1.0, np.abs(host))), float(np.abs(xd - host).max())
        assert float((xd != host).mean()) < 1e-3
    if channel == "fading":
        fh = channels.rayleigh_fading((B, L, 3), seed=12345, first_block=7)
        assert np.all(np.abs(fading.cpu().numpy() - fh) <= 2.4e-7 * np.maximum(1.0, fh))
    sub, _ = model.generate_noise(5, sig, seed=12345, first_block=7 + 11)
    assert torch.equal(sub, x[11:16])
    if channel == "awgn":      # the same stream tae_generate_inputs writes
        _, n2 = model.generate_inputs(B, sig, seed=12345, first_block=7)
        assert torch.equal(n2, x)


def test_bad_generator_arguments_are_rejected(gpu_device):
    from turboae_amd._lib import TurboAEError
    with pytest.raises(TurboAEError, match="probability"):
        _model(gpu_device, 16, "bec", {}).generate_noise(2, 1.5, seed=1)
    with pytest.raises(TurboAEError, match="vv > 2"):
        _model(gpu_device, 16, "t-dist", {"vv": 2.0}).generate_noise(2, 1.0, seed=1)


@pytest.mark.parametrize("name", sorted(REF))
def test_device_generators_match_reference_statistics(gpu_device, name):
    case, st = REF[name], REF[name]["stats"]
    x = _draw(case, gpu_device)
    n = st["n"]
    if case["channel"] in ("bec", "bsc", "ge"):
        assert set(np.unique(x).tolist()) <= {0.0, 1.0}
        p = st["frac_one"]
        tol = 5.0 * math.sqrt(2.0 * max(p * (1 - p), 1e-4) / n)
        assert float((x == 1.0).mean()) == pytest.approx(p, abs=tol)
        assert float(x[:, 0, :].mean()) == pytest.approx(st["first_pos_mean"], abs=5.0 * math.sqrt(2.0 * 0.25 / (n / case["shape"][1])) if 0 < st["first_pos_mean"] < 1 else 1e-12)
        assert float(x[:, 1:, :].mean()) == pytest.approx(st["rest_mean"], abs=tol)
        if st["var"] > 0 and x.std() > 0:
            assert _lag1(x) == pytest.approx(st["lag1"], abs=5.0 * math.sqrt(2.0 / n))
        return
    sd = math.sqrt(st["var"])
    assert float(x.mean()) == pytest.approx(st["mean"], abs=5.0 * sd * math.sqrt(2.0 / n))
    for q, rel in (("abs_q50", 0.02), ("abs_q90", 0.02), ("abs_q99", 0.04)):
        assert float(np.quantile(np.abs(x), float(q[5:]) / 100.0)) == pytest.approx(st[q], rel=rel), q
    if case["channel"] in ("awgn", "ge_awgn"):
        assert float(x.var()) == pytest.approx(st["var"], rel=5.0 * math.sqrt(4.0 * st["kurt"] / 3.0 / n))
        nf = n / case["shape"][1]
        assert float(x[:, 0, :].var()) == pytest.approx(st["first_pos_var"], rel=5.0 * math.sqrt(4.0 / nf))
        assert float(x[:, 1:, :].var()) == pytest.approx(st["rest_var"], rel=5.0 * math.sqrt(4.0 * st["kurt"] / 3.0 / n))
    assert _lag1(x * x) == pytest.approx(st["lag1"], abs=5.0 * math.sqrt(2.0 / n) + 0.004)
