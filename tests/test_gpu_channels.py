"""The test-time noise generators ON THE DEVICE (what evaluate.test draws from for every non-AWGN channel) against the statistics
of the REFERENCE's own generate_noise (tests/golden/channel_stats.json): same checks as tests/test_channels_cpu.py, with the torch
generator and the tensors on the GPU - the Gilbert-Elliott chains included (the GPU sweep test only checked that BER
falls with SNR)."""
import math

import numpy as np
import pytest
import torch

from turboae_amd import TurboAEConfig, channels
from tests.test_channels_cpu import REF, _lag1

pytestmark = pytest.mark.gpu


def _draw(case, dev, seed=3):
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    cfg = TurboAEConfig(channel=case["channel"], **case["args"])
    x = channels.generate_noise(tuple(case["shape"]), cfg, case["test_sigma"], device=dev, generator=g)
    assert x.device.type == "cuda" and x.dtype == torch.float32 and tuple(x.shape) == tuple(case["shape"])
    return x.double().cpu().numpy()


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
