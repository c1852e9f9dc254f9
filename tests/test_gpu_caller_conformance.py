"""Drop-in conformance under the REFERENCE's own eval loop.

tests/golden/caller_trainer_test.{json,npz} (oracle/make_caller_fixture.py) hold what the reference's unmodified
``trainer.test(model, args)`` (trainer.py:135-248) did to a recording proxy around a real reference ``Channel_AE``:
every attribute it read, every call with its tensors, and the stdout transcript.  Here ``Channel_AE_HIP`` must serve
that exact sequence - same attribute surface, same argument shapes (including the (B, L, 1) noise tensor of the punctured
pass), same outputs - and the numbers the reference printed must come out of the replayed outputs; ``evaluate.test`` (the
restated loop) must print the same line sequence."""
import contextlib
import io
import json
import os
import re

import numpy as np
import pytest
import torch

from turboae_amd import TurboAEConfig, weights as W

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
with open(os.path.join(GOLD, "caller_trainer_test.json")) as _fh:
    FIX = json.load(_fh)

NUM = re.compile(r"[-+]?(?:\d+\.\d*|\.\d+|\d+)(?:[eE][-+]?\d+)?")


def _shape(line_list):
    """Transcript -> one string with every number replaced by '#' and whitespace collapsed (array prints wrap by value width)."""
    return re.sub(r"(?:# ?)+", "# ", re.sub(r"\s+", " ", NUM.sub("#", " ".join(line_list)))).replace("# ]", "#]").strip()


def _model(dev, B, args):
    from turboae_amd import Channel_AE_HIP
    cfg = TurboAEConfig(precompute_norm_stats=bool(args["precompute_norm_stats"]))
    sd = W.unpack_blob(TurboAEConfig(), np.load(os.path.join(GOLD, "trained_enc2dec5_u100_fp32.npz"))["weights_fp32"])
    return Channel_AE_HIP(cfg, sd, device=dev, max_batch=B)


def _resolve(model, path):
    obj = model
    for part in path.split(".")[1:]:
        obj = getattr(obj, part)         # AttributeError = the drop-in lacks something the reference's loop touches
    return obj


@pytest.mark.parametrize("run", sorted(FIX["runs"]))
def test_channel_ae_hip_serves_the_recorded_reference_call_sequence(gpu_device, run):
    rec = FIX["runs"][run]
    T = np.load(os.path.join(GOLD, "caller_trainer_test.npz"))
    a = rec["args"]
    B, L = a["batch_size"], a["block_len"]
    model = _model(gpu_device, B, a)
    precomp = bool(a["precompute_norm_stats"])
    ber_calls = []          # (u, x_dec) of every full forward, in call order
    n_enc = 0
    for ev in rec["events"]:
        if ev["op"] == "getattr":
            _resolve(model, ev["path"])
            continue
        assert ev["op"] == "call", ev
        fn = _resolve(model, ev["path"])
        if ev["path"] == "model.eval":
            assert fn() is not None
            continue
        key = f"{run}_{ev['tensors']}"
        u = np.unpackbits(T[key + "_u"])[: B * L].reshape(B, L, 1).astype(np.float32)
        assert ev["args"][0]["tensor"] == [B, L, 1]
        if ev["path"] == "model":
            noise = T[key + "_noise"]
            assert list(noise.shape) == ev["args"][1]["tensor"]           # (B, L, 3), or (B, L, 1) in the punctured pass
            xd, codes = fn(torch.from_numpy(u).to(gpu_device), torch.from_numpy(noise).to(gpu_device))
            assert [list(xd.shape), list(codes.shape)] == [r["tensor"] for r in ev["returns"]]
            assert np.abs(codes.cpu().numpy() - T[key + "_codes"]).max() <= 1e-5
            assert np.abs(xd.cpu().numpy() - T[key + "_x_dec"]).max() <= 2e-5
            ber_calls.append((u, xd.cpu().numpy(), T[key + "_x_dec"], list(noise.shape)))
        else:
            assert ev["path"] == "model.enc"
            codes = fn(torch.from_numpy(u).to(gpu_device))
            assert list(codes.shape) == ev["returns"]["tensor"]
            assert np.abs(codes.cpu().numpy() - T[key + "_codes"]).max() <= 1e-5
            if not precomp:
                assert abs(float(codes.std()) - 1.0) <= 1e-5              # 'encoder power is tensor(1.)'
            n_enc += 1
    nb = a["num_block"] // B
    assert n_enc == (2 * nb if precomp else nb)                           # trainer.py:238-246 (+ the pre-pass, trainer.py:145-153)
    # the numbers the reference printed, recomputed from the REPLAYED outputs with the reference's own arithmetic:
    # BER = mean over the first num_test_batch forwards of each SNR point of mean(round(x_hat) != round(x)) (trainer.py:176-217)
    per_snr = len(ber_calls) // a["snr_points"]
    assert per_snr == (2 * nb if run == "pos_ber" else nb + 1)            # the punctured pass / + the accidental forward
    printed = [l for l in rec["transcript"] if l.startswith("Test SNR")]
    assert len(printed) == a["snr_points"]
    for si, line in enumerate(printed):
        vals = [float(x) for x in NUM.findall(line)]
        ber_ref, bler_ref = vals[1], vals[2]
        calls = ber_calls[si * per_snr: si * per_snr + nb]
        assert all(c[3] == [B, L, 3] for c in calls)
        ber = np.mean([((xd > 0.5) != (u > 0.5)).mean() for u, xd, _, _ in calls])
        bler = np.mean([((xd > 0.5) != (u > 0.5)).any(axis=(1, 2)).mean() for u, xd, _, _ in calls])
        flips = sum(int(((xd > 0.5) != (xr > 0.5)).sum()) for _, xd, xr, _ in calls)
        assert flips <= 1
        assert abs(ber - ber_ref) <= flips / (B * L * nb) + 1e-7, (ber, ber_ref)
        assert abs(bler - bler_ref) <= flips / (B * nb) + 1e-7
    # the forward(s) after the counted ones use the (B, L, 1) noise tensor (trainer.py:198-201)
    assert all(c[3] == [B, L, 1] for c in ber_calls[nb:per_snr])


@pytest.mark.parametrize("run", sorted(FIX["runs"]))
def test_restated_eval_loop_prints_the_reference_transcript_shape(gpu_device, run):
    """evaluate.test (trainer.test restated) on the same sweep geometry: the same lines in the same order - numbers differ
    (device Philox inputs instead of the reference's unseeded host draws), so they are masked."""
    from turboae_amd import evaluate
    rec = FIX["runs"][run]
    a = rec["args"]
    model = _model(gpu_device, a["batch_size"], a)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        res = evaluate.test_from_args(model, type("Args", (), a)(), seed=5)
    got, want = _shape(buf.getvalue().splitlines()), _shape(rec["transcript"])
    assert got == want, f"\n{got}\n!=\n{want}"
    # and the numbers are the same physics: BER of the restated sweep within counting noise of the reference's
    printed = [[float(x) for x in NUM.findall(l)] for l in rec["transcript"] if l.startswith("Test SNR")]
    for si, vals in enumerate(printed):
        n = a["num_block"] * a["block_len"]
        sd = (vals[1] * 20.0 / n) ** 0.5                                  # error events cluster (~20 bit errors per bad block)
        assert abs(res["ber"][si] - vals[1]) <= 5.0 * 2 ** 0.5 * sd + 1e-3
