"""bench.py's line ends with the flat scalars a reader of the driver's record needs (the driver keeps the last 2 000 characters of
stdout + stderr): a canned full-precision result must come out with every TAIL_KEYS entry inside the last TAIL_BUDGET bytes."""
import json

import bench


def _canned():
    out = {"metric": "decoded info bits/sec @ block_len=100, 6-iter rate-1/3 CNN; BER match", "value": 72090581.97457586, "unit": "bits/s",
           "n_gpus": 1, "steps": 20, "warmup": 5, "ms_per_step": 69.35718734748662, "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "f16x2: " + "x" * 400, "data": "synthetic", "config": {"workload": "configs[1]"},
           "roofline": {"bound": "mfma", "other_configs": [{"config": "c", "decoder_frac": 0.1234567890123}] * 8},
           "cpu_baseline": {"value": 204538.48689717762, "cores": 16, "kind": "port", "sample": "s" * 300}}
    for i, k in enumerate(bench.TAIL_KEYS):
        if k == "overrides":
            out[k] = ""
        elif "flips" in k or k.endswith("_cores"):
            out[k] = 0
        else:
            out[k] = 12345678.901234567 / (1 + i) ** 3      # full-precision floats of every magnitude the line carries
    out["some_other_flat_key"] = 0.123456789012345
    return out


def test_flat_block_closes_the_line_within_the_budget():
    out = _canned()
    line = json.dumps(bench.ordered_for_tail(out))
    back = json.loads(line)
    assert set(back) == set(out)                                   # nothing lost, nothing added
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "config", "roofline", "cpu_baseline"):
        assert back[k] == out[k]                                   # contract keys and nested objects untouched
    tail = line[-bench.TAIL_BUDGET:]
    for k in bench.TAIL_KEYS:
        assert f'"{k}": ' in tail, k                               # every one readable from the tail alone
    assert list(back)[-len(bench.TAIL_KEYS):] == list(bench.TAIL_KEYS)
    # compaction keeps 5 significant digits
    assert abs(back["f32_bits_per_s"] - out["f32_bits_per_s"]) <= 1e-4 * out["f32_bits_per_s"]


def test_oversized_tail_degrades_from_the_front():
    out = _canned()
    out["overrides"] = "TAE_DEBUG_KNOBS=1 " + "TAE_SOMETHING=1 " * 60       # a long override list pushes the first keys into the body
    line = json.dumps(bench.ordered_for_tail(out))
    assert json.loads(line)["overrides"] == out["overrides"]
    assert '"overrides": ' in line[-bench.TAIL_BUDGET - 100:]
    assert '"cpu_baseline_bits_per_s": ' in line[-bench.TAIL_BUDGET - 100:]
