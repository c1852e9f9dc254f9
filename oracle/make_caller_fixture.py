"""ORACLE support - build-container only.  Records what the REFERENCE's own eval loop does to the model it is handed.

    python oracle/make_caller_fixture.py       # needs /root/reference; writes tests/golden/caller_trainer_test*.{json,npz}

SURVEY.md App. C.6 / section 8c(3): wraps a recording proxy around a real reference ``Channel_AE`` (reference-trained weights of
tests/golden/trained_enc2dec5_u100_fp32.npz), runs the reference's unmodified ``trainer.test(model, args)``
(trainer.py:135-248) on a tiny sweep (batch 50, 100 blocks, 2 SNR points) and stores

  * caller_trainer_test.json - per run: the ordered list of everything ``test`` touched on the model (attribute reads, method
    calls with argument shapes / dtypes, return shapes) and the stdout transcript, line by line;
  * caller_trainer_test.npz  - the actual tensors of every model call (bits packed, noise, x_dec, codes), so that a drop-in
    can be REPLAYED call by call on the GPU box and must print the same numbers.

Three runs: ``--precompute_norm_stats`` (pre-pass over ``model.enc``, running statistics), the default flags (the punctured second pass dies in its bare ``except`` on the first batch - after one extra
forward with a (B, L, 1) noise tensor, trainer.py:198-201 - and prints 'no pos BER specified.'), and ``--print_pos_ber
--print_pos_power`` (the punctured pass runs; its noise is (B, L, 1), broadcast over the three code symbols by
``codes + fwd_noise``, channel_ae.py:42).  Data only: no reference source is stored.
"""
from __future__ import annotations

import contextlib
import io
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import ref_harness as R            # noqa: E402
from turboae_amd import weights as W           # noqa: E402
from turboae_amd.config import TurboAEConfig   # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def _describe(v):
    if isinstance(v, torch.Tensor):
        return {"tensor": list(v.shape), "dtype": str(v.dtype).replace("torch.", "")}
    if isinstance(v, (tuple, list)):
        return [_describe(x) for x in v]
    return {"value": repr(v)}


class Recorder:
    """Transparent proxy: logs attribute reads and calls (with tensor shapes), keeps the tensors of model-level calls."""

    def __init__(self, target, log, tensors, path="model"):
        object.__setattr__(self, "_t", target)
        object.__setattr__(self, "_log", log)
        object.__setattr__(self, "_tensors", tensors)
        object.__setattr__(self, "_path", path)

    def __getattr__(self, name):
        v = getattr(self._t, name)
        self._log.append({"op": "getattr", "path": f"{self._path}.{name}",
                          "kind": "module" if isinstance(v, torch.nn.Module) else ("callable" if callable(v) else type(v).__name__)})
        if isinstance(v, torch.nn.Module) or callable(v):
            return Recorder(v, self._log, self._tensors, f"{self._path}.{name}")
        return v

    def __setattr__(self, name, value):
        self._log.append({"op": "setattr", "path": f"{self._path}.{name}"})
        setattr(self._t, name, value)

    def __call__(self, *args, **kwargs):
        idx = len(self._log)
        out = self._t(*args, **kwargs)
        rec = {"op": "call", "path": self._path, "args": [_describe(a) for a in args], "kwargs": sorted(kwargs), "returns": _describe(out)}
        if self._path in ("model", "model.enc"):
            rec["tensors"] = f"c{idx}"
            self._tensors[f"c{idx}_u"] = np.packbits(args[0].numpy().astype(np.uint8).reshape(-1))
            if self._path == "model":
                self._tensors[f"c{idx}_noise"] = args[1].numpy().copy()
                self._tensors[f"c{idx}_x_dec"] = out[0].numpy().copy()
                self._tensors[f"c{idx}_codes"] = out[1].numpy().copy()
            else:
                self._tensors[f"c{idx}_codes"] = out.numpy().copy()
        self._log.append(rec)
        if isinstance(out, torch.nn.Module):           # model.eval() returns the module: keep recording on it
            return Recorder(out, self._log, self._tensors, self._path.rsplit(".", 1)[0])
        return out


def run(name, extra_flags, tensors, cfg_over=None):
    cfg = TurboAEConfig(**(cfg_over or {}))
    sd = W.unpack_blob(TurboAEConfig(), np.load(os.path.join(GOLD, "trained_enc2dec5_u100_fp32.npz"))["weights_fp32"])
    B = 50
    model, args = R.build_reference_model(cfg.to_dict(), B)
    R.load_weights(model, sd)
    args.num_block, args.snr_test_start, args.snr_test_end, args.snr_points = 100, 1.0, 3.0, 2
    for k, v in extra_flags.items():
        setattr(args, k, v)
    from trainer import test                     # the reference's own eval loop (trainer.py:135)
    log, local = [], {}
    proxy = Recorder(model, log, local)
    torch.manual_seed(20190928)
    np.random.seed(20190928)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        test(proxy, args, use_cuda=False)
    for k, v in local.items():
        tensors[f"{name}_{k}"] = v
    read = ("batch_size", "block_len", "code_rate_k", "code_rate_n", "num_block", "snr_test_start", "snr_test_end", "snr_points",
            "precompute_norm_stats", "test_ratio", "print_pos_ber", "print_pos_power", "num_ber_puncture", "channel")
    return {"args": {k: getattr(args, k) for k in read}, "events": log, "transcript": buf.getvalue().splitlines(),
            "torch_seed": 20190928, "numpy_seed": 20190928}


def main():
    tensors = {}
    out = {"generated_by": "oracle/make_caller_fixture.py: reference trainer.test (trainer.py:135-248) on a recording proxy around the reference Channel_AE",
           "weights": "tests/golden/trained_enc2dec5_u100_fp32.npz", "torch": torch.__version__, "runs": {}}
    out["runs"]["default"] = run("default", {}, tensors)
    out["runs"]["pos_ber"] = run("pos_ber", {"print_pos_ber": True, "print_pos_power": True, "num_ber_puncture": 5}, tensors)
    # --precompute_norm_stats: the pre-pass over model.enc and the read of model.enc.mean_scalar / std_scalar (trainer.py:145-153)
    out["runs"]["precomp"] = run("precomp", {}, tensors, cfg_over={"precompute_norm_stats": True})
    with open(os.path.join(GOLD, "caller_trainer_test.json"), "w") as fh:
        json.dump(out, fh, indent=1, sort_keys=True)
    np.savez_compressed(os.path.join(GOLD, "caller_trainer_test.npz"), **tensors)
    for n, r in out["runs"].items():
        calls = [e for e in r["events"] if e["op"] == "call"]
        print(n, "events", len(r["events"]), "calls", [(c["path"], c["args"][0].get("tensor") if c["args"] else None) for c in calls])
        print("\n".join(r["transcript"][:40]))


if __name__ == "__main__":
    main()
