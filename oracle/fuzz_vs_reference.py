"""ORACLE support - build-container only.  Randomised pin of oracle/turboae_oracle.py against the REAL reference.

    python oracle/fuzz_vs_reference.py [n_cnn] [n_variants] [seed]     # needs /root/reference; writes tests/golden/oracle_fuzz_vs_reference.json

The golden vectors (oracle/make_golden.py) pin the oracle on hand-picked configurations.  This script draws the SAME kind of random
configurations that tests/test_gpu_fuzz.py later runs through the HIP kernels (widths 1..100 chosen independently for encoder and
decoder, 1-5 layers, kernel sizes 1 / 3 / 5 / 7 / 9, block lengths 1..420 clustered at the tile / workgroup / long-block edges,
num_iter_ft 1..6, 1-3 iterations, extrinsic on / off, every enc_act; GRU decoder, GRU encoder and dense stacks with every dec_act;
and 40 random combinations of the encoder-output / channel options: block_norm_ste levels, truncation, --no_code_norm, every channel
branch of channel_ae.py:40-69, --rec_quantize),
builds the reference's own Channel_AE for each (oracle/ref_harness.py: the reference's argument parser, module classes and
forward), loads the same generated weights with strict=True and compares the reference's forward with the oracle's on the same
Philox inputs.  Only the summary travels (configuration, deviations and a three-number digest of the reference's outputs per case: data, no
reference source); tests/test_oracle_golden.py checks the recorded deviations and RE-RUNS the oracle on the small cases against the
reference digests.  The chain the GPU fuzz relies on - HIP == oracle on random shapes, oracle == reference on random shapes - is closed here.
"""
from __future__ import annotations

import importlib.util
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import ref_harness as R            # noqa: E402
from oracle import turboae_oracle as O         # noqa: E402
from turboae_amd import philox, weights as W   # noqa: E402
from turboae_amd.config import TurboAEConfig   # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def _fuzz_module():
    """the case generators shared with the GPU fuzz test (tests/_fuzz_cases.py: a helper, not a test file), so both fuzzers
    walk the same configuration space"""
    spec = importlib.util.spec_from_file_location("_tae_fuzz_cases", os.path.join(ROOT, "tests", "_fuzz_cases.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def digest(a: np.ndarray) -> dict:
    """Three float64 numbers that pin a tensor without storing it: sum, projection on a fixed +-1 Philox pattern, max |.|."""
    a = np.asarray(a, dtype=np.float64).reshape(-1)
    sign = 1.0 - 2.0 * philox.random_bits(424242, 0, a.size).astype(np.float64)
    return {"n": int(a.size), "sum": float(a.sum()), "proj": float((a * sign).sum()), "max_abs": float(np.abs(a).max())}


def run(case):
    case = dict(case)
    B, wseed = case.pop("B"), case.pop("wseed")
    case.pop("fixed_nb", None)
    case.pop("kind", None)
    cfg = TurboAEConfig(**case)
    L = cfg.block_len
    sd = W.generate_state_dict(cfg, seed=wseed, gain=1.0)
    u = philox.random_bits(wseed, 0, B * L).reshape(B, L, 1)
    noise = (np.float32(O.snr_db2sigma(1.0)) * philox.random_normal(wseed, 0, B * L * 3)).reshape(B, L, 3).astype(np.float32)
    model, _ = R.build_reference_model(cfg.to_dict(), B)
    R.load_weights(model, sd)
    x_ref, c_ref = R.reference_forward(model, u, noise)
    taps = {}
    x_or, c_or = O.channel_ae_forward(torch.from_numpy(u), torch.from_numpy(noise), O.to_torch(sd), cfg.to_dict(), taps)
    x_or, c_or = x_or.numpy(), c_or.numpy()
    rec = {"config": {k: v for k, v in case.items()}, "B": B, "weight_seed": wseed}
    if not (np.isfinite(x_ref).all() and np.isfinite(c_ref).all()):
        # constant encoder output: the reference divides by std = 0 as well - both must agree on that, too
        rec["degenerate"] = True
        rec["same_nonfinite_pattern"] = bool((np.isfinite(x_ref) == np.isfinite(x_or)).all() and (np.isfinite(c_ref) == np.isfinite(c_or)).all())
        return rec
    std = float(taps["std"])
    rec.update(degenerate=False, enc_std=std, amplify=max(1.0, 0.25 / std),
               max_abs_codes=float(np.abs(c_ref - c_or).max()), max_abs_x_dec=float(np.abs(x_ref - x_or).max()),
               decision_flips=int(((x_ref > 0.5) != (x_or > 0.5)).sum()), bits=int(B * L),
               reference_digest={"x_dec": digest(x_ref), "codes": digest(c_ref)})
    return rec


def run_channel(fz, case):
    """A random combination of encoder-output / channel options: the quantisers make the forward discontinuous, so a code symbol
    within fp32 noise of a threshold may take the neighbouring level in one of the two implementations; such symbols are counted,
    and x_dec is compared on the blocks whose codes agree."""
    cfg, sd, B, u, noise, fading = fz.channel_case_inputs(case)
    model, _ = R.build_reference_model(cfg.to_dict(), B)
    R.load_weights(model, sd)
    ft = None
    if cfg.channel == "fading":      # the reference draws the coefficients itself (channel_ae.py:51-56): reproduce its two randn draws
        torch.manual_seed(case["wseed"])
        a, b = torch.randn(noise.shape), torch.randn(noise.shape)
        ft = (torch.sqrt(a ** 2 + b ** 2) / torch.sqrt(torch.tensor(3.14 / 2.0))).type(torch.FloatTensor)
        torch.manual_seed(case["wseed"])     # the reference's forward now makes the same two draws
    x_ref, c_ref = R.reference_forward(model, u, noise)
    taps = {}
    x_or, c_or = O.channel_ae_forward(torch.from_numpy(u), torch.from_numpy(noise), O.to_torch(sd), cfg.to_dict(), taps, {}, ft)
    x_or, c_or = x_or.numpy(), c_or.numpy()
    rec = {"config": {k: v for k, v in case.items() if k not in ("B", "wseed")}, "B": B, "weight_seed": case["wseed"]}
    if not (np.isfinite(x_ref).all() and np.isfinite(c_ref).all()):
        rec["degenerate"] = True
        rec["same_nonfinite_pattern"] = bool((np.isfinite(x_ref) == np.isfinite(x_or)).all() and (np.isfinite(c_ref) == np.isfinite(c_or)).all())
        return rec
    amplify = 1.0 if cfg.no_code_norm else max(1.0, 0.25 / float(taps["std"]))
    bad_c = np.abs(c_ref - c_or) > 3e-6 * amplify
    blocks_c = bad_c.reshape(B, -1).any(axis=1)
    bad_x = (np.abs(x_ref - x_or) > 5e-6 * amplify).reshape(B, -1).any(axis=1)
    rec.update(degenerate=False, amplify=amplify, code_symbols=int(bad_c.size), code_symbols_off=int(bad_c.sum()),
               blocks_with_code_mismatch=int(blocks_c.sum()), blocks_with_x_dec_mismatch=int(bad_x.sum()),
               x_dec_mismatch_outside_those_blocks=int((bad_x & ~blocks_c).sum()),
               max_abs_x_dec_on_agreeing_blocks=float(np.abs(x_ref - x_or).reshape(B, -1)[~blocks_c].max()) if (~blocks_c).any() else 0.0)
    return rec


def main():
    n_cnn = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    n_var = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    seed = int(sys.argv[3]) if len(sys.argv) > 3 else 20260928
    torch.set_num_threads(8)
    fz = _fuzz_module()
    cases = fz.draw_cases(n_cnn, seed) + fz.draw_variant_cases(n_var, seed + 1)
    out = {"generated_by": "oracle/fuzz_vs_reference.py: reference Channel_AE.forward (channel_ae.py:20-73) vs oracle/turboae_oracle.py on random configurations",
           "seed": seed, "torch": torch.__version__, "threads": torch.get_num_threads(), "cases": []}
    worst = {"codes": 0.0, "x_dec": 0.0}
    for i, c in enumerate(cases):
        rec = run(c)
        out["cases"].append(rec)
        if not rec["degenerate"]:
            worst["codes"] = max(worst["codes"], rec["max_abs_codes"] / rec["amplify"])
            worst["x_dec"] = max(worst["x_dec"], rec["max_abs_x_dec"] / rec["amplify"])
            print(f"{i:3d} codes {rec['max_abs_codes']:.2e} x_dec {rec['max_abs_x_dec']:.2e} flips {rec['decision_flips']} amplify {rec['amplify']:.1f} {rec['config']}")
        else:
            print(f"{i:3d} degenerate (std = 0), same non-finite pattern: {rec['same_nonfinite_pattern']} {rec['config']}")
    out["worst_over_amplify"] = worst
    out["channel_cases"] = []
    for i, c in enumerate(fz.draw_channel_cases(40, seed + 2)):
        rec = run_channel(fz, c)
        out["channel_cases"].append(rec)
        print(f"ch {i:2d}", {k: v for k, v in rec.items() if k not in ("config",)}, rec["config"]["channel"], rec["config"]["train_channel_mode"],
              "recq" if rec["config"]["rec_quantize"] else "")
    # r03: the configuration space of the generic fp32 kernels (LSTM / RNN cells, ENC_interRNN depths, RNN encoder + dense CNN decoder,
    # wide / many-tap / many-feature stacks) - the oracle's newest code paths against the real reference
    out["generic_cases"] = []
    for i, c in enumerate(fz.draw_generic_cases(28, seed + 3)):
        rec = run(c)
        out["generic_cases"].append(rec)
        if not rec["degenerate"]:
            worst["codes"] = max(worst["codes"], rec["max_abs_codes"] / rec["amplify"])
            worst["x_dec"] = max(worst["x_dec"], rec["max_abs_x_dec"] / rec["amplify"])
            print(f"gen {i:2d} codes {rec['max_abs_codes']:.2e} x_dec {rec['max_abs_x_dec']:.2e} flips {rec['decision_flips']} amplify {rec['amplify']:.1f} {rec['config']}")
        else:
            print(f"gen {i:2d} degenerate, same non-finite pattern: {rec['same_nonfinite_pattern']}")
    out["worst_over_amplify_incl_generic"] = worst
    with open(os.path.join(GOLD, "oracle_fuzz_vs_reference.json"), "w") as fh:
        json.dump(out, fh, indent=1, sort_keys=True)
    print("worst (divided by the 1/std amplification):", worst)


if __name__ == "__main__":
    main()
