"""Tolerances of the recurrent-decoder parity tests in one place, and an optional log of the deviations actually measured.

ATOL_XDEC_RNN is set from measurement (VERDICT r05 "weak" 1a): with TAE_DEVIATION_LOG=<path> every recurrent parity assertion appends
{"tag", "max_abs"} to that file; `profiles/r06_rnn_deviations.txt` is the sorted log of a full `-m gpu` run, and the constant below is
about twice its worst entry."""
import json
import os

ATOL_XDEC_RNN = 5e-6      # worst of 46 x_dec entries in profiles/r06_rnn_deviations.txt: 1.55e-6 (trained LSTM, 2 dB, f32 kernels)


def note(tag: str, d: float) -> float:
    path = os.environ.get("TAE_DEVIATION_LOG")
    if path:
        with open(path, "a") as fh:
            fh.write(json.dumps({"tag": tag, "max_abs": float(d)}) + "\n")
    return float(d)
