"""ORACLE support - build-container only.  Imports the REAL reference from /root/reference.

Used by ``oracle/make_golden.py`` to (1) prove ``oracle/turboae_oracle.py`` equal to the reference
and (2) emit golden vectors.  Never imported by tests on the GPU box (the reference does not
travel); every public function raises if ``/root/reference`` is missing.

Recipe: SURVEY.md Appendix C.  Two shims are applied before import because the vendored commpy
is py2 / old-numpy era (commpy/channels.py:19 ``from numpy import complex``;
commpy/channelcoding/gfields.py:8 ``from fractions import gcd``).  The reference tree is not
modified and no bytecode is written.
"""
from __future__ import annotations

import os
import sys
from typing import Dict

import numpy as np

REFERENCE_DIR = "/root/reference"


def reference_available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_DIR, "channel_ae.py"))


def _import_reference():
    if not reference_available():
        raise RuntimeError("reference tree not present; ref_harness only works in the build container")
    import math
    import fractions
    sys.dont_write_bytecode = True
    if REFERENCE_DIR not in sys.path:
        sys.path.insert(0, REFERENCE_DIR)
    if not hasattr(np, "complex"):
        np.complex = complex            # commpy/channels.py:19
    if not hasattr(fractions, "gcd"):
        fractions.gcd = math.gcd        # commpy/channelcoding/gfields.py:8, algcode.py:6


def build_reference_model(cfg: dict, batch_size: int, is_parallel: int = 1, is_interleave: int = 1):
    """Channel_AE(ENC_interCNN, DEC_LargeCNN) built exactly as main.py:109-159 does."""
    _import_reference()
    enc_name = cfg.get("encoder", "TurboAE_rate3_cnn")
    # get_args.py:10 spells the GRU-encoder choice 'Turboae_rate3_rnn' while main.py:32 tests 'TurboAE_rate3_rnn': pass the
    # spelling argparse accepts and set the one import_enc understands afterwards
    argv = ["main.py", "-encoder", "Turboae_rate3_rnn" if enc_name == "TurboAE_rate3_rnn" else enc_name,
            "-decoder", cfg.get("decoder", "TurboAE_rate3_cnn"),
            "-enc_num_unit", str(cfg["enc_num_unit"]), "-enc_num_layer", str(cfg["enc_num_layer"]),
            "-dec_num_unit", str(cfg["dec_num_unit"]), "-dec_num_layer", str(cfg["dec_num_layer"]),
            "-num_iteration", str(cfg["num_iteration"]), "-num_iter_ft", str(cfg["num_iter_ft"]),
            "-extrinsic", str(cfg.get("extrinsic", 1)), "-enc_act", cfg.get("enc_act", "elu"), "-dec_act", cfg.get("dec_act", "linear"),
            "-enc_kernel_size", str(cfg.get("enc_kernel_size", 5)), "-dec_kernel_size", str(cfg.get("dec_kernel_size", 5)),
            "-is_parallel", str(is_parallel), "-is_interleave", str(is_interleave), "-batch_size", str(batch_size),
            "-block_len", str(cfg["block_len"]), "--no-cuda",
            "-channel", cfg.get("channel", "awgn"), "-train_channel_mode", cfg.get("train_channel_mode", "block_norm"),
            "-enc_truncate_limit", str(cfg.get("enc_truncate_limit", 0.0)),
            "-enc_value_limit", str(cfg.get("enc_value_limit", 1.0)),
            "-enc_quantize_level", str(cfg.get("enc_quantize_level", 2.0)),
            "-rec_quantize_level", str(cfg.get("rec_quantize_level", 2)),
            "-enc_rnn", cfg.get("enc_rnn", "gru"), "-dec_rnn", cfg.get("dec_rnn", "gru")]
    if cfg.get("no_code_norm", False):
        argv.append("--no_code_norm")
    if cfg.get("precompute_norm_stats", False):
        argv.append("--precompute_norm_stats")
    if cfg.get("rec_quantize", False):
        argv.append("--rec_quantize")
    old = sys.argv
    sys.argv = argv
    try:
        from get_args import get_args
        args = get_args()
    finally:
        sys.argv = old
    args.encoder = enc_name
    from main import import_enc, import_dec
    from channel_ae import Channel_AE
    from numpy import arange
    from numpy.random import mtrand
    ENC, DEC = import_enc(args), import_dec(args)
    if is_interleave == 0:
        p_array = range(args.block_len)                                     # main.py:129-131: no interleaver
    else:
        p_array = mtrand.RandomState(0).permutation(arange(args.block_len)) # main.py:123-127
    model = Channel_AE(args, ENC(args, p_array), DEC(args, p_array))
    if is_parallel:
        model.enc.set_parallel()
        model.dec.set_parallel()
    model.eval()
    return model, args


def load_weights(model, state_dict: Dict[str, np.ndarray], is_parallel: int = 1) -> None:
    """strict=True load of a '.module'-free numpy state dict (keys re-wrapped when is_parallel)."""
    import re
    import torch
    sd = {}
    for k, v in state_dict.items():
        if is_parallel:
            k = re.sub(r"^(enc\.enc_cnn_\d|enc\.enc_rnn_\d|enc\.enc_linear_\d|dec\.dec\d_cnns\.\d+|dec\.dec\d_rnns\.\d+|dec\.dec\d_outputs\.\d+)\.",
                       r"\1.module.", k)
        sd[k] = torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32))
    model.load_state_dict(sd, strict=True)


def reference_forward(model, u: np.ndarray, noise: np.ndarray):
    import torch
    with torch.no_grad():
        x_dec, codes = model(torch.from_numpy(u), torch.from_numpy(noise))
    return x_dec.numpy().copy(), codes.numpy().copy()


def reference_state_dict(model) -> Dict[str, np.ndarray]:
    import re
    return {re.sub(r"\.module(?=\.|$)", "", k): v.detach().cpu().numpy().copy() for k, v in model.state_dict().items()}
