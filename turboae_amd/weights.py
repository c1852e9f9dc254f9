"""Weight naming, generation and (de)serialisation for the TurboAE CNN hot path.

Key grammar follows the reference ``state_dict`` (modules created at
encoders.py:313-335, decoders.py:179-192).  ``-is_parallel 1`` wraps every sub-module in
``DataParallel`` (encoders.py:343-349, decoders.py:194-199), which inserts ``.module`` into the
keys; :func:`strip_module` removes it so both flavours load.

The *canonical blob* handed to the C ABI (``tae_create``) is the concatenation, in
:func:`canonical_entries` order, of the tensors in their PyTorch layouts as little-endian fp32:
conv weight ``(C_out, C_in, K)`` row-major, conv bias ``(C_out,)``, linear weight ``(out, in)``,
linear bias ``(out,)``.  Re-tiling into MFMA fragment order happens inside the library.
"""
from __future__ import annotations

import re
from typing import Dict, List, Tuple

import numpy as np

from .config import TurboAEConfig
from . import philox


def canonical_entries(cfg: TurboAEConfig) -> List[Tuple[str, Tuple[int, ...]]]:
    """Ordered (key, shape) list of every tensor of Channel_AE(ENC_interCNN, DEC_LargeCNN)."""
    out: List[Tuple[str, Tuple[int, ...]]] = []
    ue, ud, f = cfg.enc_num_unit, cfg.dec_num_unit, cfg.num_iter_ft
    ke, kd = cfg.enc_kernel_size, cfg.dec_kernel_size
    gates = {"gru": 3, "lstm": 4, "rnn": 1}       # rows of weight_ih / weight_hh per hidden unit (torch.nn.GRU / LSTM / RNN)
    ge, gd = gates[cfg.enc_rnn], gates[cfg.dec_rnn]
    for s in (1, 2, 3):
        if cfg.encoder == "TurboAE_rate3_rnn":
            # ENC_interRNN: torch.nn.GRU(1, ue, num_layers, bidirectional=True) + Linear(2 ue, 1) (encoders.py:251-268)
            for l in range(cfg.enc_num_layer):
                cin = 1 if l == 0 else 2 * ue
                for sfx in ("", "_reverse"):
                    out.append((f"enc.enc_rnn_{s}.weight_ih_l{l}{sfx}", (ge * ue, cin)))
                    out.append((f"enc.enc_rnn_{s}.weight_hh_l{l}{sfx}", (ge * ue, ue)))
                    out.append((f"enc.enc_rnn_{s}.bias_ih_l{l}{sfx}", (ge * ue,)))
                    out.append((f"enc.enc_rnn_{s}.bias_hh_l{l}{sfx}", (ge * ue,)))
            out.append((f"enc.enc_linear_{s}.weight", (1, 2 * ue)))
            out.append((f"enc.enc_linear_{s}.bias", (1,)))
            continue
        for l in range(cfg.enc_num_layer):
            cin = cfg.code_rate_k if l == 0 else ue
            if cfg.dense:       # DenseSameShapeConv1d: layer l sees the inputs and every earlier layer's output (cnn_utils.py:59-62)
                cin = cfg.code_rate_k + l * ue
            out.append((f"enc.enc_cnn_{s}.cnns.{l}.weight", (ue, cin, ke)))
            out.append((f"enc.enc_cnn_{s}.cnns.{l}.bias", (ue,)))
        out.append((f"enc.enc_linear_{s}.weight", (1, ue)))
        out.append((f"enc.enc_linear_{s}.bias", (1,)))
    rnn = cfg.decoder == "TurboAE_rate3_rnn"
    for it in range(cfg.num_iteration):
        for half in (1, 2):
            nout = 1 if (half == 2 and it == cfg.num_iteration - 1) else f
            if rnn:
                # torch.nn.GRU(2+F, ud, num_layers=2, bidirectional=True) parameter names (decoders.py:41-49)
                for l in (0, 1):
                    cin = 2 + f if l == 0 else 2 * ud
                    for sfx in ("", "_reverse"):
                        out.append((f"dec.dec{half}_rnns.{it}.weight_ih_l{l}{sfx}", (gd * ud, cin)))
                        out.append((f"dec.dec{half}_rnns.{it}.weight_hh_l{l}{sfx}", (gd * ud, ud)))
                        out.append((f"dec.dec{half}_rnns.{it}.bias_ih_l{l}{sfx}", (gd * ud,)))
                        out.append((f"dec.dec{half}_rnns.{it}.bias_hh_l{l}{sfx}", (gd * ud,)))
                out.append((f"dec.dec{half}_outputs.{it}.weight", (nout, 2 * ud)))
                out.append((f"dec.dec{half}_outputs.{it}.bias", (nout,)))
            else:
                for l in range(cfg.dec_num_layer):
                    cin = 2 + f if l == 0 else ud
                    if cfg.dec_dense:       # decoders.py:173-176: dense stacks whenever the encoder is not the plain CNN
                        cin = 2 + f + l * ud
                    out.append((f"dec.dec{half}_cnns.{it}.cnns.{l}.weight", (ud, cin, kd)))
                    out.append((f"dec.dec{half}_cnns.{it}.cnns.{l}.bias", (ud,)))
                out.append((f"dec.dec{half}_outputs.{it}.weight", (nout, ud)))
                out.append((f"dec.dec{half}_outputs.{it}.bias", (nout,)))
    return out


def num_params(cfg: TurboAEConfig) -> int:
    return int(sum(int(np.prod(s)) for _, s in canonical_entries(cfg)))


_MODULE_RE = re.compile(r"\.module(?=\.|$)")


def strip_module(state_dict: Dict[str, object]) -> Dict[str, object]:
    """Drop DataParallel's ``.module`` path components (main.py:157-159)."""
    return {_MODULE_RE.sub("", k): v for k, v in state_dict.items()}


def add_module(state_dict: Dict[str, object]) -> Dict[str, object]:
    """Inverse of :func:`strip_module`: keys as the reference produces with ``-is_parallel 1``."""
    out = {}
    for k, v in state_dict.items():
        k2 = re.sub(r"^(enc\.enc_cnn_\d|enc\.enc_rnn_\d|enc\.enc_linear_\d|dec\.dec\d_cnns\.\d+|dec\.dec\d_rnns\.\d+|dec\.dec\d_outputs\.\d+)\.",
                    r"\1.module.", k)
        out[k2] = v
    return out


def _to_numpy(v) -> np.ndarray:
    if hasattr(v, "detach"):
        v = v.detach().cpu().numpy()
    return np.ascontiguousarray(np.asarray(v, dtype=np.float32))


def check_state_dict(cfg: TurboAEConfig, state_dict: Dict[str, object]) -> Dict[str, np.ndarray]:
    """Strict key/shape check (the reference loads with strict=False and silently ignores
    mismatches, main.py:166-172; we refuse instead)."""
    sd = strip_module(state_dict)
    out: Dict[str, np.ndarray] = {}
    missing, bad = [], []
    for key, shape in canonical_entries(cfg):
        if key not in sd:
            missing.append(key)
            continue
        arr = _to_numpy(sd[key])
        if tuple(arr.shape) != tuple(shape):
            bad.append(f"{key}: got {tuple(arr.shape)}, want {tuple(shape)}")
            continue
        out[key] = arr
    extra = [k for k in sd if k not in out and k not in missing]
    if missing or bad or extra:
        raise ValueError("state_dict does not match config: missing=%s bad_shape=%s unexpected=%s"
                         % (missing[:4], bad[:4], extra[:4]))
    return out


def pack_blob(cfg: TurboAEConfig, state_dict: Dict[str, object]) -> np.ndarray:
    sd = check_state_dict(cfg, state_dict)
    return np.concatenate([sd[k].reshape(-1) for k, _ in canonical_entries(cfg)]).astype("<f4")


def unpack_blob(cfg: TurboAEConfig, blob: np.ndarray) -> Dict[str, np.ndarray]:
    blob = np.asarray(blob, dtype="<f4").reshape(-1)
    if blob.size != num_params(cfg):
        raise ValueError(f"blob has {blob.size} floats, config needs {num_params(cfg)}")
    out, off = {}, 0
    for key, shape in canonical_entries(cfg):
        n = int(np.prod(shape))
        out[key] = blob[off:off + n].reshape(shape).copy()
        off += n
    return out


def generate_state_dict(cfg: TurboAEConfig, seed: int = 20190001, gain: float = 1.0) -> Dict[str, np.ndarray]:
    """Portable random weights: every tensor ~ U(-b, b), b = gain * sqrt(3 / fan_in) for weights
    (variance-preserving "kaiming-like") and b = 1/sqrt(fan_in) for biases (PyTorch's default bias
    bound).  Drawn from Philox stream STREAM_WEIGHTS so any host reproduces them bit-for-bit.
    There are no pretrained models in the reference mount (.MISSING_LARGE_BLOBS), so performance
    and tensor-parity runs use these; BER-meaningful runs use the short-trained fixture.
    """
    out: Dict[str, np.ndarray] = {}
    off = 0
    fan_in = 1
    for key, shape in canonical_entries(cfg):
        n = int(np.prod(shape))
        u = philox.random_uniform_pm1(seed, off, n, philox.STREAM_WEIGHTS)
        off += n
        if key.rsplit(".", 1)[-1].startswith("weight"):     # "...weight", "weight_ih_l0", "weight_hh_l1_reverse", ...
            fan_in = int(np.prod(shape[1:]))
            bound = gain * np.sqrt(3.0 / fan_in)
        else:
            bound = 1.0 / np.sqrt(fan_in)
        out[key] = (u * np.float32(bound)).astype(np.float32).reshape(shape)
    return out


def scale_last_layers(state_dict: Dict[str, object], cfg: TurboAEConfig, factor: float, encoder: bool = True,
                      decoder_stacks=None) -> Dict[str, np.ndarray]:
    """Copy of a CNN `state_dict` with the LAST conv layer of the chosen stacks (all three encoder stacks when `encoder`, the
    decoder stacks listed in `decoder_stacks` as (iteration, half) pairs, default all) scaled by `factor` - weight and bias - and
    the Linear layer behind it by 1 / factor.  Exactly the same function where that layer is in the linear part of its ELU, a
    slightly different network elsewhere; what it is for: a last layer whose activations stay below 1 (factor < 1) makes the
    fp16-split kernels evaluate both expm1 branches in that head (test / bench case for that instantiation)."""
    out = {k: np.array(v, dtype=np.float32, copy=True) for k, v in state_dict.items()}
    f = np.float32(factor)
    if encoder:
        for s in (1, 2, 3):
            out[f"enc.enc_cnn_{s}.cnns.{cfg.enc_num_layer - 1}.weight"] *= f
            out[f"enc.enc_cnn_{s}.cnns.{cfg.enc_num_layer - 1}.bias"] *= f
            out[f"enc.enc_linear_{s}.weight"] /= f
    if decoder_stacks is None:
        decoder_stacks = [(it, h) for it in range(cfg.num_iteration) for h in (1, 2)]
    for it, h in decoder_stacks:
        out[f"dec.dec{h}_cnns.{it}.cnns.{cfg.dec_num_layer - 1}.weight"] *= f
        out[f"dec.dec{h}_cnns.{it}.cnns.{cfg.dec_num_layer - 1}.bias"] *= f
        out[f"dec.dec{h}_outputs.{it}.weight"] /= f
    return out


def golden_state_dict(cfg: TurboAEConfig, meta: Dict[str, object]) -> Dict[str, np.ndarray]:
    """Weights of a tests/golden/MANIFEST.json case: the portable generator at (weight_seed, gain), then the case's
    `last_layer_scale` = {"factor", "encoder", "decoder_stacks"} if it has one (scale_last_layers)."""
    sd = generate_state_dict(cfg, seed=int(meta["weight_seed"]), gain=float(meta["gain"]))
    t = meta.get("last_layer_scale")
    if t:
        stacks = t.get("decoder_stacks")
        sd = scale_last_layers(sd, cfg, float(t["factor"]), bool(t.get("encoder", True)),
                               None if stacks is None else [tuple(x) for x in stacks])
    return sd


def save_blob(path: str, cfg: TurboAEConfig, state_dict: Dict[str, object]) -> None:
    """Flat little-endian fp32 blob + JSON manifest (``<path>.json``)."""
    import json
    blob = pack_blob(cfg, state_dict)
    blob.tofile(path)
    with open(path + ".json", "w") as fh:
        json.dump({"config": cfg.to_dict(), "dtype": "<f4", "count": int(blob.size),
                   "entries": [[k, list(s)] for k, s in canonical_entries(cfg)]}, fh, indent=1)


def load_blob(path: str) -> Tuple[TurboAEConfig, Dict[str, np.ndarray]]:
    import json
    with open(path + ".json") as fh:
        man = json.load(fh)
    cfg = TurboAEConfig(**man["config"])
    return cfg, unpack_blob(cfg, np.fromfile(path, dtype="<f4"))


def from_torch_checkpoint(path: str, cfg: TurboAEConfig) -> Dict[str, np.ndarray]:
    """Load an upstream ``.pt`` the way main.py:162-172 does (whole pickled model or a plain
    state_dict, with or without ``.module``), but with strict checking."""
    import torch
    obj = torch.load(path, map_location="cpu", weights_only=False)
    sd = obj.state_dict() if hasattr(obj, "state_dict") else obj
    return check_state_dict(cfg, sd)
