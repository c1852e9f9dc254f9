"""ORACLE support - build-container only.  Writes warm-start checkpoints for oracle/train_fixture.py
(`-init_nw_weight`, main.py:162-172, loaded with strict=False) from the reference-trained enc2/dec5 fixture
(tests/golden/trained_enc2dec5_u100_fp32.npz), so that the REFERENCE TRAINER reaches an operating point for
BASELINE configs[2] (enc5/dec5) and configs[4] (CNN encoder + GRU decoder) in CPU-hours instead of GPU-days:

    python oracle/make_init_checkpoint.py enc5 /tmp/train_enc5/init.pt
        encoder layers 0, 1 and the Linear heads = the trained enc2 encoder; the three extra layers start as
        identity centre taps + N(0, 0.02) (so the encoder starts NEAR the trained code, not at it: ELU is applied
        three more times); decoder = the trained dec5.  Every tensor is then trained further by the reference.
    python oracle/make_init_checkpoint.py encoder_only /tmp/train_gru/init.pt
        only the enc.* tensors (the GRU decoder's keys / Linear shapes differ: main.py's strict=False load would
        raise on the (5, 200) vs (5, 100) heads); the GRU decoder starts from torch's default init.

Nothing here is shipped or imported by tests; the resulting fixtures record this provenance in MANIFEST.json.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from turboae_amd.config import TurboAEConfig           # noqa: E402
from turboae_amd import weights as W                   # noqa: E402


def main():
    kind, out = sys.argv[1], sys.argv[2]
    cfg = TurboAEConfig()
    z = np.load(os.path.join(HERE, "..", "tests", "golden", "trained_enc2dec5_u100_fp32.npz"))
    sd = W.unpack_blob(cfg, z["weights_fp32"])
    rng = np.random.RandomState(20190004)
    init = {}
    if kind == "enc5":
        init.update(sd)
        U = cfg.enc_num_unit
        for s in (1, 2, 3):
            for l in (2, 3, 4):
                w = (0.02 * rng.standard_normal((U, U, 5))).astype(np.float32)
                w[np.arange(U), np.arange(U), 2] += 1.0
                init[f"enc.enc_cnn_{s}.cnns.{l}.weight"] = w
                init[f"enc.enc_cnn_{s}.cnns.{l}.bias"] = np.zeros(U, np.float32)
    elif kind == "encoder_only":
        init.update({k: v for k, v in sd.items() if k.startswith("enc.")})
    else:
        raise SystemExit("kind: enc5 | encoder_only")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    torch.save({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in init.items()}, out)
    print(f"{kind}: {len(init)} tensors -> {out}")


if __name__ == "__main__":
    main()
