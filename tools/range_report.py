"""Per-layer activation exponents the range calibration picked (GPU box): python tools/range_report.py [trained|random] [gain]
Exponent A of a panel: its calibration maximum M lies in [2^(10 - A), 2^(11 - A))."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from turboae_amd import TurboAEConfig, Channel_AE_HIP, weights as W
kind = sys.argv[1] if len(sys.argv) > 1 else "trained"
cfg = TurboAEConfig()
if kind == "trained":
    z = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "trained_enc2dec5_u100_fp32.npz"))
    sd = W.unpack_blob(cfg, z["weights_fp32"])
else:
    sd = W.generate_state_dict(cfg, seed=7, gain=float(sys.argv[2]) if len(sys.argv) > 2 else 1.0)
m = Channel_AE_HIP(cfg, sd, device=torch.device("cuda", 0), max_batch=500)
enc, dec, passes = m._eng.range_info()
print(kind, "passes", passes)
print("encoder: stack inputs", enc[:3], "panels", np.array(enc[3:]).reshape(3, cfg.enc_num_layer).tolist())
ns = 2 * cfg.num_iteration
print("decoder: stack inputs", dec[:ns])
print("decoder panels (rows = stacks):")
print(np.array(dec[ns:]).reshape(ns, cfg.dec_num_layer))
print("layer maxima lie in [2^(10-A), 2^(11-A))")
