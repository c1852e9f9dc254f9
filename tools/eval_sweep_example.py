"""The reference's eval protocol (trainer.test restated: 12 SNR points -1.5 .. 4 dB) on the reference-trained fixture network:
    python tools/eval_sweep_example.py [blocks_per_snr] [batch_size] [decode_group|0] [hip_graph 0|1]
Defaults: 100 000 blocks per point in batches of 500 (the reference's own -num_block / -batch_size scale);
`50000 50000` is BASELINE configs[1] as quoted (one 50 000-block batch per point)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np                                                                    # noqa: E402
import torch                                                                          # noqa: E402
from turboae_amd import TurboAEConfig, Channel_AE_HIP, evaluate, weights as W        # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
NB = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
BS = int(sys.argv[2]) if len(sys.argv) > 2 else 500
DG = (int(sys.argv[3]) or None) if len(sys.argv) > 3 else None
HG = len(sys.argv) > 4 and sys.argv[4] == "1"
cfg = TurboAEConfig()
sd = W.unpack_blob(cfg, np.load(os.path.join(GOLD, "trained_enc2dec5_u100_fp32.npz"))["weights_fp32"])
model = Channel_AE_HIP(cfg, sd, device=torch.device("cuda", 0), max_batch=BS)
sweep = dict(snr_test_start=-1.5, snr_test_end=4.0, snr_points=12, num_block=NB, batch_size=BS, seed=1, verbose=False, decode_group=DG, hip_graph=HG)
evaluate.test(model, **{**sweep, "snr_points": 1, "snr_test_end": -1.5})             # warm-up: workspace growth, first launches
torch.cuda.synchronize()
t0 = time.perf_counter()
res = evaluate.test(model, **sweep)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
model.check_range()
blocks = 12 * (NB // BS) * BS
print(json.dumps({"seconds": round(dt, 3), "blocks": blocks, "batch_size": BS, "hip_graph": HG, "info_bits_per_s": blocks * 100 / dt,
                  "ber": res["ber"], "bler": res["bler"]}))
