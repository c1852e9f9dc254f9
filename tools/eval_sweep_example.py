"""The reference's eval protocol (trainer.test: 12 SNR points, batches of 500 blocks) on the short-trained fixture model:
    python tools/eval_sweep_example.py [blocks_per_snr] [decode_group|0] [hip_graph 0|1]
"""
import sys, os, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from turboae_amd import TurboAEConfig, Channel_AE_HIP, evaluate, weights as W
GOLD = os.path.join(ROOT, "tests", "golden")
NB = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
DG = (int(sys.argv[2]) or None) if len(sys.argv) > 2 else None
HG = len(sys.argv) > 3 and sys.argv[3] == "1"
M = json.load(open(os.path.join(GOLD, "MANIFEST.json")))
g = np.load(os.path.join(GOLD, "trained_enc2dec5_u100.npz"))
cfg = TurboAEConfig(**M["trained"]["config"])
sd = W.unpack_blob(cfg, g["weights_fp16"].astype(np.float32))
model = Channel_AE_HIP(cfg, sd, device=torch.device("cuda", 0), max_batch=500)
t0 = time.time()
res = evaluate.test(model, snr_test_start=-1.5, snr_test_end=4.0, snr_points=12, num_block=NB, batch_size=500, seed=1, verbose=False, decode_group=DG, hip_graph=HG)
torch.cuda.synchronize()
dt = time.time() - t0
print("seconds", round(dt, 2), "blocks", 12 * NB, "info bits/s", round(12 * NB * 100 / dt / 1e6, 2), "M")
print("BER ", ["%.3e" % b for b in res["ber"]])
print("BLER", ["%.3e" % b for b in res["bler"]])
