"""Times the 12-point sweep of BASELINE configs[1] (bench.py::sweep_cfg1's workload) eagerly and with one hipGraph per SNR point:
    python tools/lab/sweep_time.py [points = 12] [blocks = 50000]"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench
from turboae_amd import evaluate, weights as W
from turboae_amd.channel_ae import Channel_AE_HIP
from turboae_amd.config import TurboAEConfig

points = int(sys.argv[1]) if len(sys.argv) > 1 else 12
blocks = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
dev = torch.device("cuda", 0)
cfg = TurboAEConfig()
sd = W.unpack_blob(cfg, np.load(bench.TRAINED)["weights_fp32"])
model = Channel_AE_HIP(cfg, sd, device=dev, max_batch=blocks)
res = {}
for name, kw in (("eager", dict(hip_graph=False)), ("graph", dict(hip_graph=True))):
    args = dict(snr_test_start=-1.5, snr_test_end=4.0, snr_points=points, num_block=blocks, batch_size=blocks, seed=bench.SEED,
                verbose=False, enc_power_epilogue=False, **kw)
    evaluate.test(model, **args)
    ts = []
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res[name] = evaluate.test(model, **args)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    print(f"{name}: {min(ts):.4f} s (runs {', '.join(f'{t:.4f}' for t in ts)})  {points * blocks * cfg.block_len / min(ts) / 1e6:.2f} M bits/s", flush=True)
print("counts equal:", res["eager"]["bit_errors"] == res["graph"]["bit_errors"] and res["eager"]["block_errors"] == res["graph"]["block_errors"])
