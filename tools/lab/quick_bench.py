"""Quick decoder/forward timing on the GPU box (not the contract bench): python tools/lab/quick_bench.py [B]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from turboae_amd import TurboAEConfig, Channel_AE_HIP, weights as W
B = int(sys.argv[1]) if len(sys.argv) > 1 else 12288
dev = torch.device("cuda", 0)
cfg = TurboAEConfig()
sd = W.generate_state_dict(cfg, seed=20190001, gain=1.0)
model = Channel_AE_HIP(cfg, sd, device=dev, max_batch=B)
u, noise = model.generate_inputs(B, 2.0, seed=1)
rx = model.enc(u) + noise
for _ in range(2): model.dec(rx)
torch.cuda.synchronize()
ts = []
for _ in range(5):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); xd = model.dec(rx); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
ms = float(np.median(ts))
fl = 2.0 * cfg.macs_per_bit()["dec"] * B * cfg.block_len
print(f"B={B} dec_kernel {ms:.3f} ms  {fl/ms/1e9:.1f} TFLOP/s  frac={fl/ms/1e9/157.3:.3f}  ({B*100/ms/1e3:.2f} Mbit/s decoder-only)", flush=True)
ts = []
for _ in range(3):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); xd, codes = model(u, noise); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
ms = float(np.median(ts))
print(f"B={B} forward {ms:.3f} ms  {B*100/ms/1e3:.2f} Mbit/s", flush=True)
ts = []
for _ in range(5):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); c = model.enc(u); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
print(f"B={B} encoder + power constraint {float(np.median(ts)):.3f} ms", flush=True)
