"""Timing of the DenseSameShapeConv1d variant (encoder = decoder = TurboAE_rate3_cnn_dense): python tools/lab/quick_bench_dense.py [B]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from turboae_amd import TurboAEConfig, Channel_AE_HIP, weights as W
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = torch.device("cuda", 0)
cfg = TurboAEConfig(encoder="TurboAE_rate3_cnn_dense", decoder="TurboAE_rate3_cnn_dense")
sd = W.generate_state_dict(cfg, seed=20190001, gain=0.5)
model = Channel_AE_HIP(cfg, sd, device=dev, max_batch=B)
u, noise = model.generate_inputs(B, 2.0, seed=1)
for _ in range(2): model(u, noise)
torch.cuda.synchronize()
ts = []
for _ in range(3):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); xd, codes = model(u, noise); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
ms = float(np.median(ts))
fl = cfg.flops_per_bit() * B * cfg.block_len
print(f"dense L={cfg.block_len} B={B}: forward {ms:.2f} ms  {B*cfg.block_len/ms/1e3:.2f} Mbit/s  {fl/ms/1e9:.1f} TFLOP/s algorithmic  range={model.range_status()} kernel_info={model.kernel_info()}", flush=True)
