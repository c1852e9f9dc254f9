"""Timing of other BASELINE configs on the GPU box: python tools/lab/quick_bench_cfg.py <block_len> <B> [enc_layers]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from turboae_amd import TurboAEConfig, Channel_AE_HIP, weights as W
L = int(sys.argv[1]); B = int(sys.argv[2]); encl = int(sys.argv[3]) if len(sys.argv) > 3 else 2
decoder = sys.argv[4] if len(sys.argv) > 4 else "TurboAE_rate3_cnn"
dev = torch.device("cuda", 0)
cfg = TurboAEConfig(block_len=L, enc_num_layer=encl, decoder=decoder)
sd = W.generate_state_dict(cfg, seed=20190001, gain=1.0)
model = Channel_AE_HIP(cfg, sd, device=dev, max_batch=B)
u, noise = model.generate_inputs(B, 2.0, seed=1)
for _ in range(2): model(u, noise)
torch.cuda.synchronize()
ts = []
for _ in range(3):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); xd, codes = model(u, noise); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
ms = float(np.median(ts))
fl = cfg.flops_per_bit() * B * L
import hashlib
sha = hashlib.sha256(xd.cpu().numpy().tobytes()).hexdigest()[:12]
print(f"{decoder} L={L} B={B} enc{encl}: sha {sha} forward {ms:.2f} ms  {B*L/ms/1e3:.2f} Mbit/s  {fl/ms/1e9:.1f} TFLOP/s ({fl/ms/1e9/157.3:.3f} of fp32 MFMA peak)  kernel_info={model.kernel_info()}", flush=True)
