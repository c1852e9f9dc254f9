"""Timing of the generic RNN kernels on the GPU box: python tools/lab/quick_bench_rnn.py <cell> <B> [block_len]
(decoder = DEC_LargeRNN with -dec_rnn <cell>; 'lstm' and 'rnn' run on turboae_generic.hip, 'gru' on the MFMA kernels)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from turboae_amd import TurboAEConfig, Channel_AE_HIP, weights as W
cell = sys.argv[1]; B = int(sys.argv[2]); L = int(sys.argv[3]) if len(sys.argv) > 3 else 100
dev = torch.device("cuda", 0)
cfg = TurboAEConfig(block_len=L, decoder="TurboAE_rate3_rnn", dec_rnn=cell)
sd = W.generate_state_dict(cfg, seed=20190001, gain=1.0)
model = Channel_AE_HIP(cfg, sd, device=dev, max_batch=B)
u, noise = model.generate_inputs(B, 2.0, seed=1)
for _ in range(2): xd0, _ = model(u, noise)
torch.cuda.synchronize()
ts = []
for _ in range(3):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); xd, codes = model(u, noise); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
ms = float(np.median(ts))
import hashlib
print(f"dec_rnn={cell} L={L} B={B} NB={os.environ.get('TAE_GEN_RNN_NB', 'auto')}: forward {ms:.2f} ms  {B*L/ms/1e3:.3f} Mbit/s  "
      f"sha(x_dec)={hashlib.sha1(xd.cpu().numpy().tobytes()).hexdigest()[:12]}", flush=True)
