"""Timing of any configuration on the GPU box: python tools/lab/quick_bench_any.py B key=value ...   (TurboAEConfig fields; ints parsed)
e.g. python tools/lab/quick_bench_any.py 16384 decoder=TurboAE_rate3_rnn dec_rnn=lstm ; python tools/lab/quick_bench_any.py 2048 dec_num_unit=136 enc_num_unit=136"""
import sys, os, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from turboae_amd import TurboAEConfig, Channel_AE_HIP, weights as W
B = int(sys.argv[1])
kw = {}
for a in sys.argv[2:]:
    k, v = a.split("=", 1)
    try: v = int(v)
    except ValueError: pass
    kw[k] = v
dev = torch.device("cuda", 0)
cfg = TurboAEConfig(**kw)
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
L = cfg.block_len
print(f"{' '.join(sys.argv[2:])} B={B} conv={os.environ.get('TAE_GEN_CONV', 'mfma')} nb={os.environ.get('TAE_GEN_RNN_NB', 'auto')}: forward {ms:.2f} ms  {B*L/ms/1e3:.3f} Mbit/s  "
      f"generic={cfg.generic}  sha(x_dec)={hashlib.sha1(xd.cpu().numpy().tobytes()).hexdigest()[:12]}", flush=True)
