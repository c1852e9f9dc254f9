"""debug: one LSTM stack pair on the fused layer 1 of a -DTAE_L1F_DBG_GI build with TAE_RNN_L1=check (prints accumulator mismatches)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", ".."))
import numpy as np, torch
from turboae_amd import TurboAEConfig, philox, weights as W, Channel_AE_HIP
from oracle import turboae_oracle as O
B, L = int(sys.argv[1]), int(sys.argv[2])
cfg = TurboAEConfig(decoder="TurboAE_rate3_rnn", dec_rnn="lstm", block_len=L, num_iteration=1)
sd = W.generate_state_dict(cfg, seed=900 + L + B, gain=1.0)
u = philox.random_bits(19, 0, B * L).reshape(B, L, 1)
noise = (np.float32(O.snr_db2sigma(1.0)) * philox.random_normal(19, 0, B * L * 3)).reshape(B, L, 3).astype(np.float32)
dev = torch.device("cuda", 0)
m = Channel_AE_HIP(cfg, sd, device=dev, max_batch=B)
print(m.overrides())
xd, codes = m(torch.from_numpy(u).to(dev), torch.from_numpy(noise).to(dev))
torch.cuda.synchronize()
print("done")
