import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from turboae_amd import TurboAEConfig, Channel_AE_HIP, weights as W
dev = torch.device("cuda", 0)
for B, it in ((4096, 1),):
    cfg = TurboAEConfig(decoder="TurboAE_rate3_rnn", num_iteration=it)
    sd = W.generate_state_dict(cfg, seed=14, gain=1.0)
    model = Channel_AE_HIP(cfg, sd, device=dev, max_batch=B)
    u, noise = model.generate_inputs(B, 2.0, seed=78)
    rx = model.enc(u) + noise
    a = model.dec(rx); b = model.dec(rx)
    d = (a - b).abs().squeeze(2)
    nz = (d > 0)
    print(B, it, "max", float(d.max()), "count", int(nz.sum()), "blocks", int(nz.any(dim=1).sum()))
    if nz.any():
        bl = nz.any(dim=1).nonzero().flatten()
        print("  first blocks", bl[:24].tolist())
        print("  block%16 hist", torch.bincount(bl % 16, minlength=16).tolist())
        print("  (block//16)%8 hist", torch.bincount((bl // 16) % 8, minlength=8).tolist())
        print("  t hist (first 20 t with diffs)", nz.any(dim=0).nonzero().flatten()[:20].tolist(), int(nz.any(dim=0).sum()))
