import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from turboae_amd import TurboAEConfig, Channel_AE_HIP, weights as W
from oracle import turboae_oracle as O
dev = torch.device("cuda", 0)
for B in (40, 16400):
    cfg = TurboAEConfig(decoder="TurboAE_rate3_rnn", num_iteration=2 if B < 100 else 6)
    sd = W.generate_state_dict(cfg, seed=14, gain=1.0)
    model = Channel_AE_HIP(cfg, sd, device=dev, max_batch=B)
    u, noise = model.generate_inputs(B, 2.0, seed=78)
    x_dec, codes = model(u, noise)
    rx = codes + noise
    x2 = model.dec(rx)
    print(B, "fwd vs dec(full):", float((x2 - x_dec).abs().max()))
    for lo, hi in ((0, 5), (3, 19), (16, 32), (B - 7, B)):
        xs = model.dec(rx[lo:hi].contiguous())
        d = (xs - x_dec[lo:hi]).abs()
        print(B, (lo, hi), "max diff sub vs full", float(d.max()), "n diff", int((d > 0).sum()), "of", d.numel(), "blocks differing", (d.amax(dim=(1, 2)) > 0).nonzero().flatten().tolist()[:20])
    idx = [0, 1, B - 1]
    with torch.no_grad():
        xo = O.decode_rnn(rx[idx].cpu(), O.to_torch(sd), torch.from_numpy(O.rand_interleaver(cfg.block_len, 0)), cfg.dec_num_unit, cfg.num_iteration, cfg.num_iter_ft)
    print(B, "vs oracle", float((x_dec[idx].cpu() - xo).abs().max()))
    # determinism: same call twice
    x3 = model.dec(rx)
    print(B, "repeat:", float((x3 - x2).abs().max()))
