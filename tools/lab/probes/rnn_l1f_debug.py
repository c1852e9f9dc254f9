"""fused vs split layer 1 of the LSTM / RNN decoder stacks: where do they differ?  python tools/lab/probes/rnn_l1f_debug.py"""
import os, sys, subprocess, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", ".."))
import numpy as np, torch
from turboae_amd import TurboAEConfig, philox, weights as W, Channel_AE_HIP
from oracle import turboae_oracle as O
mode = "split" if os.environ.get("TAE_RNN_L1") == "split" else "fused"
dev = torch.device("cuda", 0)
out = {}
for cell, B, L, U, it in (("lstm", 40, 2, 100, 1), ("lstm", 64, 4, 100, 1), ("rnn", 33, 6, 100, 1), ("lstm", 16, 2, 100, 1), ("lstm", 16, 4, 100, 1), ("lstm", 1, 100, 100, 1), ("lstm", 37, 100, 100, 2), ("rnn", 70, 33, 100, 2), ("lstm", 5, 7, 100, 2), ("lstm", 2049, 11, 100, 1)):
    cfg = TurboAEConfig(decoder="TurboAE_rate3_rnn", dec_rnn=cell, block_len=L, dec_num_unit=U, num_iteration=it)
    sd = W.generate_state_dict(cfg, seed=900 + L + B, gain=1.0)
    u = philox.random_bits(19, 0, B * L).reshape(B, L, 1)
    noise = (np.float32(O.snr_db2sigma(1.0)) * philox.random_normal(19, 0, B * L * 3)).reshape(B, L, 3).astype(np.float32)
    m = Channel_AE_HIP(cfg, sd, device=dev, max_batch=B)
    xd, codes = m(torch.from_numpy(u).to(dev), torch.from_numpy(noise).to(dev))
    xd2, _ = m(torch.from_numpy(u).to(dev), torch.from_numpy(noise).to(dev))
    x = xd.cpu().numpy()
    np.save(f"/tmp/l1f_{mode}_{cell}_{B}_{L}.npy", x)
    print(mode, cell, B, L, "nan", int(np.isnan(x).sum()), "rerun equal", bool(torch.equal(xd, xd2)), flush=True)
    other = f"/tmp/l1f_{'split' if mode == 'fused' else 'fused'}_{cell}_{B}_{L}.npy"
    if os.path.isfile(other):
        y = np.load(other)
        d = np.abs(x - y)
        bad = np.argwhere(~(d <= 0))
        print("   vs other form: max diff", np.nanmax(d), "differing", len(bad), "first", bad[:6].tolist(), "blocks", sorted(set(bad[:, 0].tolist()))[:70], "positions", sorted(set(bad[:, 1].tolist()))[:20])
