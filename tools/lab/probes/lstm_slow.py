import sys, time, json, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", ".."))
import numpy as np, torch
from dataclasses import replace
from turboae_amd import TurboAEConfig, philox, weights as W, Channel_AE_HIP
from oracle import turboae_oracle as O
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "..", "tests", "golden")
man = json.load(open(GOLD + '/MANIFEST.json'))
dev = torch.device("cuda", 0)
if len(sys.argv) > 1:
    torch.set_num_threads(int(sys.argv[1]))
print("threads", torch.get_num_threads(), flush=True)
for kind, f in (("trained_cnn_gru_fp32", "trained_cnn_gru_u100_fp32.npz"), ("trained_cnn_lstm_fp32", "trained_cnn_lstm_u100_fp32.npz")):
    g = np.load(GOLD + '/' + f); cfg = TurboAEConfig(**man[kind]["config"]); sd = W.unpack_blob(cfg, g["weights_fp32"])
    B = 16; L = 100
    u = philox.random_bits(1, 0, B * L).reshape(B, L, 1); noise = (0.79 * philox.random_normal(1, 0, B * L * 3)).reshape(B, L, 3).astype(np.float32)
    for dt in (torch.float32, torch.float64):
        sdx = {k: torch.from_numpy(np.asarray(v)).to(dt) for k, v in sd.items()}
        t0 = time.time()
        x, c = O.channel_ae_forward(torch.from_numpy(u).to(dt), torch.from_numpy(noise).to(dt), sdx, cfg.to_dict())
        print(kind, "oracle", dt, round(time.time() - t0, 3), flush=True)
    for prec in ("auto", "f32"):
        t0 = time.time()
        model = Channel_AE_HIP(replace(cfg, precision=prec), sd, device=dev, max_batch=B)
        torch.cuda.synchronize(); t1 = time.time()
        xd, codes = model(torch.from_numpy(u).to(dev), torch.from_numpy(noise).to(dev))
        torch.cuda.synchronize(); t2 = time.time()
        print(kind, prec, "create", round(t1 - t0, 3), "forward", round(t2 - t1, 3), flush=True)
