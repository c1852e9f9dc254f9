"""One-off fuzz of the recurrent decoders on the tuned kernels (GRU: gru_rec_h + gru_l1f; LSTM / RNN: rnn_rec_u + rnn_proj_u) against the
oracle: random cell, batch, block length, width, num_iter_ft, iterations, extrinsic, dec_act.  python tools/lab/probes/fuzz_rnn_u.py [n] [seed]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", ".."))
import numpy as np, torch
torch.set_num_threads(16)
from turboae_amd import TurboAEConfig, philox, weights as W, Channel_AE_HIP
from oracle import turboae_oracle as O

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
dev = torch.device("cuda", 0)
worst = 0.0
for i in range(n):
    cell = ["gru", "lstm", "rnn"][i % 3]
    L = int(rng.choice([1, 2, 3, 4, 5, 16, 17, 33, 100, 127, int(rng.randint(1, 260))]))
    B = int(rng.choice([1, 2, 15, 16, 17, 31, 32, 33, 63, 65, int(rng.randint(1, 200))]))
    U = int(rng.choice([100, 100, int(rng.randint(1, 101))]))
    cfg = TurboAEConfig(decoder="TurboAE_rate3_rnn", dec_rnn=cell, block_len=L, dec_num_unit=U, num_iter_ft=int(rng.randint(1, 7)),
                        num_iteration=int(rng.randint(1, 4)), extrinsic=int(rng.randint(0, 2)),
                        dec_act=str(rng.choice(["linear", "elu", "tanh", "relu", "selu", "sigmoid"])), enc_num_unit=int(rng.choice([32, 64, 100])),
                        # r06: every fourth case puts the 2-layer GRU encoder (ENC_interRNN) in front - also of the LSTM / RNN decoders
                        encoder="TurboAE_rate3_rnn" if i % 4 == 3 else "TurboAE_rate3_cnn")
    assert not cfg.generic, cfg
    sd = W.generate_state_dict(cfg, seed=int(rng.randint(1, 1 << 30)), gain=1.0)
    u = philox.random_bits(100 + i, 0, B * L).reshape(B, L, 1)
    noise = (np.float32(O.snr_db2sigma(1.0)) * philox.random_normal(100 + i, 0, B * L * 3)).reshape(B, L, 3).astype(np.float32)
    xo, co = O.channel_ae_forward(torch.from_numpy(u), torch.from_numpy(noise), O.to_torch(sd), cfg.to_dict(), {})
    model = Channel_AE_HIP(cfg, sd, device=dev, max_batch=B)
    xd, codes = model(torch.from_numpy(u).to(dev), torch.from_numpy(noise).to(dev))
    dc = float(np.abs(codes.cpu().numpy() - co.numpy()).max()); dx = float(np.abs(xd.cpu().numpy() - xo.numpy()).max())
    mode = model.range_status()
    k = B // 2
    sub = torch.equal(model.dec((codes + torch.from_numpy(noise).to(dev))[k:k + 1].contiguous()), xd[k:k + 1])
    worst = max(worst, dx)
    flag = "" if (dc <= 1e-5 and dx <= 1e-5 and sub and mode == ("f16x2", False)) else "   <-- FAIL"      # TAE_DEBUG_KNOBS=1 TAE_RNN_L1=fused in the environment: the fused layer 1 at these (small) batches
    print(f"{i:3d} {cell:4s} B={B:3d} L={L:3d} U={U:3d} F={cfg.num_iter_ft} it={cfg.num_iteration} ex={cfg.extrinsic} act={cfg.dec_act:7s} dc={dc:.1e} dx={dx:.1e} sub={sub} {mode}{flag}", flush=True)
print("worst dx", worst)
