"""Run-to-run determinism soak on the GPU box: the same inputs through every kernel family N times, outputs compared bit for bit
(r03: a mixed-shape MFMA hazard once made the GRU recurrence's results change from run to run - see DESIGN.md 3.5).
    python tools/determinism_soak.py [repeats=20]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from turboae_amd import TurboAEConfig, Channel_AE_HIP, weights as W

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda", 0)
cases = [("cnn f16x2 50000x100", TurboAEConfig(), 50000), ("cnn f32 20000x100", TurboAEConfig(precision="f32"), 20000),
         ("enc5 f16x2 20000x100", TurboAEConfig(enc_num_layer=5), 20000), ("L=1000 f16x2 5000", TurboAEConfig(block_len=1000), 5000),
         ("gru dec f16x2 16384", TurboAEConfig(decoder="TurboAE_rate3_rnn"), 16384), ("gru dec f32 4096", TurboAEConfig(decoder="TurboAE_rate3_rnn", precision="f32"), 4096),
         ("gru enc+dec f16x2 4096", TurboAEConfig(encoder="TurboAE_rate3_rnn", decoder="TurboAE_rate3_rnn", num_iteration=2), 4096),
         ("dense f16x2 2000", TurboAEConfig(encoder="TurboAE_rate3_cnn_dense", decoder="TurboAE_rate3_cnn_dense", num_iteration=2), 2000),
         ("B=500 f16x2", TurboAEConfig(), 500), ("generic lstm 64", TurboAEConfig(decoder="TurboAE_rate3_rnn", dec_rnn="lstm", dec_num_unit=32, num_iteration=2, precision="f32"), 64),
         ("generic lstm H=100 4096 (fp32 MFMA)", TurboAEConfig(decoder="TurboAE_rate3_rnn", dec_rnn="lstm", num_iteration=2, precision="f32"), 4096),
         ("lstm H=100 16400 (unit-split f16x2)", TurboAEConfig(decoder="TurboAE_rate3_rnn", dec_rnn="lstm", num_iteration=2), 16400),
         ("rnn H=100 8192 (unit-split f16x2)", TurboAEConfig(decoder="TurboAE_rate3_rnn", dec_rnn="rnn", num_iteration=2), 8192),
         ("generic widths 256 1024 (fp32 MFMA)", TurboAEConfig(enc_num_unit=256, dec_num_unit=256, num_iteration=2), 1024),
         ("f16x1 one-product decoder 20000x100 (r06)", TurboAEConfig(precision="f16x1"), 20000),
         ("gru enc + lstm dec f16x2 4096 (r06 pairing)", TurboAEConfig(encoder="TurboAE_rate3_rnn", decoder="TurboAE_rate3_rnn", dec_rnn="lstm", num_iteration=2), 4096),
         ("generic rnn H=130 2048 (vector ALU)", TurboAEConfig(decoder="TurboAE_rate3_rnn", dec_rnn="rnn", dec_num_unit=130, num_iteration=2), 2048)]
bad = 0
for name, cfg, B in cases:
    model = Channel_AE_HIP(cfg, W.generate_state_dict(cfg, seed=5, gain=1.0), device=dev, max_batch=B)
    u, noise = model.generate_inputs(B, 1.0, seed=3)
    x0, c0 = model(u, noise)
    diff = 0
    for _ in range(reps):
        x, c = model(u, noise)
        diff += int((x != x0).sum()) + int((c != c0).sum())
    print(f"{name}: {reps} repeats, differing values {diff}", flush=True)
    bad += diff
    del model
    torch.cuda.empty_cache()
sys.exit(1 if bad else 0)
