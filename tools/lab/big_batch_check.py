"""Index-width check at BASELINE configs[3] scale on ONE GPU: block_len 1000, 200 000 blocks (2e8 positions, 6e8 floats per
code tensor, 1.6e9-float exchange buffers) and block_len 100, 2 000 000 blocks.  Decoder sub-batches must reproduce the
big call bit for bit (any 32-bit index overflow would break the tail):  python tools/lab/big_batch_check.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from turboae_amd import TurboAEConfig, Channel_AE_HIP, weights as W
dev = torch.device("cuda", 0)
for L, B, it in ((1000, 200000, 6), (100, 2000000, 6)):
    cfg = TurboAEConfig(block_len=L, num_iteration=it)
    sd = W.generate_state_dict(cfg, seed=3, gain=1.0)
    model = Channel_AE_HIP(cfg, sd, device=dev, max_batch=B)
    u, noise = model.generate_inputs(B, 2.0, seed=1)
    t0 = time.time()
    x_dec, codes = model(u, noise)
    torch.cuda.synchronize()
    dt = time.time() - t0
    assert bool(torch.isfinite(x_dec).all()) and bool(torch.isfinite(codes).all())
    m, s = float(codes.double().mean()), float(codes.double().std())
    assert abs(m) < 1e-6 and abs(s - 1.0) < 1e-6, (m, s)                 # power constraint over the whole batch
    rx = codes + noise
    full = model.dec(rx)
    assert torch.equal(full, x_dec)
    for lo, hi in ((0, 7), (B // 2 - 3, B // 2 + 5), (B - 9, B)):
        assert torch.equal(model.dec(rx[lo:hi].contiguous()), full[lo:hi]), (L, lo, hi)
    part = model.enc(u[B - 1000:].contiguous())                           # encoder tail vs the big call: same blocks, other batch statistics
    x_big = codes[B - 1000:]
    scale = float((x_big * part).sum() / (part * part).sum())
    assert float((x_big - scale * part - (x_big - scale * part).mean()).abs().max()) < 1e-3
    counts = model.count_errors(x_dec, u).cpu().tolist()
    model.check_range()
    print(f"L={L} B={B}: forward {dt:.2f} s = {B * L / dt / 1e6:.1f} Mbit/s, BER {counts[0] / (B * L):.4f}, codes mean {m:.2e} std {s:.6f}: ok", flush=True)
    del model, u, noise, x_dec, codes, rx, full
    torch.cuda.empty_cache()
