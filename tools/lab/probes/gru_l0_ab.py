"""Layer 0 of the f16x2 GRU decoder stacks: gru_rec_h_kernel<true> (one wave per 16 blocks) against gru_rec0u_kernel (seven waves per 16
blocks) - bit-identity of x_dec and forward time per batch size.  python tools/lab/probes/gru_l0_ab.py [B ...]"""
import os, sys, subprocess, hashlib
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", ".."))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    import numpy as np, torch
    from turboae_amd import TurboAEConfig, Channel_AE_HIP, weights as W
    B, L = int(sys.argv[2]), int(sys.argv[3])
    dev = torch.device("cuda", 0)
    cfg = TurboAEConfig(block_len=L, decoder="TurboAE_rate3_rnn")
    model = Channel_AE_HIP(cfg, W.generate_state_dict(cfg, seed=20190001, gain=1.0), device=dev, max_batch=B)
    u, noise = model.generate_inputs(B, 2.0, seed=1)
    for _ in range(2): xd, _ = model(u, noise)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); xd, codes = model(u, noise); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    print(f"{hashlib.sha1(xd.cpu().numpy().tobytes()).hexdigest()[:12]} {float(np.median(ts)):.3f}")
    sys.exit(0)
cases = [(int(a), 100) for a in sys.argv[1:]] or [(1, 100), (7, 3), (16, 100), (17, 37), (500, 100), (2048, 100), (4096, 100), (8192, 100), (16384, 100), (100, 1000)]
for B, L in cases:
    out = {}
    for mode in ("block", "unit"):
        env = dict(os.environ, TAE_DEBUG_KNOBS="1", TAE_GRU_L0=mode)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(B), str(L)], env=env, capture_output=True, text=True, timeout=600)
        out[mode] = r.stdout.strip().split() if r.returncode == 0 else ["ERR", r.stderr[-300:]]
    same = out["block"][0] == out["unit"][0]
    print(f"B={B:6d} L={L:4d}: block {out['block'][1]} ms, unit {out['unit'][1]} ms, x_dec identical: {same}  ({out['block'][0]})", flush=True)
