"""Probe: does the GRU decoder (cfg 5) gain from running two half-batches on two HIP streams, offset by one recurrence, so that one
half's HBM-bound kernels (gru_proj_h, gru_head) overlap the other half's latency-bound recurrences?  Two handles, one per stream.
    python tools/gru_two_stream_probe.py [B]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from turboae_amd import TurboAEConfig, Channel_AE_HIP, weights as W
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
dev = torch.device("cuda", 0)
cfg = TurboAEConfig(decoder="TurboAE_rate3_rnn")
sd = W.generate_state_dict(cfg, seed=20190001, gain=1.0)
full = Channel_AE_HIP(cfg, sd, device=dev, max_batch=B)
u, noise = full.generate_inputs(B, 2.0, seed=1)
rx = full.enc(u) + noise
def timeit(fn, n=3):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return float(np.median(ts))
t_full = timeit(lambda: full.dec(rx))
print(f"one stream, {B} blocks: decoder {t_full:.2f} ms ({B*100/t_full/1e3:.2f} Mbit/s decoder-only)", flush=True)
h = B // 2
ma = Channel_AE_HIP(cfg, sd, device=dev, max_batch=h)
mb = Channel_AE_HIP(cfg, sd, device=dev, max_batch=h)
ra, rb = rx[:h].contiguous(), rx[h:].contiguous()
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
ref = full.dec(rx)
for delay_ms in (0.0, 0.4, 0.8, 1.2, 2.0):
    out = {}
    def both():
        cur = torch.cuda.current_stream()
        sa.wait_stream(cur); sb.wait_stream(cur)
        with torch.cuda.stream(sa):
            out["a"] = ma.dec(ra)
        with torch.cuda.stream(sb):
            if delay_ms > 0: torch.cuda._sleep(int(delay_ms * 1e-3 * 2.1e9))
            out["b"] = mb.dec(rb)
        cur.wait_stream(sa); cur.wait_stream(sb)
    t2 = timeit(both)
    ok = torch.equal(torch.cat([out["a"], out["b"]]), ref)
    print(f"two streams x {h} blocks, second delayed {delay_ms:.1f} ms: {t2:.2f} ms ({B*100/t2/1e3:.2f} Mbit/s)  speed-up {t_full/t2:.2f}x  bit-identical {ok}", flush=True)
