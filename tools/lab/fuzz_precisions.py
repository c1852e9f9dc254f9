import sys; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from turboae_amd import TurboAEConfig, Channel_AE_HIP, weights as W
dev = torch.device("cuda", 0)
worst = 0.0
for L, it in ((100, 2), (40, 1), (64, 2), (150, 1), (330, 1), (1000, 1)):
    base = dict(block_len=L, num_iteration=it)
    sd = W.generate_state_dict(TurboAEConfig(**base), seed=L, gain=1.0)
    ma = Channel_AE_HIP(TurboAEConfig(**base), sd, device=dev, max_batch=64)
    mf = Channel_AE_HIP(TurboAEConfig(precision="f32", **base), sd, device=dev, max_batch=64)
    for B in (1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 17, 31, 64):
        u, n = ma.generate_inputs(B, 1.0, seed=B)
        xa, ca = ma(u, n); xf, cf = mf(u, n)
        d = max(float((xa - xf).abs().max()), float((ca - cf).abs().max()))
        worst = max(worst, d)
        assert d < 2e-5, (L, B, d)
    ma.check_range()
    print("L", L, "ok; kernel_info", ma.kernel_info(), ma.range_status()[0], flush=True)
print("worst |f16x2 - f32| =", worst)
