"""Numerics probe (CPU, no GPU): would a minimal-filtering (Winograd / Cook-Toom) 1-D convolution keep the hot path inside its
parity bound?  DESIGN.md section 7 names it as the one lever left for the decoder (fewer multiplies per output: F(2,5) 6 instead of
10 per output pair, F(4,5) 8 instead of 20); its transforms amplify rounding, so this script measures by how much ON THIS NETWORK:
the reference-trained enc2/dec5 model of tests/golden, every 5-tap Conv1d replaced by F(m,5) evaluated in fp32 (transforms and the
per-point channel contractions in fp32, like an fp32-grade kernel would), against a float64 evaluation of the direct form.

    python tests/probes/winograd_numerics_probe.py [blocks]

Uses the oracle as the network definition; lives under tests/ because only test infrastructure may import oracle/."""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import turboae_oracle as O                                   # noqa: E402
from turboae_amd import TurboAEConfig, philox, weights as W             # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def cook_toom(m, r, points):
    """Matrices of Y = AT [(G g) * (BT d)] for the correlation y_k = sum_j g_j d_{k+j}, k < m, from n - 1 = m + r - 2 finite
    points plus infinity (Lavin & Gray / wincnn construction), in float64."""
    n = m + r - 1
    p = np.asarray(points, dtype=np.float64)
    assert p.size == n - 1
    AT = np.zeros((m, n))
    for k in range(m):
        AT[k, :n - 1] = p ** k
    AT[m - 1, n - 1] = 1.0
    G = np.zeros((n, r))
    for i in range(n - 1):
        f = np.prod([p[i] - p[j] for j in range(n - 1) if j != i])
        G[i, :] = p[i] ** np.arange(r) / f
    G[n - 1, r - 1] = 1.0
    # B: column i (< n-1) = coefficients of prod_{j != i} (x - p_j); last column = coefficients of prod_j (x - p_j)
    B = np.zeros((n, n))
    for i in range(n - 1):
        c = np.poly1d([1.0])
        for j in range(n - 1):
            if j != i:
                c = c * np.poly1d([1.0, -p[j]])
        B[:n - 1, i] = c.coeffs[::-1]
    c = np.poly1d([1.0])
    for j in range(n - 1):
        c = c * np.poly1d([1.0, -p[j]])
    B[:, n - 1] = c.coeffs[::-1]
    BT = B.T
    # self-check on random data (float64): must reproduce the direct correlation
    rng = np.random.RandomState(0)
    g, d = rng.randn(r), rng.randn(n)
    y = AT @ ((G @ g) * (BT @ d))
    y_ref = np.array([sum(g[j] * d[k + j] for j in range(r)) for k in range(m)])
    assert np.abs(y - y_ref).max() < 1e-9, (y, y_ref)
    return AT, G, BT


def make_conv(m, points, dtype):
    AT, G, BT = [torch.tensor(x, dtype=dtype) for x in cook_toom(m, 5, points)]
    n = m + 4

    def conv1d_same(x, wt, b):
        """x (B, C_in, L), wt (C_out, C_in, 5), zero padding 2 -> (B, C_out, L), every m outputs from one n-point tile"""
        Bn, Ci, L = x.shape
        Lp = (L + m - 1) // m * m
        xp = F.pad(x.to(dtype), (2, 2 + Lp - L))
        tiles = xp.unfold(2, n, m)                                       # (B, C_in, Lp / m, n)
        V = torch.einsum("pn,bctn->pbct", BT, tiles)                     # input transform
        U = torch.einsum("pr,oir->poi", G, wt.to(dtype))                 # filter transform (once per layer; host side in a kernel)
        M = torch.einsum("poi,pbit->pbot", U, V)                         # n independent C_in contractions
        Y = torch.einsum("kp,pbot->botk", AT, M).reshape(Bn, wt.shape[0], Lp)[:, :, :L]
        return Y + b.to(dtype).view(1, -1, 1)
    return conv1d_same


def forward(u, noise, w, cfg, conv, dtype):
    """Channel_AE.forward with every Conv1d going through `conv` (None: F.conv1d)."""
    real = F.conv1d

    def patched(h, wt, b, stride=1, padding=0):
        if conv is None or wt.shape[2] != 5:
            return real(h, wt, b, stride=stride, padding=padding)
        return conv(h, wt, b)
    F.conv1d = patched
    try:
        wd = {k: v.to(dtype) for k, v in w.items()}
        x, c = O.channel_ae_forward(u.to(dtype), noise.to(dtype), wd, cfg.to_dict())
    finally:
        F.conv1d = real
    return x.double(), c.double()


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    torch.set_num_threads(4)
    cfg = TurboAEConfig()
    sd = W.unpack_blob(cfg, np.load(os.path.join(GOLD, "trained_enc2dec5_u100_fp32.npz"))["weights_fp32"])
    w = O.to_torch(sd)
    L = cfg.block_len
    u = torch.from_numpy(philox.random_bits(7, 0, B * L).reshape(B, L, 1))
    noise = torch.from_numpy((np.float32(O.snr_db2sigma(2.0)) * philox.random_normal(7, 0, B * L * 3)).reshape(B, L, 3).astype(np.float32))
    x64, c64 = forward(u, noise, w, cfg, None, torch.float64)
    rows = [("direct conv, fp32 (the oracle)", None, torch.float32)]
    variants = {
        "F(2,5) points 0, +-1, +-1/2": (2, [0.0, 1.0, -1.0, 0.5, -0.5]),
        "F(2,5) points 0, +-1, +-2": (2, [0.0, 1.0, -1.0, 2.0, -2.0]),
        "F(4,5) points 0, +-1, +-1/2, +-2": (4, [0.0, 1.0, -1.0, 0.5, -0.5, 2.0, -2.0]),
    }
    for name, (m, pts) in variants.items():
        rows.append((name + ", fp32", make_conv(m, pts, torch.float32), torch.float32))
    rows.append(("F(4,5) points 0, +-1, +-1/2, +-2, float64 (algebra check)", make_conv(4, variants["F(4,5) points 0, +-1, +-1/2, +-2"][1], torch.float64), torch.float64))
    print(f"{B} blocks of {L}, trained enc2/dec5, 2 dB; deviations from a float64 evaluation of the direct form (parity bound: codes 1e-5, x_dec 2e-5)")
    for name, conv, dt in rows:
        x, c = forward(u, noise, w, cfg, conv, dt)
        flips = int(((x > 0.5) != (x64 > 0.5)).sum())
        print(f"  {name:58s} max|d codes| {float((c - c64).abs().max()):.2e}   max|d x_dec| {float((x - x64).abs().max()):.2e}   decision flips {flips}")


if __name__ == "__main__":
    main()
