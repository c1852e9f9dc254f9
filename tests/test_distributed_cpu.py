"""world_size-2 gloo tests (CPU) of the N>1 sharding logic: the all-reduced power-constraint
statistics reproduce the reference's single-batch normalisation, error counts add up, and the
Philox streams are shard-consistent.  The oracle stands in for the per-shard encoder here (tests
may use it as a checker); the HIP kernels themselves are covered by the -m gpu tier."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import turboae_oracle as O
from turboae_amd import TurboAEConfig, philox, weights as W
from turboae_amd.distributed import all_reduce_sum_, mean_std_from_stats, shard_bounds, stats_from_tensor


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, B, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    cfg = TurboAEConfig(enc_num_unit=32, dec_num_unit=32, num_iteration=2)
    sd = O.to_torch(W.generate_state_dict(cfg, seed=3))
    L = cfg.block_len
    lo, hi = shard_bounds(B, rank, world)
    # shard-keyed inputs: the shard of the global stream
    u = torch.from_numpy(philox.random_bits(9, lo * L, (hi - lo) * L).reshape(hi - lo, L, 1))
    noise = torch.from_numpy(philox.random_normal(9, lo * L * 3, (hi - lo) * L * 3).reshape(hi - lo, L, 3))
    p = torch.from_numpy(O.rand_interleaver(L, 0))
    with torch.no_grad():
        x_tx = O.encode_prenorm(u, sd, p, cfg.enc_num_layer)
        stats = all_reduce_sum_(stats_from_tensor(x_tx))
        mean, std = mean_std_from_stats(stats)
        codes = (x_tx - mean) / std
        x_dec = O.decode(codes + noise, sd, p, cfg.dec_num_layer, cfg.num_iteration, cfg.num_iter_ft)
        be, ble = O.error_counts(u, x_dec)
        counts = all_reduce_sum_(torch.tensor([be, ble], dtype=torch.int64))
    q.put((rank, lo, hi, codes.numpy(), x_dec.numpy(), counts.tolist(), float(stats[2])))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_normalisation_equals_global_batch():
    world, B = 2, 7                      # ragged: 4 + 3 blocks
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, B, q)) for r in range(world)]
    for p_ in procs:
        p_.start()
    res = sorted([q.get(timeout=300) for _ in range(world)])
    for p_ in procs:
        p_.join(timeout=60)
        assert p_.exitcode == 0
    cfg = TurboAEConfig(enc_num_unit=32, dec_num_unit=32, num_iteration=2)
    sd = O.to_torch(W.generate_state_dict(cfg, seed=3))
    L = cfg.block_len
    u = torch.from_numpy(philox.random_bits(9, 0, B * L).reshape(B, L, 1))
    noise = torch.from_numpy(philox.random_normal(9, 0, B * L * 3).reshape(B, L, 3))
    x_ref, c_ref = O.channel_ae_forward(u, noise, sd, cfg.to_dict())
    codes = np.concatenate([r[3] for r in res])
    x_dec = np.concatenate([r[4] for r in res])
    assert [(r[1], r[2]) for r in res] == [(0, 4), (4, 7)]
    assert res[0][6] == B * L * 3
    assert np.abs(codes - c_ref.numpy()).max() <= 2e-6          # global-batch power constraint reproduced
    assert np.abs(x_dec - x_ref.numpy()).max() <= 5e-6
    be, ble = O.error_counts(u, x_ref)
    assert res[0][5] == res[1][5] == [be, ble]


def test_shard_bounds_cover_everything():
    for n in (0, 1, 7, 500, 50000):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def test_mean_std_from_stats_matches_torch():
    x = torch.randn(5, 100, 3) * 0.4 + 0.1
    m, s = mean_std_from_stats(stats_from_tensor(x))
    assert m == pytest.approx(float(x.mean()), abs=1e-7)
    assert s == pytest.approx(float(x.std()), rel=1e-6)
