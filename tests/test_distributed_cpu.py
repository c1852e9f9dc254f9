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


def test_bench_guard_turns_sigterm_and_hangs_into_one_error_line(tmp_path):
    """bench.py's Guard (first-N>1-run hardening): a rank blocked in a call that never returns (stands in for a collective whose peer
    died) still reports - the launcher's SIGTERM reaches the watchdog thread through signal.set_wakeup_fd, a hung phase through its
    deadline; rank 0 prints ONE line with `error`, `rccl_ranks_seen` and every rank's last recorded phase; a non-zero rank stays
    silent while rank 0 lives and takes over when it does not."""
    import json
    import signal
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = (
        "import sys, time, argparse\n"
        f"sys.path.insert(0, {root!r})\n"
        "import bench\n"
        "rank, mode = int(sys.argv[1]), sys.argv[2]\n"
        "args = argparse.Namespace(block_len=100, steps=2, warmup=1, strong=False, precision='auto')\n"
        "g = bench.Guard(args, rank, 2)\n"
        "g.phase('pg_init'); g.phase('pg_ready')\n"
        "g.phase('first_collective', timeout=1.0 if mode == 'hang' else 60.0)\n"
        "print('READY', flush=True)\n"
        "time.sleep(120)\n")

    def run(rank, mode, peer):
        d = tmp_path / f"{rank}_{mode}_{peer['phase']}"
        d.mkdir()
        (d / f"rank{1 - rank}.json").write_text(json.dumps(dict(peer, rank=1 - rank)))
        env = dict(os.environ, TAE_BENCH_STATE_DIR=str(d))
        p = subprocess.Popen([sys.executable, "-c", script, str(rank), mode], stdout=subprocess.PIPE, text=True, env=env, cwd=root)
        assert p.stdout.readline().strip() == "READY"
        t0 = time.time()
        if mode == "term":
            p.send_signal(signal.SIGTERM)
        rest, _ = p.communicate(timeout=60)
        assert time.time() - t0 < 20.0 and p.returncode == 3
        return [json.loads(l) for l in rest.splitlines() if l.startswith("{")]

    dead = subprocess.Popen([sys.executable, "-c", "pass"])
    dead.wait()
    for mode in ("term", "hang"):
        lines = run(0, mode, {"pid": dead.pid, "phase": "timed_pass"})
        assert len(lines) == 1
        res = lines[0]
        assert res["value"] == 0.0 and res["n_gpus"] == 2 and res["unit"] == "bits/s" and res["metric"].startswith("decoded info bits/sec")
        assert res["rccl_ranks_seen"] == 2                       # both ranks had their process group up
        assert ("signal" if mode == "term" else "time limit") in res["error"]
        st = res["config"]["rank_states"]
        assert st[0]["phase"] == "failed" and st[0]["failed_in"] == "first_collective" and st[1]["phase"] == "timed_pass"
    # rank 1: silent while rank 0's process exists (rank 0 reports), the reporter when it does not
    assert run(1, "term", {"pid": os.getpid(), "phase": "timed_pass"}) == []
    lines = run(1, "term", {"pid": dead.pid, "phase": "pg_init"})
    assert len(lines) == 1 and lines[0]["rccl_ranks_seen"] == 1 and lines[0]["config"]["rank_states"][0]["phase"] == "pg_init"
