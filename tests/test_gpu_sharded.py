"""The N > 1 path on real kernels: two ranks (sharing the one GPU of the test box, collectives over gloo) run the sharded
eval sweep and must reproduce the single-process numbers - the power-constraint statistics and the error counts are
all-reduced, everything else is per block.  (The 8-GPU RCCL run is the driver's; this checks the sharding logic end to
end through the C ABI.)"""
import os
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _load_model(dev, max_batch, channel="awgn"):
    import json
    from dataclasses import replace
    from turboae_amd import TurboAEConfig, Channel_AE_HIP, weights as W
    manifest = json.load(open(os.path.join(GOLD, "MANIFEST.json")))
    g = np.load(os.path.join(GOLD, "trained_enc2dec5_u100.npz"))
    cfg = replace(TurboAEConfig(**manifest["trained"]["config"]), channel=channel)
    return Channel_AE_HIP(cfg, W.unpack_blob(cfg, g["weights_fp16"].astype(np.float32)), device=dev, max_batch=max_batch)


SWEEP = dict(snr_test_start=0.0, snr_test_end=2.0, snr_points=2, num_block=600, batch_size=150, seed=123, verbose=False)


def _worker(rank, world, port, q, channel="awgn", sweep=None):
    import torch.distributed as dist
    from turboae_amd import evaluate
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    model = _load_model(dev, 150, channel)
    res = evaluate.test(model, **(sweep or SWEEP))
    q.put((rank, res))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_eval_sweep_equals_single_process(gpu_device, world):
    import torch.multiprocessing as mp
    from turboae_amd import evaluate
    single = evaluate.test(_load_model(gpu_device, 150), **SWEEP)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=600) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for r in range(world):
        res = got[r]
        # identical decisions up to logits within fp32 noise of zero: the shard statistics are summed in another order
        for si in range(2):
            assert abs(res["bit_errors"][si] - single["bit_errors"][si]) <= 2, (r, si)
            assert abs(res["block_errors"][si] - single["block_errors"][si]) <= 1, (r, si)
            assert abs(res["ber"][si] - single["ber"][si]) <= 1e-4
        assert abs(res["enc_power"] - single["enc_power"]) <= 1e-6
    assert single["bit_errors"][0] > single["bit_errors"][1] > 0


@pytest.mark.parametrize("channel,lo,hi", [("ge_awgn", 4.0, 0.0), ("bsc", 0.02, 0.15), ("fading", 6.0, 2.0)])
def test_sharded_sweep_on_other_channels_equals_single_process(gpu_device, channel, lo, hi):
    """The noise of every channel is drawn by a device kernel keyed by the GLOBAL block index (tae_generate_noise): two ranks sharing a
    batch draw exactly the blocks one process draws - Gilbert-Elliott chains, masks and fading coefficients included."""
    import torch.multiprocessing as mp
    from turboae_amd import evaluate
    sweep = dict(SWEEP, snr_test_start=lo, snr_test_end=hi)
    single = evaluate.test(_load_model(gpu_device, 150, channel), **sweep)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, channel, sweep)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=600) for _ in range(2))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for r in range(2):
        for si in range(2):
            assert abs(got[r]["bit_errors"][si] - single["bit_errors"][si]) <= 2, (channel, r, si)
            assert abs(got[r]["block_errors"][si] - single["block_errors"][si]) <= 1, (channel, r, si)
    assert single["bit_errors"][1] > single["bit_errors"][0]


def test_bench_runs_with_two_ranks(gpu_device):
    """bench.py's N > 1 path (stats all-reduce per step, barrier + max-over-ranks timing, error-count reduce, one JSON line
    from rank 0) launched exactly as the driver launches it, with the gloo test hook so both ranks can share this box's GPU."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, TAE_BENCH_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--batch", "3000"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                   # rank 0 only
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["steps"] == 2 and res["warmup"] == 1 and res["scaling"] == "weak"
    assert res["config"]["global_blocks"] == 6000 and res["config"]["blocks_per_gpu"] == 3000
    assert res["value"] > 0 and abs(res["value"] - 6000 * 100 * 2 / (res["ms_per_step"] * 2e-3)) <= 1e-3 * res["value"]
    assert "cpu_baseline" not in res and res["roofline"]["frac"] > 0
    # diagnostics for the first real multi-GPU run: every rank reached the collectives, and every rank's own clock is in the line
    assert res["config"]["rccl_ranks_seen"] == 2 and res["config"]["collective_backend"] == "gloo"
    pr = res["config"]["per_rank_ms_per_step"]
    assert len(pr["all"]) == 2 and pr["min"] <= pr["max"] and abs(pr["max"] - res["ms_per_step"]) <= 1e-6 * res["ms_per_step"] + 1e-3
    assert "other_configs" not in res["roofline"] and res["roofline"]["sustained_probe_tflops"] > 500      # probe on rank 0, other configs at N = 1 only


def test_plain_bench_command_relaunches_itself_for_n_gpus(gpu_device):
    """`python bench.py --gpus 2` WITHOUT a torch.distributed environment (how the driver starts N = 1; nobody knows
    how it will start N = 8) re-executes itself under torch.distributed.run with one rank per GPU and prints the same single line."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["TAE_BENCH_BACKEND"] = "gloo"                    # both ranks share this box's one GPU
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "2000"],
                         capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["config"]["global_blocks"] == 4000 and res["config"]["rccl_ranks_seen"] == 2


def test_bench_runs_with_eight_ranks_strong_and_ragged(gpu_device):
    """The driver's widest launch shape - 8 ranks - on this box's one GPU (gloo hook): a global batch that does not divide by 8
    (--strong 4003 -> shards of 500 / 501 blocks) decodes to the same BER as ONE process decoding the same 4003 blocks, and the
    line's throughput is the global block count over the max-over-ranks time."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = [os.path.join(root, "bench.py"), "--steps", "2", "--warmup", "1", "--batch", "4003", "--strong", "--no-f32-pass",
              "--no-cpu-baseline", "--no-parity"]
    env = dict(os.environ, TAE_BENCH_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port())] + common[:1] + ["--gpus", "8"] + common[1:]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    res = json.loads(lines[0])
    assert res["n_gpus"] == 8 and res["scaling"] == "strong" and res["config"]["global_blocks"] == 4003
    assert res["config"]["rccl_ranks_seen"] == 8 and len(res["config"]["per_rank_ms_per_step"]["all"]) == 8
    assert abs(res["value"] - 4003 * 100 / (res["ms_per_step"] * 1e-3)) <= 1e-3 * res["value"]
    one = subprocess.run([sys.executable] + common[:1] + ["--gpus", "1"] + common[1:], capture_output=True, text=True, timeout=600, cwd=root,
                         env={k: v for k, v in os.environ.items() if k not in ("TAE_BENCH_BACKEND", "TAE_BENCH_FORCE_DIST")})
    assert one.returncode == 0, one.stderr[-2000:]
    ref = json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][0])
    assert ref["config"]["global_blocks"] == 4003
    assert res["ber"] == ref["ber"] and res["bler"] == ref["bler"]          # same blocks, same global statistics, whatever the sharding


def test_bench_rccl_backend_at_world_size_1(gpu_device):
    """The RCCL ("nccl") branch of bench.py - process-group init bound to the device, stats all-reduce per step, barrier,
    max-over-ranks and error-count all-reduces - executed for real under torch.distributed.run (TAE_BENCH_FORCE_DIST=1 keeps the
    collectives on at world size 1; N > 1 needs the driver's multi-GPU node), plus --strong."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, TAE_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("TAE_BENCH_BACKEND", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
           "--batch", "3000", "--strong", "--no-cpu-baseline", "--no-f32-pass"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    res = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert res["n_gpus"] == 1 and res["scaling"] == "strong" and res["config"]["global_blocks"] == 3000
    assert res["value"] > 0 and 1e-4 < res["ber"] < 0.03            # trained weights at 2 dB (chance level would be 0.45)
    assert res["parity"]["ber_gpu_first500"] > 0


def _graph_worker(port, q):
    import torch.distributed as dist
    from turboae_amd import evaluate
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)         # RCCL
    model = _load_model(dev, 150)
    eager = evaluate.test(model, **SWEEP)
    graphed = evaluate.test(model, hip_graph=True, **SWEEP)                      # the RCCL all-reduces are captured into the graph
    q.put((eager, graphed))
    dist.barrier()
    dist.destroy_process_group()


def test_eval_sweep_hip_graph_captures_rccl_allreduce(gpu_device):
    """One hipGraph per SNR point (north_star) with the collectives INSIDE the capture: the power-constraint statistics
    and error-count all-reduces run on the RCCL backend (world size 1 on this box) and the graphed sweep returns the
    eager numbers."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_graph_worker, args=(_free_port(), q))
    p.start()
    eager, graphed = q.get(timeout=600)
    p.join(timeout=120)
    assert p.exitcode == 0
    assert eager["bit_errors"] == graphed["bit_errors"] and eager["block_errors"] == graphed["block_errors"]
    assert eager["ber"] == graphed["ber"] and eager["bit_errors"][0] > eager["bit_errors"][1] > 0


# ---- first-N>1-run hardening: every way a multi-rank launch can die still ends in ONE parsed JSON line ----------

def _bench_lines(out):
    import json
    return [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]


@pytest.mark.parametrize("victim", [1, 0])
def test_bench_rank_dies_mid_run_still_prints_one_error_line(gpu_device, victim):
    """A rank exits hard when it enters the timed pass (test hook TAE_BENCH_TEST_DIE): the launcher SIGTERMs the survivor while it
    sits inside a collective; its watchdog thread (signal.set_wakeup_fd) prints the error line - rank 0 if it is alive, otherwise
    the lowest rank that is - with every rank's last phase, within a minute."""
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, TAE_BENCH_BACKEND="gloo", TAE_BENCH_TEST_DIE=f"{victim}:timed_pass")
    env.pop("TAE_BENCH_STATE_DIR", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "2000"]
    t0 = time.time()
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env, cwd=root)
    assert time.time() - t0 < 60.0
    assert out.returncode != 0
    lines = _bench_lines(out)
    assert len(lines) == 1, (out.stdout[-1500:], out.stderr[-1500:])
    res = lines[0]
    assert res["value"] == 0.0 and res["n_gpus"] == 2 and "error" in res and res["rccl_ranks_seen"] == 2
    st = res["config"]["rank_states"]
    assert st[victim]["phase"] == "timed_pass"                       # the last thing the dead rank recorded
    # the survivor failed inside the timed pass: either its collective raised (gloo notices the closed connection) or the launcher's
    # SIGTERM reached its watchdog first
    assert st[1 - victim]["phase"] == "failed" and st[1 - victim]["failed_in"] in ("collectives_ok", "timed_pass")     # (rank 0 runs the MFMA probe first)
    assert "signal" in st[1 - victim]["error"] or "Error" in st[1 - victim]["error"]


def test_bench_port_busy_prints_an_error_line(gpu_device):
    """Rendezvous port already taken (ranks started by hand, as a scheduler without torch.distributed.run would): rank 0's store
    cannot bind, init_process_group raises, the line says so."""
    import socket
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as holder:
        holder.bind(("127.0.0.1", 0))
        holder.listen(1)
        port = holder.getsockname()[1]
        env = dict(os.environ, TAE_BENCH_BACKEND="gloo", RANK="0", LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        t0 = time.time()
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "2000",
                              "--dist-timeout", "20"], capture_output=True, text=True, timeout=300, env=env, cwd=root)
        assert time.time() - t0 < 60.0
    lines = _bench_lines(out)
    assert out.returncode != 0 and len(lines) == 1, (out.stdout[-1500:], out.stderr[-1500:])
    assert lines[0]["value"] == 0.0 and "error" in lines[0] and lines[0]["rccl_ranks_seen"] == 0
    assert lines[0]["config"]["rank_states"][0]["failed_in"] == "pg_init"


def test_bench_fewer_devices_than_ranks_prints_an_error_line(gpu_device):
    """`python bench.py --gpus N` on a box with fewer than N GPUs (RCCL backend: one GPU per rank): every rank refuses before the
    rendezvous, rank 0 prints the line, the self-relaunching parent passes it through."""
    import subprocess
    import sys
    import time
    n = torch.cuda.device_count() + 1
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "TAE_BENCH_BACKEND", "TAE_BENCH_STATE_DIR")}
    t0 = time.time()
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(n), "--steps", "2", "--warmup", "1"],
                         capture_output=True, text=True, timeout=300, env=env, cwd=root)
    assert time.time() - t0 < 60.0
    lines = _bench_lines(out)
    assert out.returncode != 0 and len(lines) == 1, (out.stdout[-1500:], out.stderr[-1500:])
    assert lines[0]["value"] == 0.0 and lines[0]["n_gpus"] == n and "device_count" in lines[0]["error"]


def test_bench_also_configs3_in_the_same_launch(gpu_device):
    """N > 1 runs BASELINE configs[3] (block_len 1000: 25 000 blocks per rank, and the --strong form, 200 000 blocks over the ranks)
    after the headline pass and reports it inside the ONE line (`configs3`, flat `cfg3_*` scalars) - one multi-GPU lease yields both
    configurations BASELINE.json names.  Two ranks share this box's GPU through the gloo hook."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, TAE_BENCH_BACKEND="gloo")
    env.pop("TAE_BENCH_STATE_DIR", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "2000",
           "--no-f32-pass"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = _bench_lines(out)
    assert len(lines) == 1
    res = lines[0]
    assert res["n_gpus"] == 2 and res["value"] > 0 and "error" not in res
    c3 = res["configs3"]
    assert c3["weak"]["global_blocks"] == 50000 and c3["strong"]["global_blocks"] == 200000
    for tag in ("weak", "strong"):
        assert c3[tag]["rccl_ranks_seen"] == 2 and 1e-4 < c3[tag]["ber"] < 0.03 and c3[tag]["value"] > 0
        assert res[f"cfg3_{tag}_bits_per_s"] == c3[tag]["value"] and 0.05 < res[f"cfg3_{tag}_decoder_frac"] < 1.0
    # the flat block that closes the line carries 5 significant digits (bench.ordered_for_tail)
    assert res["roofline_frac"] == pytest.approx(res["roofline"]["frac"], rel=1e-4) and res["overrides"] == ""
    assert list(res)[-1] == "overrides"
