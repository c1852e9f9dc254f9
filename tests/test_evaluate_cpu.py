"""turboae_amd/evaluate.py::test (the reference's trainer.test restated) as HOST LOGIC on CPU: an oracle-backed stand-in
for Channel_AE_HIP (tests may use the oracle as a checker; the product path never does) serves the calls evaluate.test makes,
so the sweep arithmetic, the printed transcript and the sharding over ranks run in the CPU tier:

  * the lines evaluate.test prints have the shape of the REFERENCE's transcript (tests/golden/caller_trainer_test.json,
    recorded from the reference's unmodified trainer.test, oracle/make_caller_fixture.py) - default flags and
    --print_pos_ber --print_pos_power;
  * BER / BLER are the mean of per-batch means of exactly the blocks a direct oracle forward decodes;
  * two gloo ranks (world_size 2, ragged shards) reproduce the single-process numbers."""
import contextlib
import io
import json
import os
import re
import socket

import numpy as np
import pytest
import torch

from oracle import turboae_oracle as O
from turboae_amd import TurboAEConfig, evaluate, philox, weights as W
from turboae_amd.distributed import mean_std_from_stats, stats_from_tensor

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
with open(os.path.join(GOLD, "caller_trainer_test.json")) as _fh:
    FIX = json.load(_fh)
NUM = re.compile(r"[-+]?(?:\d+\.\d*|\.\d+|\d+)(?:[eE][-+]?\d+)?")


def _shape(lines):
    return re.sub(r"(?:# ?)+", "# ", re.sub(r"\s+", " ", NUM.sub("#", " ".join(lines)))).replace("# ]", "#]").strip()


class OracleModel:
    """The slice of Channel_AE_HIP that evaluate.test touches, computed by oracle/turboae_oracle.py on the CPU."""

    def __init__(self, cfg, sd):
        self.cfg = cfg
        self.this_device = torch.device("cpu")
        self.w = O.to_torch(sd)
        self.p = torch.from_numpy(O.rand_interleaver(cfg.block_len, 0))

    def generate_inputs(self, B, snr_db, seed, first_block=0, seed_noise=None):
        L = self.cfg.block_len
        u = philox.random_bits(seed, first_block * L, B * L).reshape(B, L, 1)
        z = philox.random_normal(seed if seed_noise is None else seed_noise, first_block * L * 3, B * L * 3).reshape(B, L, 3)
        return torch.from_numpy(u), torch.from_numpy((np.float32(O.snr_db2sigma(snr_db)) * z).astype(np.float32))

    def encode_prenorm(self, u):
        with torch.no_grad():
            x = O.encode_prenorm(u, self.w, self.p, self.cfg.enc_num_layer)
        return x, stats_from_tensor(x)

    def normalize(self, x_tx, stats, fwd_noise=None, want_codes=True, fading=None):
        mean, std = mean_std_from_stats(stats)
        codes = (x_tx - np.float32(mean)) / np.float32(std)
        return (codes if want_codes else None), (None if fwd_noise is None else codes + fwd_noise)

    def dec(self, rx):
        with torch.no_grad():
            return O.decode(rx, self.w, self.p, self.cfg.dec_num_layer, self.cfg.num_iteration, self.cfg.num_iter_ft)

    def count_errors(self, x_dec, u, counts):
        a, b = O.error_counts(u, x_dec)
        counts += torch.tensor([a, b], dtype=torch.int64)
        return counts

    def update_precomp(self, stats):
        pass

    def check_range(self):
        pass

    def reserve(self, n):
        pass


def _model():
    cfg = TurboAEConfig()
    sd = W.unpack_blob(cfg, np.load(os.path.join(GOLD, "trained_enc2dec5_u100_fp32.npz"))["weights_fp32"])
    return OracleModel(cfg, sd), cfg


@pytest.mark.parametrize("run", [r for r in sorted(FIX["runs"]) if not FIX["runs"][r]["args"]["precompute_norm_stats"]])
def test_transcript_shape_and_sweep_arithmetic(run):
    torch.set_num_threads(4)
    rec = FIX["runs"][run]
    a = rec["args"]
    model, cfg = _model()
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        res = evaluate.test_from_args(model, type("Args", (), a)(), seed=5)
    got, want = _shape(buf.getvalue().splitlines()), _shape(rec["transcript"])
    assert got == want, f"\n{got}\n!=\n{want}"
    # BER / BLER = mean over batches of per-batch means (trainer.py:176-177,215-216) of a direct forward on the same blocks
    B, L, nb = a["batch_size"], cfg.block_len, a["num_block"] // a["batch_size"]
    for si, snr in enumerate(res["snrs"]):
        ber = bler = 0.0
        for i in range(nb):
            u, noise = model.generate_inputs(B, snr, 5, first_block=(si * nb + i) * B)
            x, _ = O.channel_ae_forward(u, noise, model.w, cfg.to_dict())
            ber += O.errors_ber(u, x) / nb
            bler += O.errors_bler(u, x) / nb
        assert res["ber"][si] == pytest.approx(ber, abs=1e-7) and res["bler"][si] == pytest.approx(bler, abs=1e-7)     # errors_ber returns a float32 mean
    assert res["enc_power"] == pytest.approx(1.0, abs=1e-5)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


SWEEP = dict(snr_test_start=1.0, snr_test_end=3.0, snr_points=2, num_block=42, batch_size=21, seed=11, verbose=False)


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    model, _ = _model()
    q.put((rank, evaluate.test(model, **SWEEP)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_gloo_ranks_reproduce_the_single_process_sweep():
    import torch.multiprocessing as mp
    torch.set_num_threads(4)
    model, _ = _model()
    single = evaluate.test(model, **SWEEP)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=600) for _ in range(2))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for r in range(2):          # batch of 21 over 2 ranks: 11 + 10 blocks; statistics and counts all-reduced
        assert got[r]["bit_errors"] == single["bit_errors"] and got[r]["block_errors"] == single["block_errors"]
        assert got[r]["ber"] == pytest.approx(single["ber"], abs=1e-12)
        assert got[r]["enc_power"] == pytest.approx(single["enc_power"], abs=1e-6)
    assert single["bit_errors"][0] >= single["bit_errors"][1]
