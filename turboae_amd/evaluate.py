"""Eval sweep of the reference restated for the HIP path: ``trainer.test`` (trainer.py:135-248).

Same protocol - ``snr_points`` SNRs from ``snr_test_start`` to ``snr_test_end`` (trainer.py:157-158),
``num_block / batch_size`` batches per SNR (trainer.py:165), BER = mean over batches of the per-batch
bit error rate, BLER likewise (trainer.py:176-177,215-216), the same printed lines
(``Test SNR <snr> with ber <x> with bler <y>``, trainer.py:217; final lists :230-235) and the
encoder-power epilogue (trainer.py:238-248) - with three deliberate differences:
  * inputs come from the counter-based Philox streams on the device instead of the unseeded host
    RNG (trainer.py:167-169), keyed by (seed, snr index, global block index);
  * the accidental extra forward per SNR point (trainer.py:194-213, dies with NameError and prints
    'no pos BER specified.') is not run;
  * with torch.distributed initialised every batch is sharded over the ranks by block; the
    power-constraint statistics and the error counts are all-reduced (turboae_amd/distributed.py), so
    the numbers equal the single-GPU ones;
  * the decoder runs once per GROUP of batches (``decode_group``): encoder, power constraint (per-batch
    statistics, as in the reference) and channel run batch by batch, the received blocks of the group are
    decoded in one call and the errors are counted per batch again.  The decoder never mixes blocks
    (tests/test_gpu_parity.py::test_decoder_is_block_independent_and_batch_ragged), so every number is the
    same as with one decoder call per batch; a batch of 500 alone cannot fill 256 CUs.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import torch

from . import channels
from .distributed import all_reduce_sum_, shard_bounds


def snr_db2sigma(snr_db: float) -> float:      # utils.py:69-70
    return 10 ** (-snr_db * 1.0 / 20)


def snr_sigma2db(sigma: float) -> float:       # utils.py:72-76
    return -20.0 * math.log(sigma, 10)


def test(model, snr_test_start: float = -1.5, snr_test_end: float = 4.0, snr_points: int = 12, num_block: int = 1000,
         batch_size: int = 100, seed: int = 20190001, verbose: bool = True, enc_power_epilogue: bool = True,
         decode_group: Optional[int] = None, hip_graph: bool = False) -> Dict[str, List[float]]:
    """model: turboae_amd.Channel_AE_HIP.  Returns {'snrs', 'ber', 'bler', 'bit_errors', 'block_errors', 'enc_power'}.
    decode_group: batches decoded per decoder call (None: enough for about 24 576 blocks per rank; 1: one call per batch).
    hip_graph: capture every SNR point (all of its launches, the all-reduces included) into one hipGraph and launch that
    (AWGN only: the other channels draw their noise from a torch generator).  The sweep is GPU-bound either way; the
    option exists because the entry points are capturable (no allocation, no synchronisation) and a launch-bound caller
    (tiny batches) can use it."""
    import torch.distributed as dist
    rank, world = 0, 1
    if dist.is_available() and dist.is_initialized():
        rank, world = dist.get_rank(), dist.get_world_size()
    say = print if (verbose and rank == 0) else (lambda *a, **k: None)
    L = model.cfg.block_len
    if snr_points > 1:
        step = (snr_test_end - snr_test_start) * 1.0 / (snr_points - 1)
    else:
        step = 0.0
    snrs = [step * i + snr_test_start for i in range(snr_points)]
    say("SNRS", snrs)
    num_test_batch = int(num_block / batch_size)
    lo, hi = shard_bounds(batch_size, rank, world)
    nloc = hi - lo
    dev = model.this_device
    if decode_group is None:
        decode_group = max(1, -(-24576 // max(nloc, 1)))
    decode_group = max(1, min(int(decode_group), max(num_test_batch, 1)))
    if hip_graph and model.cfg.channel != "awgn":
        raise ValueError("hip_graph=True needs channel='awgn' (device-side Philox inputs)")
    if hip_graph and world > 1 and dist.get_backend() != "nccl":
        raise ValueError("hip_graph=True with torch.distributed needs the nccl (RCCL) backend")
    ber_res, bler_res, bit_res, blk_res = [], [], [], []

    def run_point(si, snr, per_batch):
        for g0 in range(0, num_test_batch, decode_group):
            group_u, group_rx = [], []
            for batch_idx in range(g0, min(g0 + decode_group, num_test_batch)):
                first = (si * num_test_batch + batch_idx) * batch_size + lo       # global block index of this shard
                fading = None
                if nloc > 0:
                    u, noise = model.generate_inputs(nloc, snr, seed=seed, first_block=first)
                    if model.cfg.channel != "awgn":
                        # other channels: generate_noise restated on the device (turboae_amd/channels.py), one generator per
                        # (seed, global first block of the shard): shards of any world size draw independent streams
                        gen = torch.Generator(device=dev)
                        gen.manual_seed((seed * 1000003 + first) & 0x7FFFFFFFFFFFFFFF)
                        noise = channels.generate_noise((nloc, L, 3), model.cfg, snr, device=dev, generator=gen)
                        if model.cfg.channel == "fading":
                            fading = channels.rayleigh_fading((nloc, L, 3), device=dev, generator=gen)
                    x_tx, stats = model.encode_prenorm(u)
                else:
                    stats = torch.zeros(3, dtype=torch.float64, device=dev)
                all_reduce_sum_(stats)                                            # batch-global mean/std (encoders.py:107-108)
                if nloc > 0:
                    _, rx = model.normalize(x_tx, stats, noise, want_codes=False, fading=fading)
                    group_u.append(u)
                    group_rx.append(rx)
            if nloc > 0:
                x_dec = model.dec(group_rx[0] if len(group_rx) == 1 else torch.cat(group_rx))
                for i, u in enumerate(group_u):
                    model.count_errors(x_dec[i * nloc:(i + 1) * nloc], u, per_batch[g0 + i])
        all_reduce_sum_(per_batch)

    if hip_graph and nloc > 0:
        model.reserve(decode_group * nloc)           # no workspace growth (an allocation) inside a capture
    for si, snr in enumerate(snrs):
        # per-batch (bit errors, block errors) stay on the device; one host read per SNR point
        per_batch = torch.zeros((max(num_test_batch, 1), 2), dtype=torch.int64, device=dev)
        if hip_graph:
            torch.cuda.synchronize(dev)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                run_point(si, snr, per_batch)
            graph.replay()
            torch.cuda.synchronize(dev)
            del graph
        else:
            run_point(si, snr, per_batch)
        pb = per_batch.cpu().tolist()
        # BER / BLER = mean over batches of the per-batch rates (trainer.py:176-177,215-216), accumulated in the same order
        test_ber, test_bler = 0.0, 0.0
        for c in pb[:num_test_batch]:
            test_ber += c[0] / float(batch_size * L)                           # errors_ber, utils.py:6-18
            test_bler += c[1] / float(batch_size)                              # errors_bler, utils.py:49-66
        test_ber /= num_test_batch
        test_bler /= num_test_batch
        tot = per_batch.sum(dim=0)
        say("Test SNR", snr, "with ber ", float(test_ber), "with bler", float(test_bler))
        ber_res.append(float(test_ber))
        bler_res.append(float(test_bler))
        t = tot.cpu().tolist()
        bit_res.append(int(t[0]))
        blk_res.append(int(t[1]))
        model.check_range()          # fp16-split kernels: fail loudly if an activation left the fp16 range
    say("final results on SNRs ", snrs)
    say("BER", ber_res)
    say("BLER", bler_res)
    out = {"snrs": snrs, "ber": ber_res, "bler": bler_res, "bit_errors": bit_res, "block_errors": blk_res}
    if enc_power_epilogue:
        # trainer.py:238-248: mean over batches of std(model.enc(X)); 1.0 for the power-normalised encoder
        enc_power = 0.0
        for idx in range(num_test_batch):
            first = ((snr_points + 0) * num_test_batch + idx) * batch_size + lo
            stats = torch.zeros(3, dtype=torch.float64, device=dev)
            if nloc > 0:
                u, _ = model.generate_inputs(nloc, 0.0, seed=seed, first_block=first)
                x_tx, stats = model.encode_prenorm(u)
            all_reduce_sum_(stats)
            loc = torch.zeros(3, dtype=torch.float64, device=dev)
            if nloc > 0:
                codes, _ = model.normalize(x_tx, stats)
                c = codes.double()
                loc = torch.stack([c.sum(), (c * c).sum(), torch.tensor(float(c.numel()), dtype=torch.float64, device=dev)])
            all_reduce_sum_(loc)
            s, ss, n = loc.cpu().tolist()
            enc_power += math.sqrt(max((ss - s * s / n) / (n - 1.0), 0.0))
        enc_power /= float(num_test_batch)
        say("encoder power is", enc_power)
        adj = [snr_sigma2db(snr_db2sigma(item) / enc_power) for item in snrs]
        say("adjusted SNR should be", adj)
        out["enc_power"] = enc_power
        out["adjusted_snrs"] = adj
    return out
