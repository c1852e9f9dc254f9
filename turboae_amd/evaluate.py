"""Eval sweep of the reference restated for the HIP path: ``trainer.test`` (trainer.py:135-248).

Same protocol - ``snr_points`` SNRs from ``snr_test_start`` to ``snr_test_end`` (trainer.py:157-158),
``num_block / batch_size`` batches per SNR (trainer.py:165), BER = mean over batches of the per-batch
bit error rate, BLER likewise (trainer.py:176-177,215-216), the same printed lines
(``Test SNR <snr> with ber <x> with bler <y>``, trainer.py:217; final lists :230-235) and the
encoder-power epilogue (trainer.py:238-248), the ``--precompute_norm_stats`` pre-pass (trainer.py:145-153) and the
``--print_pos_ber`` / ``--print_pos_power`` outputs (trainer.py:179-193) - with these deliberate differences:
  * inputs come from the counter-based Philox streams on the device instead of the unseeded host
    RNG (trainer.py:167-169), keyed by (seed, snr index, global block index);
  * the second, "punctured" pass per SNR point (trainer.py:194-213) runs only with ``print_pos_ber`` - without it the
    reference's own pass dies with a NameError on the first batch and prints 'no pos BER specified.';
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

import numpy as np
import torch

from .distributed import all_reduce_sum_, shard_bounds


def snr_db2sigma(snr_db: float) -> float:      # utils.py:69-70
    return 10 ** (-snr_db * 1.0 / 20)


def snr_sigma2db(sigma: float) -> float:       # utils.py:72-76
    return -20.0 * math.log(sigma, 10)


def test(model, snr_test_start: float = -1.5, snr_test_end: float = 4.0, snr_points: int = 12, num_block: int = 1000,
         batch_size: int = 100, seed: int = 20190001, verbose: bool = True, enc_power_epilogue: bool = True,
         decode_group: Optional[int] = None, hip_graph: bool = False, test_ratio: int = 1,
         print_pos_ber: bool = False, print_pos_power: bool = False, num_ber_puncture: int = 5) -> Dict[str, List[float]]:
    """model: turboae_amd.Channel_AE_HIP.  Returns {'snrs', 'ber', 'bler', 'bit_errors', 'block_errors', 'enc_power'}.
    decode_group: batches decoded per decoder call (None: enough for about 24 576 blocks per rank; 1: one call per batch).
    hip_graph: capture every SNR point (all of its launches, the all-reduces included) into one hipGraph and launch that
    (every channel: all noise generators are device kernels keyed by Philox counters).  The points are pipelined: the host captures
    point i + 2 while the device runs point i and has point i + 1 queued (with the positional statistics of --print_pos_ber /
    --print_pos_power, whose second pass needs the first one's ranking on the host, the points are captured one after another)."""
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
    num_test_batch = int(num_block / batch_size)
    lo, hi = shard_bounds(batch_size, rank, world)
    nloc = hi - lo
    dev = model.this_device
    if decode_group is None:
        decode_group = max(1, -(-24576 // max(nloc, 1)))
    decode_group = max(1, min(int(decode_group), max(num_test_batch, 1)))
    precomp = bool(model.cfg.precompute_norm_stats) and not model.cfg.no_code_norm
    if hip_graph and precomp:
        raise ValueError("hip_graph=True cannot be combined with precompute_norm_stats (the running statistics live on the host)")
    if precomp:
        # trainer.py:145-153: the encoder alone over int(num_block / batch_size * test_ratio) batches fills the running mean / std
        # (ENCBase.power_constraint, encoders.py:110-114); they keep averaging over every later call, as in the reference
        for idx in range(int(num_block / batch_size * test_ratio)):
            first = ((snr_points + 1) * num_test_batch + idx) * batch_size + lo
            stats = torch.zeros(3, dtype=torch.float64, device=dev)
            if nloc > 0:
                u, _ = model.generate_inputs(nloc, 0.0, seed=seed, first_block=first)
                _, stats = model.encode_prenorm(u)
            all_reduce_sum_(stats)
            model.update_precomp(stats)
        # the reference prints its (1,)-shaped running-statistics tensors (trainer.py:153)
        say("Pre-computed norm statistics mean ", torch.tensor([model._eng.mean_scalar], dtype=torch.float32),
            "std ", torch.tensor([model._eng.std_scalar], dtype=torch.float32))
    say("SNRS", snrs)                      # after the pre-pass line, as trainer.py:153-160 prints them
    if hip_graph and world > 1 and dist.get_backend() != "nccl":
        raise ValueError("hip_graph=True with torch.distributed needs the nccl (RCCL) backend")
    ber_res, bler_res, bit_res, blk_res = [], [], [], []
    pos_ber_res, pos_power_res, ber_punc_res, bler_punc_res = [], [], [], []
    # --print_pos_ber / --print_pos_power (trainer.py:179-193): per-position error rate (errors_ber_pos, utils.py:31-40) and
    # per-position code power (code_power, utils.py:42-48), both averaged over the batches of the SNR point
    want_pos = print_pos_ber or print_pos_power
    pos_err = torch.zeros(L, dtype=torch.float64, device=dev)
    pos_pow = torch.zeros(L, dtype=torch.float64, device=dev)

    def make_batch(first, snr):
        """(u, noise, fading) of this rank's shard of the batch whose first global block is `first`."""
        u, noise = model.generate_inputs(nloc, snr, seed=seed, first_block=first)
        fading = None
        if model.cfg.channel != "awgn":
            # other channels: generate_noise restated as a device kernel (tae_generate_noise), keyed like the AWGN draw by
            # (seed, global block index), so shards of any world size reproduce the single-device stream; `snr` is the
            # reference's test_sigma (an SNR in dB, or the erase / flip probability of bec / bsc / ge: trainer.py:160-169)
            noise, fading = model.generate_noise(nloc, snr, seed=seed, first_block=first)
        return u, noise, fading

    def punctured_point(si, snr, positions):
        """The second pass of trainer.py:194-213 (it only runs when --print_pos_ber defined the position ranking): fresh batches,
        errors at the `positions` with the worst positional BER are not counted (errors_ber / errors_bler with positions,
        utils.py:6-18,49-66 - the BER still divides by the full block length)."""
        keep = torch.ones(L, dtype=torch.bool, device=dev)
        keep[torch.as_tensor(positions, dtype=torch.long, device=dev)] = False
        acc = torch.zeros((max(num_test_batch, 1), 2), dtype=torch.int64, device=dev)
        for batch_idx in range(num_test_batch):
            first = ((snr_points + 2 + si) * num_test_batch + batch_idx) * batch_size + lo
            stats = torch.zeros(3, dtype=torch.float64, device=dev)
            if nloc > 0:
                u, noise, fading = make_batch(first, snr)
                # the reference draws this pass's noise with X_test.shape = (B, L, 1) (trainer.py:198) and codes + fwd_noise
                # (channel_ae.py:42) broadcasts it: all three code symbols of a position see the SAME noise value here
                noise = noise[:, :, 0:1].expand(-1, -1, 3).contiguous()
                x_tx, stats = model.encode_prenorm(u)
            all_reduce_sum_(stats)
            if precomp:
                model.update_precomp(stats)
            if nloc > 0:
                _, rx = model.normalize(x_tx, stats, noise, want_codes=False, fading=fading)
                err = ((model.dec(rx) > 0.5) != (u > 0.5)).squeeze(2) & keep
                acc[batch_idx, 0] = err.sum()
                acc[batch_idx, 1] = err.any(dim=1).sum()
        all_reduce_sum_(acc)
        pb = acc.cpu().tolist()
        ber = sum(c[0] / float(batch_size * L) for c in pb[:num_test_batch]) / num_test_batch
        bler = sum(c[1] / float(batch_size) for c in pb[:num_test_batch]) / num_test_batch
        return ber, bler

    def run_point(si, snr, per_batch):
        for g0 in range(0, num_test_batch, decode_group):
            group_u, group_rx = [], []
            for batch_idx in range(g0, min(g0 + decode_group, num_test_batch)):
                first = (si * num_test_batch + batch_idx) * batch_size + lo       # global block index of this shard
                fading = None
                if nloc > 0:
                    u, noise, fading = make_batch(first, snr)
                    x_tx, stats = model.encode_prenorm(u)
                else:
                    stats = torch.zeros(3, dtype=torch.float64, device=dev)
                all_reduce_sum_(stats)                                            # batch-global mean/std (encoders.py:107-108)
                if precomp:
                    model.update_precomp(stats)                                   # running averages (encoders.py:110-114)
                if nloc > 0:
                    codes, rx = model.normalize(x_tx, stats, noise, want_codes=print_pos_power, fading=fading)
                    if print_pos_power:
                        pos_pow.add_((codes.double() ** 2).sum(dim=2).sum(dim=0) / 3.0)
                    group_u.append(u)
                    group_rx.append(rx)
            if nloc > 0:
                x_dec = model.dec(group_rx[0] if len(group_rx) == 1 else torch.cat(group_rx))
                for i, u in enumerate(group_u):
                    model.count_errors(x_dec[i * nloc:(i + 1) * nloc], u, per_batch[g0 + i])
                    if print_pos_ber:      # torch.round is half-to-even: round(y) != round(x) <=> (y > 0.5) != (x > 0.5)
                        pos_err.add_(((x_dec[i * nloc:(i + 1) * nloc] > 0.5) != (u > 0.5)).sum(dim=0).squeeze(1).double())
        all_reduce_sum_(per_batch)
        if want_pos:
            all_reduce_sum_(pos_err)
            all_reduce_sum_(pos_pow)

    if hip_graph and nloc > 0:
        model.reserve(decode_group * nloc)           # no workspace growth (an allocation) inside a capture

    def report_point(si, snr, pb):
        """pb: the point's per-batch (bit errors, block errors) on the host.  Everything trainer.py:176-236 prints / collects for it."""
        # BER / BLER = mean over batches of the per-batch rates (trainer.py:176-177,215-216), accumulated in the same order
        test_ber, test_bler = 0.0, 0.0
        for c in pb[:num_test_batch]:
            test_ber += c[0] / float(batch_size * L)                           # errors_ber, utils.py:6-18
            test_bler += c[1] / float(batch_size)                              # errors_bler, utils.py:49-66
        test_ber /= num_test_batch
        test_bler /= num_test_batch
        # printed objects have the reference's types (numpy fp32 array / torch fp32 tensor), so the lines read the same
        if print_pos_power:
            pos_power_res.append((pos_pow / float(batch_size * num_test_batch)).cpu().tolist())
            say("code power", np.asarray(pos_power_res[-1], dtype=np.float32))            # code_power returns numpy (utils.py:42-48)
        if print_pos_ber:
            res_pos = pos_err / float(batch_size * num_test_batch)
            pos_ber_res.append(res_pos.cpu().tolist())
            say("positional ber", res_pos.float().cpu())
            res_pos_arg = torch.argsort(res_pos, descending=True, stable=True).cpu().tolist()
            say("positional argmax", res_pos_arg)
        ber_p, bler_p = 0.0, 0.0
        if print_pos_ber:
            ber_p, bler_p = punctured_point(si, snr, res_pos_arg[:num_ber_puncture])
        else:
            # trainer.py:194-213 without --print_pos_ber: the second pass dies in its bare `except` on the first batch (NameError on
            # res_pos_arg) - AFTER one full model(X_test, fwd_noise) forward.  The forward's results are thrown away,
            # but with --precompute_norm_stats its encoder call has already folded that batch's mean / std into the running
            # statistics (encoders.py:110-114), which every later SNR point then normalises with: mirror that one encoder call
            if precomp and num_test_batch > 0:
                first = ((snr_points + 2 + si) * num_test_batch) * batch_size + lo
                stats = torch.zeros(3, dtype=torch.float64, device=dev)
                if nloc > 0:
                    u, _ = model.generate_inputs(nloc, snr, seed=seed, first_block=first)
                    _, stats = model.encode_prenorm(u)
                all_reduce_sum_(stats)
                model.update_precomp(stats)
            say("no pos BER specified.")
        say("Test SNR", snr, "with ber ", float(test_ber), "with bler", float(test_bler))
        say("Punctured Test SNR", snr, "with ber ", float(ber_p), "with bler", float(bler_p))     # trainer.py:220-225 (0.0 / 0.0 without the ranking)
        ber_punc_res.append(float(ber_p))
        bler_punc_res.append(float(bler_p))
        ber_res.append(float(test_ber))
        bler_res.append(float(test_bler))
        bit_res.append(int(sum(c[0] for c in pb)))
        blk_res.append(int(sum(c[1] for c in pb)))

    def new_counts():
        # per-batch (bit errors, block errors) stay on the device; one host read per SNR point
        return torch.zeros((max(num_test_batch, 1), 2), dtype=torch.int64, device=dev)

    if hip_graph and not want_pos and snr_points > 0 and num_test_batch > 0:
        # One hipGraph per SNR point, PIPELINED: while the device runs point i, the host captures and instantiates point i + 2 and has
        # already queued point i + 1, so the stream never drains between points (captured serially, each point cost ~2 ms of idle
        # device: capture + instantiate + two synchronisations).  The graphs share one memory pool - they are launched in capture
        # order on one stream, never concurrently - and the counts of a finished point are fetched on a side stream behind an event.
        main, side, fetch = torch.cuda.current_stream(dev), torch.cuda.Stream(dev), torch.cuda.Stream(dev)
        pool = torch.cuda.graph_pool_handle()
        graphs, counts, done = {}, {}, {}

        def capture(si):
            counts[si] = new_counts()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.stream(side):
                g.capture_begin(pool=pool)
                try:
                    run_point(si, snrs[si], counts[si])
                finally:
                    g.capture_end()
            graphs[si] = g

        def launch(si):
            graphs[si].replay()
            done[si] = torch.cuda.Event()
            done[si].record(main)

        capture(0)
        launch(0)
        if snr_points > 1:
            capture(1)
        for si, snr in enumerate(snrs):
            if si + 1 < snr_points:
                launch(si + 1)
            with torch.cuda.stream(fetch):
                fetch.wait_event(done[si])
                pb = counts[si].cpu().tolist()        # synchronises `fetch` only: point si + 1 keeps running
            del graphs[si], counts[si], done[si]
            report_point(si, snr, pb)
            if si + 2 < snr_points:
                capture(si + 2)
        torch.cuda.synchronize(dev)
        # fp16-split kernels: fail loudly if an activation left the fp16 range.  The flag is sticky and reading it synchronises the
        # device, so the pipelined sweep reads it ONCE, here: a failure cannot name the point, and the lines already printed for this
        # sweep are then invalid - the error says so (the serial path checks after every point)
        try:
            model.check_range()
        except Exception as e:
            raise type(e)(f"{e} [pipelined hip_graph sweep: the sticky range flag was read after all {snr_points} SNR points "
                          f"({snrs[0]:g} .. {snrs[-1]:g} dB); every BER / BLER line printed for this sweep is invalid - rerun with "
                          "hip_graph=False to find the first point that overflows]") from e
    else:
        for si, snr in enumerate(snrs):
            per_batch = new_counts()
            pos_err.zero_()
            pos_pow.zero_()
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
            report_point(si, snr, per_batch.cpu().tolist())
            model.check_range()          # fp16-split kernels: fail loudly if an activation left the fp16 range
    say("final results on SNRs ", snrs)
    say("BER", ber_res)
    say("BLER", bler_res)
    out = {"snrs": snrs, "ber": ber_res, "bler": bler_res, "bit_errors": bit_res, "block_errors": blk_res}
    say("final results on punctured SNRs ", snrs)
    say("BER", ber_punc_res)
    say("BLER", bler_punc_res)
    if print_pos_ber:
        out["pos_ber"] = pos_ber_res
        out["ber_punc"], out["bler_punc"] = ber_punc_res, bler_punc_res
    if print_pos_power:
        out["pos_power"] = pos_power_res
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
            if precomp:
                model.update_precomp(stats)
            loc = torch.zeros(3, dtype=torch.float64, device=dev)
            if nloc > 0:
                codes, _ = model.normalize(x_tx, stats)
                c = codes.double()
                loc = torch.stack([c.sum(), (c * c).sum(), torch.tensor(float(c.numel()), dtype=torch.float64, device=dev)])
            all_reduce_sum_(loc)
            s, ss, n = loc.cpu().tolist()
            enc_power += math.sqrt(max((ss - s * s / n) / (n - 1.0), 0.0))
        enc_power /= float(num_test_batch)
        say("encoder power is", torch.tensor(enc_power, dtype=torch.float32))      # the reference prints the 0-dim tensor (trainer.py:246)
        adj = [snr_sigma2db(snr_db2sigma(item) / enc_power) for item in snrs]
        say("adjusted SNR should be", adj)
        out["enc_power"] = enc_power
        out["adjusted_snrs"] = adj
    return out


def test_from_args(model, args, **overrides) -> Dict[str, List[float]]:
    """`test(model, args)` of the reference (trainer.py:135) with the sweep read from a reference-style namespace:
    snr_test_start / snr_test_end / snr_points / num_block / batch_size / test_ratio / print_pos_ber / print_pos_power /
    num_ber_puncture (get_args.py:103-120,212-215); keyword overrides win (seed, decode_group, hip_graph, verbose ...)."""
    kw = dict(snr_test_start=getattr(args, "snr_test_start", -1.5), snr_test_end=getattr(args, "snr_test_end", 4.0),
              snr_points=getattr(args, "snr_points", 12), num_block=getattr(args, "num_block", 1000),
              batch_size=getattr(args, "batch_size", 100), test_ratio=getattr(args, "test_ratio", 1),
              print_pos_ber=bool(getattr(args, "print_pos_ber", False)), print_pos_power=bool(getattr(args, "print_pos_power", False)),
              num_ber_puncture=getattr(args, "num_ber_puncture", 5))
    kw.update(overrides)
    return test(model, **kw)


test.__test__ = False            # not pytest tests (the names follow the reference)
test_from_args.__test__ = False
