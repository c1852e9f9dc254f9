"""The oracle holds no state: a stage function gives the same answer whatever ran before it (a module-global
dense switch left behind by a dense forward once sent a plain network through the dense stack and took the GPU suite down).

A dense Channel_AE forward runs first in this process; each stage function is then called on a plain (and on a GRU) network and
compared bit for bit with the same calls made by a FRESH interpreter that has never seen a dense network."""
import os
import subprocess
import sys
import types

import numpy as np
import torch

from turboae_amd import TurboAEConfig, philox, weights as W
from oracle import turboae_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PLAIN = dict(block_len=40, enc_num_unit=32, dec_num_unit=32, enc_num_layer=2, dec_num_layer=3, num_iteration=2, num_iter_ft=3)
DENSE = dict(block_len=24, enc_num_unit=32, dec_num_unit=32, enc_num_layer=2, dec_num_layer=2, num_iteration=2, num_iter_ft=3,
             encoder="TurboAE_rate3_cnn_dense", decoder="TurboAE_rate3_cnn_dense")


def _inputs(cfg, B, seed):
    L = cfg.block_len
    sd = W.generate_state_dict(cfg, seed=seed, gain=1.0)
    u = philox.random_bits(seed, 0, B * L).reshape(B, L, 1)
    noise = (np.float32(0.8) * philox.random_normal(seed, 0, B * L * 3)).reshape(B, L, 3).astype(np.float32)
    return O.to_torch(sd), torch.from_numpy(u), torch.from_numpy(noise)


def stage_outputs():
    """every stage function of the oracle on a plain network (also what the fresh interpreter runs: see __main__)"""
    torch.set_num_threads(1)             # one summation order in both processes
    cfg = TurboAEConfig(**PLAIN)
    w, u, noise = _inputs(cfg, 3, 4711)
    p = torch.from_numpy(O.rand_interleaver(cfg.block_len, 0))
    with torch.no_grad():
        h = O.same_shape_conv1d(2.0 * u - 1.0, w, "enc.enc_cnn_1", cfg.enc_num_layer)
        x_tx = O.encode_prenorm(u, w, p, cfg.enc_num_layer)
        codes = O.encode(u, w, p, cfg.enc_num_layer)
        x_dec = O.decode(codes + noise, w, p, cfg.dec_num_layer, cfg.num_iteration, cfg.num_iter_ft)
        x_fwd, c_fwd = O.channel_ae_forward(u, noise, w, cfg.to_dict())
    return {"stack": h.numpy(), "x_tx": x_tx.numpy(), "codes": codes.numpy(), "x_dec": x_dec.numpy(), "x_fwd": x_fwd.numpy(),
            "c_fwd": c_fwd.numpy()}


def test_stage_functions_do_not_depend_on_what_ran_before(tmp_path):
    # 1. a fresh interpreter that never saw a dense network
    out = str(tmp_path / "fresh.npz")
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    subprocess.run([sys.executable, os.path.abspath(__file__), out], check=True, env=env, cwd=ROOT, timeout=300)
    fresh = np.load(out)
    # 2. this process: a dense forward first, then the same stage calls
    dcfg = TurboAEConfig(**DENSE)
    w, u, noise = _inputs(dcfg, 2, 99)
    xd, cd = O.channel_ae_forward(u, noise, w, dcfg.to_dict())
    assert np.isfinite(xd.numpy()).all()
    here = stage_outputs()
    for k in here:
        assert np.array_equal(here[k], fresh[k]), k
    # the fused forward is the composition of its stages
    assert np.array_equal(here["x_fwd"], here["x_dec"]) and np.array_equal(here["c_fwd"], here["codes"])
    # and the dense forward itself is unchanged by the plain calls in between
    xd2, cd2 = O.channel_ae_forward(u, noise, w, dcfg.to_dict())
    assert torch.equal(xd, xd2) and torch.equal(cd, cd2)


def test_dense_stage_functions_need_the_flag_spelled_out():
    """a dense network through the stage functions: explicit dense=True equals the fused forward; dense=False must not run at all
    on dense weights (channel counts differ), i.e. nothing silently picks a stack"""
    torch.set_num_threads(1)
    dcfg = TurboAEConfig(**DENSE)
    w, u, noise = _inputs(dcfg, 2, 99)
    p = torch.from_numpy(O.rand_interleaver(dcfg.block_len, 0))
    assert O.is_dense(dcfg) and O.is_dense(dcfg.to_dict()) and not O.is_dense(TurboAEConfig(**PLAIN))
    with torch.no_grad():
        x_fwd, c_fwd = O.channel_ae_forward(u, noise, w, dcfg.to_dict())
        codes = O.encode(u, w, p, dcfg.enc_num_layer, dense=True)
        x_dec = O.decode(codes + noise, w, p, dcfg.dec_num_layer, dcfg.num_iteration, dcfg.num_iter_ft, dense=True)
        assert torch.equal(codes, c_fwd) and torch.equal(x_dec, x_fwd)
        try:
            O.encode(u, w, p, dcfg.enc_num_layer)
        except RuntimeError:
            pass
        else:
            raise AssertionError("dense weights went through the plain stack")


def test_oracle_module_has_no_mutable_globals():
    for name, v in vars(O).items():
        if name.startswith("__") or isinstance(v, (types.ModuleType, types.FunctionType, type)):
            continue
        if getattr(v, "__module__", "") == "typing":
            continue
        assert not isinstance(v, (dict, list, set)), f"oracle.{name} is a mutable module global"


if __name__ == "__main__":
    np.savez(sys.argv[1], **stage_outputs())
