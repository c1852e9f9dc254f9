"""Write the two files examples/c_host/turboae_sweep reads - the canonical float32 weight blob and the interleaver - from
an upstream .pt checkpoint (whole pickled model or state_dict, main.py:162-172), from the short-trained test fixture, or
from the portable generator:

    python tools/export_for_c_host.py <model.pt | fixture | random> <out_prefix> [block_len=100] [enc_num_layer=2]

-> <out_prefix>.f32 (+ .f32.json manifest: configuration and tensor order) and <out_prefix>.perm.i32
"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from turboae_amd import TurboAEConfig, weights as W
from turboae_amd.interleaver import rand_interleaver

src, prefix = sys.argv[1], sys.argv[2]
L = int(sys.argv[3]) if len(sys.argv) > 3 else 100
cfg = TurboAEConfig(block_len=L, enc_num_layer=int(sys.argv[4]) if len(sys.argv) > 4 else 2)
if src == "fixture":
    gold = os.path.join(ROOT, "tests", "golden")
    cfg = TurboAEConfig(**json.load(open(os.path.join(gold, "MANIFEST.json")))["trained"]["config"])
    sd = W.unpack_blob(cfg, np.load(os.path.join(gold, "trained_enc2dec5_u100.npz"))["weights_fp16"].astype(np.float32))
elif src == "random":
    sd = W.generate_state_dict(cfg)
else:
    sd = W.from_torch_checkpoint(src, cfg)          # needs the reference's classes importable only for whole pickled models
W.save_blob(prefix + ".f32", cfg, sd)
rand_interleaver(cfg.block_len, cfg.interleaver_seed).astype("<i4").tofile(prefix + ".perm.i32")
print(f"{prefix}.f32: {W.num_params(cfg)} floats; {prefix}.perm.i32: {cfg.block_len} int32")
