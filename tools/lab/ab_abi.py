"""Decoder / encoder kernel time of ANY build of the library on the bench workload (trained enc2/dec5, 50 000 blocks), talking to it
through ctypes with the tae_config of ITS ABI version - so that an older round's library (e.g. r03 = ABI 9, built from its commit into
tools/lab/probes/libs/) can be timed on the same box as the current one:   python tools/lab/ab_abi.py <lib.so> [B]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from turboae_amd import TurboAEConfig, weights as W, philox
from turboae_amd.interleaver import rand_interleaver
lib = C.CDLL(os.path.abspath(sys.argv[1])); B = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
lib.tae_abi_version.restype = C.c_int
abi = lib.tae_abi_version()
names = ["struct_size", "block_len", "enc_num_layer", "enc_num_unit", "enc_kernel_size", "dec_num_layer", "dec_num_unit", "dec_kernel_size",
         "num_iteration", "num_iter_ft", "extrinsic", "enc_act", "max_batch", "dec_type", "enc_type", "dense", "precision", "dec_act", "enc_rnn", "dec_rnn"]
if abi >= 10: names += ["range_calibration", "range_fallback"]
class Cfg(C.Structure): _fields_ = [(n, C.c_int32) for n in names]
c = Cfg(); c.struct_size = C.sizeof(Cfg); c.block_len = 100; c.enc_num_layer = 2; c.enc_num_unit = 100; c.enc_kernel_size = 5; c.dec_num_layer = 5
c.dec_num_unit = 100; c.dec_kernel_size = 5; c.num_iteration = 6; c.num_iter_ft = 5; c.extrinsic = 1; c.max_batch = B; c.dec_act = 1
cfg = TurboAEConfig()
blob = np.ascontiguousarray(np.load(os.path.join(ROOT, "tests", "golden", "trained_enc2dec5_u100_fp32.npz"))["weights_fp32"], dtype="<f4")
P = C.c_void_p
lib.tae_create.argtypes = [C.POINTER(Cfg), P, C.c_size_t, C.POINTER(P)]; lib.tae_set_interleaver.argtypes = [P, P, C.c_int32]
lib.tae_encode.argtypes = [P, P, P, C.c_int32, P]; lib.tae_decode.argtypes = [P, P, P, C.c_int32, P]; lib.tae_last_error.restype = C.c_char_p
h = P(); torch.cuda.set_device(0)
assert lib.tae_create(C.byref(c), blob.ctypes.data_as(P), blob.size, C.byref(h)) == 0, lib.tae_last_error()
p = rand_interleaver(100, 0).astype(np.int32); assert lib.tae_set_interleaver(h, p.ctypes.data_as(P), 100) == 0
dev = torch.device("cuda", 0)
u = torch.from_numpy(philox.random_bits(1, 0, B * 100).reshape(B, 100, 1)).to(dev)
noise = torch.from_numpy((0.794 * philox.random_normal(1, 0, B * 300)).astype(np.float32).reshape(B, 100, 3)).to(dev)
codes = torch.empty(B, 100, 3, device=dev); xd = torch.empty(B, 100, 1, device=dev)
st = torch.cuda.current_stream().cuda_stream
def timed(fn, n=6):
    for _ in range(2): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return float(np.median(ts))
te = timed(lambda: lib.tae_encode(h, u.data_ptr(), codes.data_ptr(), B, st))
rx = codes + noise
td = timed(lambda: lib.tae_decode(h, rx.data_ptr(), xd.data_ptr(), B, st))
ber = float(((xd > 0.5) != (u > 0.5)).float().mean())
print(f"{os.path.basename(sys.argv[1]):34s} ABI {abi}  decoder {td:.2f} ms  encoder+norm {te:.2f} ms  BER {ber:.5f}", flush=True)
