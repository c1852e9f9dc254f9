"""Host-side mirror of the reference's ``Channel_AE`` for the MI355X HIP path.

``Channel_AE_HIP(args_or_cfg, state_dict)`` exposes the same surface the reference's eval loop uses
(trainer.py:135-248, main.py:146-172):

    model.eval(); model.to(device)
    x_dec, codes = model(X, fwd_noise)        # Channel_AE.forward, channel_ae.py:20-73
    codes = model.enc(X)                      # ENC_interCNN.forward, encoders.py:351-377
    x_dec = model.dec(received)               # DEC_LargeCNN.forward, decoders.py:206-269
    model.enc.set_interleaver(p); model.dec.set_interleaver(p); model.enc.set_parallel()
    model.load_state_dict(sd); model.state_dict()

All arithmetic runs in libturboae_hip.so through the C ABI (include/turboae_hip.h); torch is used
only for device memory and streams.  There is no CPU fallback: tensors must live on a ROCm device.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional, Tuple

import numpy as np
import torch

from . import _lib
from .config import TurboAEConfig
from .interleaver import rand_interleaver
from . import weights as W

_RNN = {"gru": 0, "lstm": 1, "rnn": 2}                                                # TAE_RNN_*
_ACT = {"elu": 0, "linear": 1, "tanh": 2, "relu": 3, "selu": 4, "sigmoid": 5}      # TAE_ACT_* (include/turboae_hip.h)


def _as_cfg(args_or_cfg) -> TurboAEConfig:
    if isinstance(args_or_cfg, TurboAEConfig):
        return args_or_cfg
    if isinstance(args_or_cfg, dict):
        return TurboAEConfig(**args_or_cfg)
    return TurboAEConfig.from_args(args_or_cfg)


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


class _Engine:
    """Owns the tae_handle (weights + workspace on one GPU)."""

    def __init__(self, cfg: TurboAEConfig, state_dict: Dict[str, object], device: torch.device, max_batch: int):
        cfg.validate()
        if device.type != "cuda":
            raise _lib.TurboAEError("Channel_AE_HIP needs a ROCm GPU device (no CPU fallback)")
        self.cfg = cfg
        if device.index is None:      # 'cuda' names the current device; tensors report cuda:<index>, so comparisons need the index too
            device = torch.device("cuda", torch.cuda.current_device())
        self.device = device
        self.lib = _lib.load()
        self._state = W.check_state_dict(cfg, state_dict)
        blob = W.pack_blob(cfg, self._state)
        c = _lib.TaeConfig(C.sizeof(_lib.TaeConfig), cfg.block_len, cfg.enc_num_layer, cfg.enc_num_unit,
                           cfg.enc_kernel_size, cfg.dec_num_layer, cfg.dec_num_unit, cfg.dec_kernel_size,
                           cfg.num_iteration, cfg.num_iter_ft, cfg.extrinsic, _ACT[cfg.enc_act], max_batch,
                           1 if cfg.decoder == "TurboAE_rate3_rnn" else 0, 1 if cfg.encoder == "TurboAE_rate3_rnn" else 0,
                           1 if cfg.dense else 0, {"auto": 0, "f32": 1, "f16x1": 2}[cfg.precision], _ACT[cfg.dec_act],
                           _RNN[cfg.enc_rnn], _RNN[cfg.dec_rnn], 0 if cfg.range_calibration else 1, 1 if cfg.range_fallback else 0)
        n = self.lib.tae_num_weights(C.byref(c))
        if n != blob.size:
            raise _lib.TurboAEError(f"weight count mismatch: library wants {n}, blob has {blob.size}")
        h = C.c_void_p()
        with torch.cuda.device(device):
            _lib.check(self.lib.tae_create(C.byref(c), blob.ctypes.data_as(C.c_void_p), blob.size, C.byref(h)))
        self.h = h
        self.cap = max_batch
        self.fell_back = False
        self.p_array: Optional[np.ndarray] = None
        self.set_interleaver(rand_interleaver(cfg.block_len, cfg.interleaver_seed))
        self.reset_precomp()
        self.apply_channel_opts()

    # -- encoder-output / channel variants -------------------------------------------------------
    def reset_precomp(self) -> None:               # ENCBase.reset_precomp, encoders.py:84-87
        self.mean_scalar, self.std_scalar, self.num_test_block = 0.0, 1.0, 0.0

    def apply_channel_opts(self, fixed: Optional[Tuple[float, float]] = None) -> None:
        cfg = self.cfg
        o = _lib.TaeChannelOpts()
        o.struct_size = C.sizeof(_lib.TaeChannelOpts)
        o.norm_mode = 1 if cfg.no_code_norm else (2 if fixed is not None else 0)
        o.mean, o.std = (fixed if fixed is not None else (0.0, 1.0))
        o.ste = 1 if cfg.train_channel_mode == "block_norm_ste" else 0
        o.enc_value_limit, o.enc_quantize_level = cfg.enc_value_limit, cfg.enc_quantize_level
        o.enc_truncate_limit = cfg.enc_truncate_limit
        o.channel = {"bec": 1, "bsc": 2, "ge": 2, "fading": 3}.get(cfg.channel, 0)
        o.rec_quantize = 1 if cfg.rec_quantize else 0
        # channel_ae.py:69 passes rec_quantize_level for BOTH the limit and the level
        o.rec_quantize_limit = float(cfg.rec_quantize_level)
        o.rec_quantize_level = float(cfg.rec_quantize_level)
        _lib.check(self.lib.tae_set_channel_opts(self.h, C.byref(o)))
        _lib.check(self.lib.tae_set_noise_opts(self.h, C.byref(self.noise_opts())))      # the generator tae_eval_snr draws from

    def noise_opts(self) -> "_lib.TaeNoiseOpts":
        """tae_noise_opts of the configured channel (-channel, -vv, -radar_prob, -radar_power; get_args.py:43,53-56)."""
        cfg = self.cfg
        o = _lib.TaeNoiseOpts()
        o.struct_size = C.sizeof(_lib.TaeNoiseOpts)
        o.kind = _lib.NOISE_KIND[cfg.channel]
        o.vv, o.radar_prob, o.radar_power = float(cfg.vv), float(cfg.radar_prob), float(cfg.radar_power)
        o.p_gg, o.p_bb = 0.8, 0.8                                       # channels.py:60-61,86-87
        return o

    def update_precomp(self, stats: torch.Tensor) -> None:
        """--precompute_norm_stats (encoders.py:110-114): running averages of the per-call mean / std."""
        from .distributed import mean_std_from_stats
        this_mean, this_std = mean_std_from_stats(stats)
        f32 = np.float32
        self.num_test_block += 1.0
        n = f32(self.num_test_block)
        self.mean_scalar = float((f32(self.mean_scalar) * (n - f32(1.0)) + f32(this_mean)) / n)
        self.std_scalar = float((f32(self.std_scalar) * (n - f32(1.0)) + f32(this_std)) / n)
        self.apply_channel_opts(fixed=(self.mean_scalar, self.std_scalar))

    def close(self) -> None:
        if getattr(self, "h", None) is not None and self.h:
            self.lib.tae_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_interleaver(self, p_array) -> None:
        p = np.ascontiguousarray(np.asarray(p_array).reshape(-1), dtype=np.int32)
        if self.p_array is not None and p.shape == self.p_array.shape and np.array_equal(p, self.p_array):
            return
        with torch.cuda.device(self.device):
            _lib.check(self.lib.tae_set_interleaver(self.h, p.ctypes.data_as(C.c_void_p), int(p.size)))
        self.p_array = p

    def reserve(self, B: int) -> None:
        if B > self.cap:
            with torch.cuda.device(self.device):
                _lib.check(self.lib.tae_reserve(self.h, int(B)))
            self.cap = B

    def kernel_info(self) -> Tuple[int, int]:
        nb, lds = C.c_int32(), C.c_int32()
        _lib.check(self.lib.tae_kernel_info(self.h, C.byref(nb), C.byref(lds)))
        return nb.value, lds.value

    def kernel_variants(self) -> Tuple[bool, bool]:
        """(encoder, decoder): True where the production launches use the both-expm1-branches head instantiation (tae_kernel_variants)."""
        e, d = C.c_int32(), C.c_int32()
        _lib.check(self.lib.tae_kernel_variants(self.h, C.byref(e), C.byref(d)))
        return bool(e.value), bool(d.value)

    def overrides(self) -> str:
        """Debug knobs in effect in this process ("NAME=value;..."; empty: none) - tae_overrides."""
        buf = C.create_string_buffer(2048)
        self.lib.tae_overrides(self.h, buf, 2048)
        return buf.value.decode()

    def range_word(self) -> Tuple[str, int]:
        """('f16x2' | 'f32', bits): the arithmetic in use and the TAE_RANGE_* bits raised since the last call (_lib.RANGE_HIGH:
        a scaled activation left the fp16 range - results invalid; RANGE_LOW: data far below the calibrated window - results no
        longer fp32-grade; RANGE_FELL_BACK: a flagged call was re-run in fp32 and fp32 serves every call since).  Synchronises the device."""
        prec, ovf = C.c_int32(), C.c_int32()
        with torch.cuda.device(self.device):
            _lib.check(self.lib.tae_range_status(self.h, C.byref(prec), C.byref(ovf)))
        return {0: "f32", 1: "f16x2", 2: "f16x1"}[prec.value], int(ovf.value)

    def range_status(self) -> Tuple[str, bool]:
        """('f16x2' | 'f32', out_of_window): True when a launch since the last call left the window of the fp16-split
        representation on either side and was NOT re-run in fp32."""
        mode, bits = self.range_word()
        self.fell_back = self.fell_back or bool(bits & _lib.RANGE_FELL_BACK)
        return mode, bool(bits & (_lib.RANGE_HIGH | _lib.RANGE_LOW)) and not (bits & _lib.RANGE_FELL_BACK)

    def check_range(self) -> None:
        mode, bits = self.range_word()
        self.fell_back = self.fell_back or bool(bits & _lib.RANGE_FELL_BACK)
        if bits & _lib.RANGE_FELL_BACK:
            return
        if bits & _lib.RANGE_HIGH:
            raise _lib.TurboAEError("an activation exceeded the fp16 range in the fp16-split kernels; results are not trustworthy - "
                                    "calibrate on representative data (calibrate_range), use range_fallback=True or TurboAEConfig(precision='f32')")
        if bits & _lib.RANGE_LOW:
            raise _lib.TurboAEError("activations fell far below the calibrated window of the fp16-split kernels; results are finite but not "
                                    "fp32-grade - calibrate on representative data (calibrate_range), use range_fallback=True or precision='f32'")

    def calibrate_range(self, u: Optional[torch.Tensor] = None, noise: Optional[torch.Tensor] = None) -> None:
        """tae_calibrate_range: re-measure the per-layer exponents on the caller's batch (or, both None, the synthetic one).
        `u` is a validated (B, L, 1) tensor on the device and `noise` the tensor the library expects for the configured channel
        (Channel_AE_HIP._channel_input: (B, L, 3), or [fading | noise] for channel='fading') - Channel_AE_HIP.calibrate_range
        builds both from what forward() takes."""
        if (u is None) != (noise is None):
            raise ValueError("calibrate_range needs input and fwd_noise together (or neither: the library's synthetic batch)")
        with torch.cuda.device(self.device):
            if u is None:
                _lib.check(self.lib.tae_calibrate_range(self.h, None, None, 0))
            else:
                B = int(u.shape[0])
                want = B * self.cfg.block_len * 3 * (2 if self.cfg.channel == "fading" else 1)
                if noise.numel() != want or noise.device != self.device or noise.dtype != torch.float32 or not noise.is_contiguous():
                    raise ValueError(f"calibration noise must be {want} contiguous float32 values on {self.device}, got {noise.numel()}")
                self.reserve(B)
                torch.cuda.synchronize(self.device)
                _lib.check(self.lib.tae_calibrate_range(self.h, _ptr(u), _ptr(noise), B))

    def range_info(self):
        """(encoder exponents, decoder exponents, passes): per side one exponent per stack input, then one per (stack, layer)."""
        ne, nd, ps = C.c_int32(), C.c_int32(), C.c_int32()
        _lib.check(self.lib.tae_range_info(self.h, C.byref(ne), C.byref(nd), None, 0, C.byref(ps)))
        buf = (C.c_int32 * max(1, ne.value + nd.value))()
        _lib.check(self.lib.tae_range_info(self.h, C.byref(ne), C.byref(nd), buf, ne.value + nd.value, C.byref(ps)))
        v = list(buf)
        return v[:ne.value], v[ne.value:ne.value + nd.value], ps.value

    # ---- tensor helpers
    def _in(self, t: torch.Tensor, last: int, name: str) -> torch.Tensor:
        L = self.cfg.block_len
        if t.dim() != 3 or t.shape[1] != L or t.shape[2] != last:
            raise ValueError(f"{name} must have shape (B, {L}, {last}), got {tuple(t.shape)}")
        if t.device != self.device:
            t = t.to(self.device)
        return t.contiguous().float()

    def _out(self, B: int, last: int) -> torch.Tensor:
        return torch.empty((B, self.cfg.block_len, last), dtype=torch.float32, device=self.device)


class _EncView:
    """model.enc: callable like ENC_interCNN (encoders.py:306-377)."""

    def __init__(self, eng: _Engine):
        self._e = eng

    def set_interleaver(self, p_array) -> None:        # encoders.py:340-341
        self._e.set_interleaver(p_array)

    def set_parallel(self) -> None:                    # encoders.py:343-349: DataParallel wrappers; nothing to do
        pass

    # --precompute_norm_stats state of ENCBase (encoders.py:82-87,110-114); trainer.test prints mean_scalar / std_scalar (trainer.py:153)
    @property
    def mean_scalar(self) -> float:
        return self._e.mean_scalar

    @property
    def std_scalar(self) -> float:
        return self._e.std_scalar

    @property
    def num_test_block(self) -> float:
        return self._e.num_test_block

    def reset_precomp(self) -> None:                   # encoders.py:84-87
        self._e.reset_precomp()
        self._e.apply_channel_opts()

    def __call__(self, inputs: torch.Tensor) -> torch.Tensor:
        e = self._e
        u = e._in(inputs, 1, "inputs")
        B = u.shape[0]
        e.reserve(B)
        codes = e._out(B, 3)
        with torch.cuda.device(e.device):
            if e.cfg.precompute_norm_stats and not e.cfg.no_code_norm:
                x_tx = e._out(B, 3)
                stats = torch.empty(3, dtype=torch.float64, device=e.device)
                _lib.check(e.lib.tae_encode_prenorm(e.h, _ptr(u), _ptr(x_tx), _ptr(stats), B, _stream()))
                e.update_precomp(stats)
                _lib.check(e.lib.tae_normalize(e.h, _ptr(x_tx), _ptr(stats), None, _ptr(codes), None, B, _stream()))
            else:
                _lib.check(e.lib.tae_encode(e.h, _ptr(u), _ptr(codes), B, _stream()))
        return codes

    forward = __call__


class _DecView:
    """model.dec: callable like DEC_LargeCNN (decoders.py:157-269)."""

    def __init__(self, eng: _Engine):
        self._e = eng

    def set_interleaver(self, p_array) -> None:        # decoders.py:202-204
        self._e.set_interleaver(p_array)

    def set_parallel(self) -> None:                    # decoders.py:194-199
        pass

    def __call__(self, received: torch.Tensor) -> torch.Tensor:
        e = self._e
        rx = e._in(received, 3, "received")
        B = rx.shape[0]
        e.reserve(B)
        x_dec = e._out(B, 1)
        with torch.cuda.device(e.device):
            _lib.check(e.lib.tae_decode(e.h, _ptr(rx), _ptr(x_dec), B, _stream()))
        return x_dec

    forward = __call__


class Channel_AE_HIP:
    """Drop-in for ``Channel_AE(args, enc, dec)`` on the AWGN / rate-1/3 CNN eval path."""

    def __init__(self, args_or_cfg, state_dict: Dict[str, object], device: Optional[torch.device] = None,
                 max_batch: int = 500, is_same_interleaver: Optional[int] = None, is_variable_block_len: Optional[bool] = None,
                 is_interleave: Optional[int] = None, range_fallback: bool = False):
        cfg = _as_cfg(args_or_cfg)
        # -is_interleave / -is_same_interleaver (get_args.py:85,87), taken from a reference-style namespace when not given:
        #   is_interleave == 0: the identity permutation set at construction (main.py:129-131) and forward leaves whatever
        #       enc/dec.set_interleaver installed alone (channel_ae.py:22-23);
        #   is_same_interleaver == 1 (default): RandInterlv(block_len, 0) on every forward (channel_ae.py:32-36);
        #   is_same_interleaver == 0: a fresh RandInterlv(block_len, np.random.randint(0, 1000)) per forward (:25-30).
        if is_interleave is None:
            is_interleave = int(getattr(args_or_cfg, "is_interleave", 1))
        if is_same_interleaver is None:
            is_same_interleaver = int(getattr(args_or_cfg, "is_same_interleaver", 1))
        self.is_interleave = is_interleave
        # --is_variable_block_len (get_args.py:125; encoders.py:353-360, decoders.py:208-215): the fully convolutional
        # model runs on any block length with the seed-0 permutation of that length; one engine per length
        if is_variable_block_len is None:
            is_variable_block_len = bool(getattr(args_or_cfg, "is_variable_block_len", False))
        self.is_variable_block_len = is_variable_block_len
        self._by_len: Dict[int, _Engine] = {}
        # range_fallback (= tae_config.range_fallback): the library waits for every call, reads the fp16-split kernels' range word and,
        # if a launch left the window, runs that call again on an fp32 twin of the engine - which serves every later call; `fell_back`
        # says so.  The retry happens inside the library call, so the mirror's own side effects (running statistics of
        # --precompute_norm_stats, the per-call interleaver draw) happen once.  Off by default: the entry points stay asynchronous and
        # capturable, the caller checks with check_range().
        if range_fallback and not cfg.range_fallback:
            from dataclasses import replace
            cfg = replace(cfg, range_fallback=True)
        self.range_fallback = bool(cfg.range_fallback)
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
        self.cfg = cfg
        self.is_same_interleaver = is_same_interleaver
        self._eng = _Engine(cfg, state_dict, torch.device(device), max_batch)
        self.enc = _EncView(self._eng)
        self.dec = _DecView(self._eng)
        self.this_device = self._eng.device
        if self.is_interleave == 0:
            self._eng.set_interleaver(np.arange(cfg.block_len))

    # -- torch.nn.Module look-alikes used by main.py / trainer.py
    def eval(self):
        return self

    def to(self, device):
        if torch.device(device) != self._eng.device and torch.device(device).type != "cuda":
            raise _lib.TurboAEError("Channel_AE_HIP cannot move to a non-GPU device")
        return self

    def state_dict(self) -> Dict[str, torch.Tensor]:
        return {k: torch.from_numpy(v.copy()) for k, v in self._eng._state.items()}

    def load_state_dict(self, state_dict, strict: bool = True):
        old = self._eng
        self._eng = _Engine(self.cfg, state_dict, old.device, old.cap)
        self._eng.set_interleaver(old.p_array)
        self._eng.mean_scalar, self._eng.std_scalar, self._eng.num_test_block = old.mean_scalar, old.std_scalar, old.num_test_block
        self.enc._e = self._eng
        self.dec._e = self._eng
        old.close()
        for e in self._by_len.values():        # per-length engines (is_variable_block_len) hold the OLD weights: rebuild on demand
            e.close()
        self._by_len.clear()
        return self

    def kernel_info(self):
        return self._eng.kernel_info()

    def kernel_variants(self):
        return self._eng.kernel_variants()

    def overrides(self) -> str:
        return self._eng.overrides()

    def reserve(self, max_batch: int) -> None:
        """Grow the library's workspace to `max_batch` blocks per call now (tae_reserve) instead of on first use - needed
        before a hipGraph capture, which must not allocate."""
        self._eng.reserve(max_batch)

    def range_status(self):
        return self._eng.range_status()

    def check_range(self) -> None:
        """Raises TurboAEError if the fp16-split kernels saw an out-of-range activation (call after a batch of forwards)."""
        self._eng.check_range()

    def _engine_for(self, L: int) -> _Engine:
        if L == self.cfg.block_len:
            return self._eng
        if not self.is_variable_block_len:
            raise ValueError(f"input block length {L} != configured block_len {self.cfg.block_len} "
                             "(pass is_variable_block_len=True to allow other lengths)")
        if L not in self._by_len:
            from dataclasses import replace
            e = _Engine(replace(self.cfg, block_len=L), self._eng._state, self._eng.device, self._eng.cap)
            if self.is_interleave == 0:        # no interleaver (main.py:129-131): forward() never installs one, so do it here
                e.set_interleaver(np.arange(L))
            self._by_len[L] = e
        return self._by_len[L]

    def _channel_input(self, e, fwd_noise: torch.Tensor, fading: Optional[torch.Tensor]) -> torch.Tensor:
        """The tensor handed to the library as `noise`: for channel='fading' the fading coefficients followed by the
        additive noise (include/turboae_hip.h, tae_channel_opts.channel = 3); otherwise the noise itself."""
        if fwd_noise.dim() == 3 and fwd_noise.shape[2] == 1:
            # the reference's punctured test pass hands a (B, L, 1) noise tensor (generate_noise(X_test.shape, ...),
            # trainer.py:198-201) and `codes + fwd_noise` (channel_ae.py:42) broadcasts it over the three code symbols
            fwd_noise = fwd_noise.expand(-1, -1, 3)
        noise = e._in(fwd_noise, 3, "fwd_noise")
        if e.cfg.channel != "fading":
            if fading is not None:
                raise ValueError("fading coefficients given but the configured channel is not 'fading'")
            return noise
        if fading is None:
            raise ValueError("channel='fading' needs the fading coefficients (the reference draws them inside forward, "
                             "channel_ae.py:51-56; Channel_AE_HIP.generate_noise(B, test_sigma, seed) returns (noise, fading) on the device)")
        fh = e._in(fading, 3, "fading")
        if fh.shape != noise.shape:
            raise ValueError("fading and fwd_noise shapes differ")
        return torch.cat([fh.reshape(-1), noise.reshape(-1)])

    @property
    def fell_back(self) -> bool:
        """True once a call of a range_fallback model was re-run on the fp32 kernels (they serve every call since)."""
        engines = [self._eng] + list(self._by_len.values())
        if self.range_fallback and not any(e.fell_back for e in engines):
            for e in engines:
                e.range_status()
        return any(e.fell_back for e in engines)

    def calibrate_range(self, input: Optional[torch.Tensor] = None, fwd_noise: Optional[torch.Tensor] = None,
                        fading: Optional[torch.Tensor] = None) -> None:
        """Re-measure the fp16-split kernels' per-layer exponents on this batch (default: the library's synthetic batch).  The
        arguments are the ones forward() takes and go through the same checks: `input` (B, L, 1), `fwd_noise` (B, L, 3) or the
        punctured pass's (B, L, 1), `fading` for channel='fading'; with is_variable_block_len the engine of input.shape[1] is the
        one calibrated (every engine starts from the synthetic calibration of its own length)."""
        if (input is None) != (fwd_noise is None):
            raise ValueError("calibrate_range needs input and fwd_noise together (or neither: the library's synthetic batch)")
        if input is None:
            if fading is not None:
                raise ValueError("fading coefficients given without input / fwd_noise")
            for e in [self._eng] + list(self._by_len.values()):
                e.calibrate_range()
            return
        e = self._engine_for(input.shape[1]) if input.dim() == 3 else self._eng
        u = e._in(input, 1, "input")
        if fwd_noise.shape[0] != u.shape[0]:
            raise ValueError("input and fwd_noise batch sizes differ")
        e.calibrate_range(u, self._channel_input(e, fwd_noise, fading))

    def forward(self, input: torch.Tensor, fwd_noise: torch.Tensor, fading: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        return self._forward_once(input, fwd_noise, fading)

    def _forward_once(self, input: torch.Tensor, fwd_noise: torch.Tensor, fading: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        e = self._engine_for(input.shape[1]) if input.dim() == 3 else self._eng
        if self.is_interleave == 0:      # channel_ae.py:22-23
            pass
        elif self.is_same_interleaver == 0:   # channel_ae.py:25-30 (the global numpy RNG, like the reference)
            e.set_interleaver(rand_interleaver(e.cfg.block_len, int(np.random.randint(0, 1000))))
        else:                            # channel_ae.py:32-36: RandInterlv(block_len, 0) on every call
            e.set_interleaver(rand_interleaver(e.cfg.block_len, 0))
        u = e._in(input, 1, "input")
        B = u.shape[0]
        if fwd_noise.shape[0] != B:
            raise ValueError("input and fwd_noise batch sizes differ")
        noise = self._channel_input(e, fwd_noise, fading)
        e.reserve(B)
        x_dec, codes = e._out(B, 1), e._out(B, 3)
        with torch.cuda.device(e.device):
            if e.cfg.precompute_norm_stats and not e.cfg.no_code_norm:
                x_tx, rx = e._out(B, 3), e._out(B, 3)
                stats = torch.empty(3, dtype=torch.float64, device=e.device)
                _lib.check(e.lib.tae_encode_prenorm(e.h, _ptr(u), _ptr(x_tx), _ptr(stats), B, _stream()))
                e.update_precomp(stats)
                _lib.check(e.lib.tae_normalize(e.h, _ptr(x_tx), _ptr(stats), _ptr(noise), _ptr(codes), _ptr(rx), B, _stream()))
                _lib.check(e.lib.tae_decode(e.h, _ptr(rx), _ptr(x_dec), B, _stream()))
            else:
                _lib.check(e.lib.tae_forward(e.h, _ptr(u), _ptr(noise), _ptr(x_dec), _ptr(codes), B, _stream()))
        return x_dec, codes

    __call__ = forward

    # -- split form used for multi-GPU sharding
    def encode_prenorm(self, input: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        e = self._eng
        u = e._in(input, 1, "input")
        B = u.shape[0]
        e.reserve(B)
        x_tx = e._out(B, 3)
        stats = torch.empty(3, dtype=torch.float64, device=e.device)
        with torch.cuda.device(e.device):
            _lib.check(e.lib.tae_encode_prenorm(e.h, _ptr(u), _ptr(x_tx), _ptr(stats), B, _stream()))
        return x_tx, stats

    def normalize(self, x_tx: torch.Tensor, stats: torch.Tensor, fwd_noise: Optional[torch.Tensor] = None,
                  want_codes: bool = True, fading: Optional[torch.Tensor] = None):
        e = self._eng
        x = e._in(x_tx, 3, "x_tx")
        B = x.shape[0]
        noise = None if fwd_noise is None else self._channel_input(e, fwd_noise, fading)
        codes = e._out(B, 3) if want_codes else None
        rx = e._out(B, 3) if noise is not None else None
        st = stats.to(device=e.device, dtype=torch.float64).contiguous()
        with torch.cuda.device(e.device):
            _lib.check(e.lib.tae_normalize(e.h, _ptr(x), _ptr(st), _ptr(noise), _ptr(codes), _ptr(rx), B, _stream()))
        return codes, rx

    def decode_taps(self, received: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """model.dec(received) plus what every half-iteration hands to the next one (tae_decode_taps, a debug
        instantiation of the decoder kernel): returns (x_dec, taps) with taps of shape (2 * num_iteration - 1, B, L, num_iter_ft):
        taps[2 * it] = x_plr after dec1 of iteration `it` (natural order, decoders.py:233-236), taps[2 * it + 1] = x_plr after
        dec2 (interleaved order; `prior` of the next iteration is its deinterleave, decoders.py:244-249)."""
        e = self._eng
        rx = e._in(received, 3, "received")
        B = rx.shape[0]
        e.reserve(B)
        x_dec = e._out(B, 1)
        taps = torch.zeros((2 * e.cfg.num_iteration - 1, B, e.cfg.block_len, e.cfg.num_iter_ft), dtype=torch.float32, device=e.device)
        with torch.cuda.device(e.device):
            _lib.check(e.lib.tae_decode_taps(e.h, _ptr(rx), _ptr(x_dec), _ptr(taps), B, _stream()))
        return x_dec, taps

    def count_errors(self, x_dec: torch.Tensor, u: torch.Tensor, counts: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Accumulate (bit errors, block errors) (utils.py:6-18,49-66) into a device int64[2] tensor."""
        e = self._eng
        xd = e._in(x_dec, 1, "x_dec")
        uu = e._in(u, 1, "u")
        if counts is None:
            counts = torch.zeros(2, dtype=torch.int64, device=e.device)
        with torch.cuda.device(e.device):
            _lib.check(e.lib.tae_count_errors(e.h, _ptr(xd), _ptr(uu), xd.shape[0], _ptr(counts), _stream()))
        return counts

    def update_precomp(self, stats: torch.Tensor) -> None:
        """--precompute_norm_stats: fold one call's (all-reduced) statistics into the running mean / std that normalize()
        then uses (ENCBase.power_constraint, encoders.py:110-114).  model(...) and model.enc(...) do this themselves; callers of
        the split form encode_prenorm -> normalize do it in between.  Reads the statistics back (synchronises)."""
        if self._eng.cfg.precompute_norm_stats and not self._eng.cfg.no_code_norm:
            self._eng.update_precomp(stats)

    def eval_snr(self, snr_db: float, batch: int, n_batches: int, seed: int, first_block: int = 0,
                 seed_noise: Optional[int] = None) -> torch.Tensor:
        """One SNR point of trainer.test on the device (tae_eval_snr): int64 (n_batches, 2) tensor of per-batch
        (bit errors, block errors) for the Philox inputs of generate_inputs(batch, snr_db, seed, first_block + i * batch)."""
        e = self._eng
        counts = torch.empty((n_batches, 2), dtype=torch.int64, device=e.device)
        sn = seed if seed_noise is None else seed_noise
        with torch.cuda.device(e.device):
            _lib.check(e.lib.tae_eval_snr(e.h, float(snr_db), int(batch), int(n_batches), int(first_block), int(seed), int(sn),
                                          _ptr(counts), _stream()))
        e.cap = max(e.cap, min(n_batches, -(-24576 // batch)) * batch)      # the library grew its workspace to one decode group
        return counts

    def generate_noise(self, B: int, test_sigma: float, seed: int, first_block: int = 0):
        """Device-side ``generate_noise(noise_shape, args, test_sigma=...)`` of the configured channel (channels.py:27-109) through
        tae_generate_noise: returns (noise, fading) - `fading` is the Rayleigh coefficient tensor the reference draws inside forward
        for -channel fading (channel_ae.py:51-56), None for every other channel.  `test_sigma` is the SNR in dB for the additive
        channels and the erase / flip probability for bec / bsc / ge.  turboae_amd/channels.py is the numpy mirror of the draw."""
        e = self._eng
        n = B * e.cfg.block_len * 3
        is_fading = e.cfg.channel == "fading"
        buf = torch.empty(2 * n if is_fading else n, dtype=torch.float32, device=e.device)
        noise = buf[n:] if is_fading else buf
        with torch.cuda.device(e.device):
            _lib.check(e.lib.tae_generate_noise(e.h, C.byref(e.noise_opts()), float(test_sigma), _ptr(noise), _ptr(buf) if is_fading else None,
                                                B, int(first_block), int(seed), _stream()))
        shape = (B, e.cfg.block_len, 3)
        return noise.view(shape), (buf[:n].view(shape) if is_fading else None)

    def generate_inputs(self, B: int, snr_db: float, seed: int, first_block: int = 0,
                        seed_noise: Optional[int] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        """Device-side test inputs (replaces trainer.py:167-169), Philox streams of turboae_amd/philox.py."""
        e = self._eng
        u, noise = e._out(B, 1), e._out(B, 3)
        sn = seed if seed_noise is None else seed_noise
        with torch.cuda.device(e.device):
            _lib.check(e.lib.tae_generate_inputs(e.h, _ptr(u), _ptr(noise), B, int(first_block), int(seed), int(sn),
                                                 float(snr_db), _stream()))
        return u, noise
