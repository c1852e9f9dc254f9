"""Configuration of the TurboAE rate-1/3 CNN hot path.

Field names follow the reference's flat argparse namespace (get_args.py:73-122) so a
reference ``args`` object can be converted with :meth:`TurboAEConfig.from_args`.
Only the flags the hot path reads ("config / flags") are kept.
"""
from __future__ import annotations

from dataclasses import dataclass, asdict


@dataclass(frozen=True)
class TurboAEConfig:
    block_len: int = 100          # get_args.py:122
    code_rate_k: int = 1          # get_args.py:73
    code_rate_n: int = 3          # get_args.py:74
    enc_num_layer: int = 2        # get_args.py:93
    enc_num_unit: int = 100       # get_args.py:98
    enc_kernel_size: int = 5      # get_args.py:89
    dec_num_layer: int = 5        # get_args.py:94
    dec_num_unit: int = 100       # get_args.py:97
    dec_kernel_size: int = 5      # get_args.py:90
    num_iteration: int = 6        # get_args.py:82
    num_iter_ft: int = 5          # get_args.py:84
    extrinsic: int = 1            # get_args.py:83
    enc_act: str = "elu"          # get_args.py:100 ("only elu works"): tanh / selu / relu / elu / sigmoid / linear (encoders.py:86-100)
    dec_act: str = "linear"       # get_args.py:101: the same choices on the GRU decoder's Linear outputs (decoders.py:59-73,103-143);
                                  # DEC_LargeCNN has no dec_act
    encoder: str = "TurboAE_rate3_cnn"   # main.py:32-36: 'TurboAE_rate3_cnn' (ENC_interCNN) or 'TurboAE_rate3_rnn' (ENC_interRNN,
                                         # encoders.py:231-298: three 2-layer bidirectional GRU(1 -> U) + Linear(2U -> 1); the
                                         # reference's own CLI cannot select it - its -encoder choice is spelled 'Turboae_rate3_rnn')
    decoder: str = "TurboAE_rate3_cnn"   # get_args.py:26 / main.py:75-76,87-88: 'TurboAE_rate3_cnn' (DEC_LargeCNN) or
                                         # 'TurboAE_rate3_rnn' (DEC_LargeRNN, 2-layer bidirectional GRU, dec_rnn='gru')
    enc_rnn: str = "gru"          # get_args.py:79: cell of ENC_interRNN (encoders.py:242-247): 'gru' | 'lstm' | 'rnn'
    dec_rnn: str = "gru"          # get_args.py:80: cell of DEC_LargeRNN (decoders.py:27-32)
    interleaver_seed: int = 0     # channel_ae.py:33 (RandInterlv(block_len, 0))
    precision: str = "auto"       # no reference counterpart: 'auto' = fp16-split MFMA contraction (fp32-grade)
                                  # where the whole-block kernels apply; 'f32' = fp32 MFMA everywhere
    range_calibration: bool = True   # no reference counterpart (the reference's fp32 conv, cnn_utils.py:36-46, has no range): per-layer
                                     # power-of-two exponents for the fp16-split panels, measured at engine creation (tae_config.range_calibration)
    range_fallback: bool = False     # tae_config.range_fallback: re-run a call that left the fp16 window on the fp32 kernels (synchronises per call)
    # ---- encoder-output / channel variants on the same kernels
    channel: str = "awgn"                 # get_args.py:43; channel_ae.py:41-56 ('fading': the caller supplies fading_h, which the
                                          # reference draws inside forward)
    vv: float = 5.0                       # get_args.py:53, t-dist degrees of freedom (noise generation only, channels.py:40-41)
    radar_prob: float = 0.05              # get_args.py:55 (channels.py:43-49)
    radar_power: float = 5.0              # get_args.py:56
    no_code_norm: bool = False            # get_args.py:159; encoders.py:104-105
    precompute_norm_stats: bool = False   # get_args.py:218; encoders.py:110-114 (running mean/std over calls)
    train_channel_mode: str = "block_norm"   # get_args.py:135; 'block_norm_ste' quantises the codes (encoders.py:118-120)
    enc_truncate_limit: float = 0.0       # get_args.py:138; encoders.py:122-123
    enc_value_limit: float = 1.0          # get_args.py:168
    enc_quantize_level: float = 2.0       # get_args.py:167
    rec_quantize: bool = False            # get_args.py:205; channel_ae.py:67-69
    rec_quantize_level: int = 2           # get_args.py:207 (the reference also passes it as the clamp limit)

    @property
    def dense(self) -> bool:
        """DenseSameShapeConv1d stacks (cnn_utils.py:49-82) in encoder AND decoder: the reference keys both on the encoder
        name (encoders.py:312-330, decoders.py:173-176)."""
        return self.encoder == "TurboAE_rate3_cnn_dense"

    @property
    def dec_dense(self) -> bool:
        """The CNN decoder is built from DenseSameShapeConv1d whenever the encoder is NOT the plain CNN (decoders.py:173-176)."""
        return self.decoder != "TurboAE_rate3_rnn" and self.encoder != "TurboAE_rate3_cnn"

    @property
    def generic(self) -> bool:
        """True when the configuration lies outside the MFMA kernels' envelope and runs on the library's generic fp32 kernels
        (csrc/turboae_generic.hip; mirrors tae::generic_needed): one launch per layer, same results, far slower."""
        ks = (self.enc_kernel_size, self.dec_kernel_size)
        enc_rnn, dec_rnn = self.encoder == "TurboAE_rate3_rnn", self.decoder == "TurboAE_rate3_rnn"
        # CNN stacks up to 124 wide on the MFMA kernels of both arithmetics (32 / 64 / 100 / 124 instantiated), recurrent kernels up to 100
        cnn_max = 124
        enc_max, dec_max = (100 if enc_rnn else cnn_max), (100 if dec_rnn else cnn_max)
        if max(ks) > 9 or self.enc_num_unit > enc_max or self.dec_num_unit > dec_max or self.num_iter_ft > 6:
            return True
        # LSTM / vanilla-RNN encoder cells (2 layers): unit-split f16x2 kernels since r06; fp32 stays generic
        if enc_rnn and self.enc_rnn != "gru" and self.precision == "f32":
            return True
        # LSTM / vanilla-RNN decoder: unit-split f16x2 kernels (csrc/turboae_rnn_u.hip) behind the CNN or the 2-layer GRU encoder; fp32 stays generic
        if dec_rnn and self.dec_rnn != "gru" and self.precision == "f32":
            return True
        if enc_rnn and (self.enc_num_layer != 2 or not dec_rnn):
            return True
        if self.dense and (dec_rnn or self.enc_num_unit not in (32, 64, 100) or self.dec_num_unit not in (32, 64, 100)):
            return True
        return self.precision == "f32" and (self.dense or max(ks) > 5)

    def validate(self) -> None:
        if self.code_rate_k != 1 or self.code_rate_n != 3:
            raise ValueError("only the rate-1/3 code (code_rate_k=1, code_rate_n=3) is on the hot path")
        if self.enc_rnn not in ("gru", "lstm", "rnn") or self.dec_rnn not in ("gru", "lstm", "rnn"):
            raise ValueError("enc_rnn / dec_rnn must be 'gru', 'lstm' or 'rnn' (get_args.py:79-80)")
        if self.generic:
            self._validate_generic()
            return
        ks = (self.enc_kernel_size, self.dec_kernel_size)
        if any(k not in (1, 3, 5, 7, 9) for k in ks):
            raise ValueError("kernel sizes must be 1, 3, 5, 7 or 9 (odd: SameShapeConv1d pads with kernel_size // 2)")
        if max(ks) > 5 and (self.precision not in ("auto", "f16x1") or self.dense):
            raise ValueError("kernel sizes 7 and 9 are built in the fp16-split kernels only (precision='auto', no dense stacks)")
        if self.precision not in ("auto", "f32", "f16x1"):
            raise ValueError("precision must be 'auto', 'f32' or 'f16x1'")
        if self.precision == "f16x1" and (self.decoder != "TurboAE_rate3_cnn" or self.dense or self.dec_kernel_size > 5
                                          or not 65 <= self.dec_num_unit <= 100 or self.block_len > 320):
            raise ValueError("precision='f16x1' (one fp16 product per slab: NOT fp32-grade, no parity claim) exists for the 100-wide whole-block "
                             "CNN decoder only: decoder='TurboAE_rate3_cnn', 65 <= dec_num_unit <= 100, dec_kernel_size <= 5, block_len <= 320")
        acts = ("tanh", "selu", "relu", "elu", "sigmoid", "linear")
        if self.enc_act not in acts or self.dec_act not in acts:
            raise ValueError("enc_act / dec_act must be one of tanh, selu, relu, elu, sigmoid, linear (get_args.py:100-101)")
        if not (1 <= self.enc_num_unit <= 124 and 1 <= self.dec_num_unit <= 124):
            raise ValueError("channel widths (enc_num_unit, dec_num_unit) must be in 1..124 (kernels exist for 32 / 64 / 100 / 124; a "
                             "narrower stack is embedded exactly into the next wider one)")
        if self.dense and (self.enc_num_unit not in (32, 64, 100) or self.dec_num_unit not in (32, 64, 100)):
            raise ValueError("dense stacks need channel widths of 32, 64 or 100")
        if not (1 <= self.num_iter_ft <= 6):
            raise ValueError("num_iter_ft must be in 1..6 (7-channel decoder input is padded to 8)")
        if self.num_iteration < 1 or self.enc_num_layer < 1 or self.dec_num_layer < 1:
            raise ValueError("layer / iteration counts must be >= 1")
        if self.block_len < 1:
            raise ValueError("block_len must be >= 1")
        if self.channel not in ("awgn", "t-dist", "radar", "ge_awgn", "bec", "bsc", "ge", "fading"):
            raise ValueError("channel must be one of awgn, t-dist, radar, ge_awgn, bec, bsc, ge, fading")
        if self.encoder not in ("TurboAE_rate3_cnn", "TurboAE_rate3_cnn_dense", "TurboAE_rate3_rnn"):
            raise ValueError("encoder must be 'TurboAE_rate3_cnn', 'TurboAE_rate3_cnn_dense' or 'TurboAE_rate3_rnn'")
        if self.dense:
            if self.decoder not in ("TurboAE_rate3_cnn", "TurboAE_rate3_cnn_dense"):
                raise ValueError("the dense encoder pairs with the (then also dense) CNN decoder")
            if self.precision != "auto":
                raise ValueError("DenseSameShapeConv1d is built on the fp16-split long-block kernels only (precision='auto')")
        if self.encoder == "TurboAE_rate3_rnn" and (self.enc_num_layer != 2 or self.decoder != "TurboAE_rate3_rnn"):
            raise ValueError("the GRU encoder runs on the GRU decoder's kernels: enc_num_layer = 2, decoder = 'TurboAE_rate3_rnn' "
                             "(with any other decoder the reference switches the decoder to DenseSameShapeConv1d, decoders.py:173-176)")
        if self.decoder not in ("TurboAE_rate3_cnn", "TurboAE_rate3_cnn_dense", "TurboAE_rate3_rnn"):
            raise ValueError("decoder must be 'TurboAE_rate3_cnn', 'TurboAE_rate3_cnn_dense' or 'TurboAE_rate3_rnn'")
        if self.decoder == "TurboAE_rate3_cnn_dense" and not self.dense:
            raise ValueError("the reference builds DenseSameShapeConv1d decoders from the ENCODER name (decoders.py:173-176): "
                             "use encoder='TurboAE_rate3_cnn_dense'")

    def _validate_generic(self) -> None:
        """Limits of the generic fp32 kernels (tae::generic_check)."""
        for k in (self.enc_kernel_size, self.dec_kernel_size):
            if k < 1 or k > 63 or k % 2 == 0:
                raise ValueError("kernel sizes must be odd and in 1..63 (SameShapeConv1d pads with kernel_size // 2: an even size changes the length)")
        if not (1 <= self.enc_num_unit <= 1024 and 1 <= self.dec_num_unit <= 1024):
            raise ValueError("enc_num_unit / dec_num_unit must be in 1..1024")
        if not (1 <= self.num_iter_ft <= 64):
            raise ValueError("num_iter_ft must be in 1..64")
        if self.precision not in ("auto", "f32"):
            raise ValueError("precision must be 'auto' or 'f32'")
        acts = ("tanh", "selu", "relu", "elu", "sigmoid", "linear")
        if self.enc_act not in acts or self.dec_act not in acts:
            raise ValueError("enc_act / dec_act must be one of tanh, selu, relu, elu, sigmoid, linear (get_args.py:100-101)")
        if self.num_iteration < 1 or self.enc_num_layer < 1 or self.dec_num_layer < 1 or self.block_len < 1:
            raise ValueError("layer / iteration counts and block_len must be >= 1")
        if self.channel not in ("awgn", "t-dist", "radar", "ge_awgn", "bec", "bsc", "ge", "fading"):
            raise ValueError("channel must be one of awgn, t-dist, radar, ge_awgn, bec, bsc, ge, fading")
        if self.encoder not in ("TurboAE_rate3_cnn", "TurboAE_rate3_cnn_dense", "TurboAE_rate3_rnn"):
            raise ValueError("encoder must be 'TurboAE_rate3_cnn', 'TurboAE_rate3_cnn_dense' or 'TurboAE_rate3_rnn'")
        if self.decoder not in ("TurboAE_rate3_cnn", "TurboAE_rate3_cnn_dense", "TurboAE_rate3_rnn"):
            raise ValueError("decoder must be 'TurboAE_rate3_cnn', 'TurboAE_rate3_cnn_dense' or 'TurboAE_rate3_rnn'")
        if self.decoder == "TurboAE_rate3_cnn_dense" and not self.dec_dense:
            raise ValueError("the reference builds DenseSameShapeConv1d decoders from the ENCODER name (decoders.py:173-176): "
                             "use a non-plain-CNN encoder")

    @staticmethod
    def from_args(args) -> "TurboAEConfig":
        """Build from a reference-style argparse namespace (get_args.py)."""
        names = TurboAEConfig.__dataclass_fields__.keys()
        kw = {n: getattr(args, n) for n in names if hasattr(args, n)}
        cfg = TurboAEConfig(**kw)
        return cfg

    def to_dict(self) -> dict:
        return asdict(self)

    # ---- algorithmic work per information bit (SURVEY.md section 8d / BASELINE.md section 3)
    def macs_per_bit(self) -> dict:
        ke, kd = self.enc_kernel_size, self.dec_kernel_size
        ue, ud, f = self.enc_num_unit, self.dec_num_unit, self.num_iter_ft
        gates = {"gru": 3, "lstm": 4, "rnn": 1}
        enc = 3 * (1 * ke * ue + (self.enc_num_layer - 1) * ue * ke * ue + ue)
        if self.encoder == "TurboAE_rate3_rnn":
            g = gates[self.enc_rnn]
            enc = 3 * (2 * sum(g * ue * ((1 if l == 0 else 2 * ue) + ue) for l in range(self.enc_num_layer)) + 2 * ue)
        if self.decoder == "TurboAE_rate3_rnn":
            # 2-layer bidirectional GRU(2+F -> ud): per direction 3*ud*(in + ud) MAC per layer (244 200)
            g = gates[self.dec_rnn]
            stack = 2 * (g * ud * ((2 + f) + ud) + g * ud * (2 * ud + ud))
            dec = 2 * self.num_iteration * stack + (2 * self.num_iteration - 1) * 2 * ud * f + 2 * ud
        else:
            stack = (2 + f) * kd * ud + (self.dec_num_layer - 1) * ud * kd * ud
            if self.dec_dense:      # layer l convolves 2 + F + l * ud input channels (cnn_utils.py:59-62)
                stack = sum((2 + f + l * ud) * kd * ud for l in range(self.dec_num_layer))
            dec = 2 * self.num_iteration * stack + (2 * self.num_iteration - 1) * ud * f + ud
        if self.dense:
            enc = 3 * (sum((1 + l * ue) * ke * ue for l in range(self.enc_num_layer)) + ue)
        return {"enc": enc, "dec": dec, "total": enc + dec}

    def flops_per_bit(self) -> int:
        return 2 * self.macs_per_bit()["total"]
