"""Host-side logic: Philox known answers, weight naming / packing, config accounting."""
import numpy as np
import pytest

from turboae_amd import TurboAEConfig, philox, weights as W


def test_philox_known_answers():
    # Random123 kat_vectors for philox4x32-10
    kat = [
        ((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
        ((0xffffffff,) * 4, (0xffffffff, 0xffffffff), (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
        ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
         (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
    ]
    for ctr, key, want in kat:
        got = philox.philox4x32_10(*[np.array([c], dtype=np.uint32) for c in ctr], key[0], key[1])
        assert tuple(int(g[0]) for g in got) == want


def test_philox_streams_are_sliceable():
    a = philox.random_u32(5, philox.STREAM_BITS, 0, 1000)
    b = philox.random_u32(5, philox.STREAM_BITS, 333, 100)
    assert np.array_equal(a[333:433], b)
    n = philox.random_normal(9, 0, 4000)
    m = philox.random_normal(9, 1001, 50)
    assert np.array_equal(n[1001:1051], m)
    assert abs(float(n.mean())) < 0.06 and abs(float(n.std()) - 1.0) < 0.05
    bits = philox.random_bits(3, 0, 10000)
    assert set(np.unique(bits)) == {0.0, 1.0} and abs(bits.mean() - 0.5) < 0.03


def test_param_count_matches_reference():
    # enc 152 403 + dec 2 453 656 = 2 606 059 params, 162 tensors (enc2/dec5)
    cfg = TurboAEConfig()
    ents = W.canonical_entries(cfg)
    assert len(ents) == 162
    assert W.num_params(cfg) == 2606059
    enc = sum(int(np.prod(s)) for k, s in ents if k.startswith("enc."))
    assert enc == 152403


def test_macs_per_bit_accounting():
    # SURVEY.md section 8d
    cfg = TurboAEConfig()
    m = cfg.macs_per_bit()
    assert m == {"enc": 151800, "dec": 2447600, "total": 2599400}
    assert cfg.flops_per_bit() == 5198800
    assert TurboAEConfig(enc_num_layer=5).flops_per_bit() == 6098800


def test_strip_and_add_module():
    cfg = TurboAEConfig(enc_num_unit=32, dec_num_unit=32)
    sd = W.generate_state_dict(cfg, 1)
    wrapped = W.add_module(sd)
    assert "enc.enc_cnn_1.module.cnns.0.weight" in wrapped
    assert "dec.dec2_outputs.5.module.bias" in wrapped
    assert set(W.strip_module(wrapped)) == set(sd)
    chk = W.check_state_dict(cfg, wrapped)
    assert all(np.array_equal(chk[k], sd[k]) for k in sd)


def test_blob_roundtrip_and_strictness(tmp_path):
    cfg = TurboAEConfig(enc_num_unit=32, dec_num_unit=32, num_iteration=2)
    sd = W.generate_state_dict(cfg, 2)
    blob = W.pack_blob(cfg, sd)
    assert blob.size == W.num_params(cfg)
    back = W.unpack_blob(cfg, blob)
    assert all(np.array_equal(back[k], sd[k]) for k in sd)
    W.save_blob(str(tmp_path / "w.bin"), cfg, sd)
    cfg2, sd2 = W.load_blob(str(tmp_path / "w.bin"))
    assert cfg2 == cfg and all(np.array_equal(sd2[k], sd[k]) for k in sd)
    bad = dict(sd)
    bad.pop("enc.enc_linear_1.bias")
    with pytest.raises(ValueError):
        W.check_state_dict(cfg, bad)
    bad = dict(sd)
    bad["dec.dec1_outputs.0.weight"] = np.zeros((4, 32), np.float32)
    with pytest.raises(ValueError):
        W.check_state_dict(cfg, bad)


def test_generated_weights_are_reproducible():
    cfg = TurboAEConfig(enc_num_unit=32, dec_num_unit=32)
    a, b = W.generate_state_dict(cfg, 5), W.generate_state_dict(cfg, 5)
    assert all(np.array_equal(a[k], b[k]) for k in a)
    c = W.generate_state_dict(cfg, 6)
    assert not np.array_equal(a["enc.enc_cnn_1.cnns.0.weight"], c["enc.enc_cnn_1.cnns.0.weight"])
    w = a["dec.dec1_cnns.0.cnns.1.weight"]
    assert abs(float(w.std()) - 1.0 / np.sqrt(32 * 5)) < 0.01      # variance-preserving scale


def test_config_validation():
    TurboAEConfig().validate()
    TurboAEConfig(enc_kernel_size=3, dec_kernel_size=1, enc_num_unit=32, dec_num_unit=64).validate()
    TurboAEConfig(enc_kernel_size=7, dec_kernel_size=9).validate()
    with pytest.raises(ValueError):
        TurboAEConfig(enc_kernel_size=4).validate()
    TurboAEConfig(enc_num_unit=48, dec_num_unit=7).validate()
    assert not TurboAEConfig(enc_num_unit=48, dec_num_unit=7).generic and not TurboAEConfig(enc_kernel_size=7, dec_kernel_size=9).generic
    # outside the MFMA kernels' envelope: the generic fp32 kernels (r03) - still validated, with their own (wider) limits
    for over in (dict(enc_kernel_size=7, precision="f32"), dict(enc_num_unit=128, dec_num_unit=128), dict(num_iter_ft=9), dict(dec_kernel_size=21),
                 dict(decoder="TurboAE_rate3_rnn", dec_rnn="lstm", precision="f32"), dict(decoder="TurboAE_rate3_rnn", dec_rnn="rnn", dec_num_unit=101),
                 dict(encoder="TurboAE_rate3_rnn", decoder="TurboAE_rate3_rnn", dec_rnn="lstm", precision="f32"),
                 dict(encoder="TurboAE_rate3_rnn", decoder="TurboAE_rate3_rnn", enc_num_layer=3),
                 dict(encoder="TurboAE_rate3_rnn"), dict(encoder="TurboAE_rate3_cnn_dense", precision="f32")):
        cfg = TurboAEConfig(**over)
        cfg.validate()
        assert cfg.generic, over
    # r06: the optional one-product decoder precision exists for the 100-wide whole-block CNN decoder only, and is never generic
    TurboAEConfig(precision="f16x1").validate()
    assert not TurboAEConfig(precision="f16x1").generic
    for over in (dict(decoder="TurboAE_rate3_rnn"), dict(dec_num_unit=32), dict(block_len=321), dict(dec_num_unit=128, enc_num_unit=128)):
        with pytest.raises(ValueError):
            TurboAEConfig(precision="f16x1", **over).validate()
    # r06: LSTM / vanilla-RNN cells in the 2-layer encoder itself run on the tuned kernels in precision auto, on the generic ones in f32
    assert not TurboAEConfig(encoder="TurboAE_rate3_rnn", decoder="TurboAE_rate3_rnn", enc_rnn="lstm").generic
    assert TurboAEConfig(encoder="TurboAE_rate3_rnn", decoder="TurboAE_rate3_rnn", enc_rnn="lstm", precision="f32").generic
    assert TurboAEConfig(encoder="TurboAE_rate3_rnn", decoder="TurboAE_rate3_rnn", enc_rnn="rnn", enc_num_layer=1).generic
    # r06: the 2-layer GRU encoder in front of an LSTM / vanilla-RNN decoder runs on the tuned kernels (GRU kernels + turboae_rnn_u.hip)
    assert not TurboAEConfig(encoder="TurboAE_rate3_rnn", decoder="TurboAE_rate3_rnn", dec_rnn="lstm").generic
    # r05: LSTM / vanilla-RNN decoders behind the CNN encoder have unit-split f16x2 kernels (turboae_rnn_u.hip)
    for over in (dict(decoder="TurboAE_rate3_rnn", dec_rnn="lstm"), dict(decoder="TurboAE_rate3_rnn", dec_rnn="rnn", dec_num_unit=40)):
        cfg = TurboAEConfig(**over)
        cfg.validate()
        assert not cfg.generic, over
    for over in (dict(enc_num_unit=2000), dict(dec_kernel_size=12), dict(num_iter_ft=100), dict(dec_rnn="elman"), dict(dec_kernel_size=65)):
        with pytest.raises(ValueError):
            TurboAEConfig(**over).validate()
    with pytest.raises(ValueError):
        TurboAEConfig(code_rate_n=2).validate()


def test_variant_configs_and_param_counts():
    # GRU decoder: SURVEY.md section 8f-3 measured 2 970 456 parameters for DEC_LargeRNN
    rnn = TurboAEConfig(decoder="TurboAE_rate3_rnn")
    assert sum(int(np.prod(s)) for k, s in W.canonical_entries(rnn) if k.startswith("dec.")) == 2970456
    # GRU encoder needs the GRU decoder; dense stacks need the fp16-split kernels and are keyed on the ENCODER name
    TurboAEConfig(encoder="TurboAE_rate3_rnn", decoder="TurboAE_rate3_rnn").validate()
    TurboAEConfig(encoder="TurboAE_rate3_rnn", decoder="TurboAE_rate3_rnn", enc_num_unit=40, dec_num_unit=25).validate()
    # an RNN encoder in front of the CNN decoder: the reference then builds the decoder from DenseSameShapeConv1d (decoders.py:173-176)
    mixed = TurboAEConfig(encoder="TurboAE_rate3_rnn")
    mixed.validate()
    assert mixed.dec_dense and not mixed.dense and dict(W.canonical_entries(mixed))["dec.dec1_cnns.0.cnns.2.weight"] == (100, 7 + 200, 5)
    lstm = TurboAEConfig(decoder="TurboAE_rate3_rnn", dec_rnn="lstm")
    assert dict(W.canonical_entries(lstm))["dec.dec2_rnns.3.weight_hh_l1_reverse"] == (400, 100)
    dense = TurboAEConfig(encoder="TurboAE_rate3_cnn_dense", decoder="TurboAE_rate3_cnn_dense")
    dense.validate()
    assert dense.dense and not TurboAEConfig().dense
    with pytest.raises(ValueError):
        TurboAEConfig(decoder="TurboAE_rate3_cnn_dense").validate()
    shapes = dict(W.canonical_entries(dense))
    assert shapes["dec.dec1_cnns.0.cnns.4.weight"] == (100, 7 + 400, 5)        # cnn_utils.py:59-62: in_channels + idx * out_channels
    assert shapes["enc.enc_cnn_3.cnns.1.weight"] == (100, 1 + 100, 5)
    assert dense.macs_per_bit()["dec"] > 2 * TurboAEConfig().macs_per_bit()["dec"]
    with pytest.raises(ValueError):
        TurboAEConfig(channel="nope").validate()
    TurboAEConfig(channel="fading").validate()


# ---- static ISA hazard audit (tools/isa_audit.py) -----------------------------------------------------------
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

def _isa_audit():
    import importlib.util
    spec = importlib.util.spec_from_file_location("isa_audit", os.path.join(ROOT, "tools", "isa_audit.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_isa_audit_is_clean_on_the_built_library():
    """One MFMA shape per kernel, no scratch on the kernels the bench times, no memory instruction in inline assembly - on the
    library as built (the code objects inside libturboae_hip.so and their metadata), in a few seconds."""
    A = _isa_audit()
    if not A.tools_available() or not os.path.isfile(A.DEFAULT_LIB):
        pytest.skip("needs the ROCm llvm tools and the built library")
    res = A.audit_binary(A.DEFAULT_LIB)
    assert res["violations"] == [], res["violations"]
    assert A.audit_sources() == []
    names = list(res["kernels"])
    for prefix in A.BENCH_KERNELS:                       # every bench kernel was actually found (a renamed kernel must not drop out of rule ii)
        assert any(n.startswith(prefix) for n in names), prefix
    for inst in ("tae::dec_kernel_h<100, 5, false, false, 3>", "tae::dec_kernel_h<100, 5, false, true, 3>", "tae::dec_kernel_h<100, 5, false, false, 1>",
                 "tae::(anonymous namespace)::gru_l1f_kernel"):
        k = res["kernels"][[n for n in names if n.startswith(inst)][0]]       # plain decoder, its both-branch-head twin, the one-product (f16x1) decoder, the fused GRU layer 1
        # (the both-branch-head twin may hold the audit's documented cold-scratch allowance since the two-MFMA tail slabs, isa_audit.COLD_SCRATCH_OK)
        cold = any(inst.startswith(p) for p in A.COLD_SCRATCH_OK)
        assert list(k["mfma"]) == ["v_mfma_f32_16x16x32_f16"] and k["vgpr"] <= 256, inst
        assert (k["scratch"] == 0 and k["vgpr_spill"] == 0) or (cold and k["scratch"] <= A.COLD_SCRATCH_MAX), inst
    assert len(names) >= 88


def test_isa_audit_flags_a_mixed_shape_kernel_and_inline_asm_loads(tmp_path):
    """Red cases: the isolated gfx950 hazard (tools/lab/probes/mfma_mixed_shape_hazard.hip mixes 16x16x32 and 16x16x16 f16 MFMAs on
    purpose) must be reported by rule (i); a source with a hand-issued global_load by rule (iii)."""
    import shutil
    import subprocess
    A = _isa_audit()
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not A.tools_available() or not os.path.isfile(hipcc):
        pytest.skip("needs hipcc and the ROCm llvm tools")
    obj = str(tmp_path / "mixed.o")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-c", os.path.join(ROOT, "tools", "lab", "probes", "mfma_mixed_shape_hazard.hip"), "-o", obj])
    res = A.audit_binary(obj)
    assert any(v.startswith("(i) mixed MFMA shapes") for v in res["violations"]), res
    src = tmp_path / "csrc"
    src.mkdir()
    (src / "bad.hip").write_text('__device__ void f(float* p, float& v) {\n    asm volatile("global_load_dword %0, %1, off" : "=v"(v) : "v"(p));\n'
                                 '    asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(v) : "v"(v), "v"(v));\n}\n')
    bad = A.audit_sources(str(src))
    assert len(bad) == 1 and "global_load" in bad[0] and "bad.hip:2" in bad[0], bad


def test_y0_region_layout_is_a_direction_private_bijection(tmp_path):
    """turboae_y0.hpp (layer-0 outputs of the f16x2 recurrent stacks in HBM): the header's own offset functions, compiled for the host,
    map the 16 rows x 2 planes x 200 halves of a (group, step) one-to-one onto its 12 800 bytes; region A holds forward units only,
    region B backward units only, region C the one shared 16-byte piece per row; the writers' linear-in-tile shortcut and the readers'
    three per-lane bases (slabs 0..2 | slab 3 | slabs 4..) agree with the per-half definition."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.isfile(hipcc):
        pytest.skip("needs hipcc")
    src = tmp_path / "y0.cpp"
    src.write_text('#include <hip/hip_runtime.h>\n#include <stdio.h>\n#include "turboae_y0.hpp"\n'
                   'int main() {\n'
                   '  for (int n = 0; n < 16; ++n) for (int j = 0; j < 200; ++j) printf("H %d %d %u %u\\n", n, j, tae::y0_half(n, j), tae::y0_half_lo(n, j));\n'
                   '  for (int n = 0; n < 16; ++n) for (int p = 0; p < 25; ++p) printf("P %d %d %u %u\\n", n, p, tae::y0_piece(n, p), tae::y0_lo_add(p));\n'
                   '  printf("K %u %u %u %u %u %u\\n", tae::kY0StepB, tae::kY0A, tae::kY0B, tae::kY0C, tae::kY0PlaneAB, tae::kY0PlaneC);\n  return 0;\n}\n')
    exe = str(tmp_path / "y0")
    subprocess.check_call([hipcc, "-O1", "-I", os.path.join(ROOT, "turboae_amd", "csrc"), str(src), "-o", exe])
    hi, lo, piece = {}, {}, {}
    for line in subprocess.check_output([exe], text=True).splitlines():
        f = line.split()
        if f[0] == "H":
            hi[(int(f[1]), int(f[2]))], lo[(int(f[1]), int(f[2]))] = int(f[3]), int(f[4])
        elif f[0] == "P":
            piece[(int(f[1]), int(f[2]))] = (int(f[3]), int(f[4]))
        else:
            step, a0, b0, c0, plane_ab, plane_c = map(int, f[1:])
    assert (step, a0) == (16 * 800, 0) and b0 < c0 < step
    offs = list(hi.values()) + list(lo.values())
    assert len(set(offs)) == 2 * 16 * 200 and min(offs) == 0 and max(offs) == step - 2 and all(o % 2 == 0 for o in offs)      # a bijection on halves
    for (n, j), o in hi.items():
        assert o == piece[(n, j >> 3)][0] + 2 * (j & 7) and lo[(n, j)] == o + piece[(n, j >> 3)][1]                            # halves sit in their 16-byte piece
        region = "A" if o < b0 else ("B" if o < c0 else "C")
        assert region == ("A" if j < 96 else ("C" if j < 104 else "B"))                                                         # forward | shared piece 12 | backward
        assert (lo[(n, j)] - o) == (plane_c if region == "C" else plane_ab)
    line = lambda o: o // 128
    fwd = {line(o) for (n, j), o in list(hi.items()) + list(lo.items()) if j < 96}
    bwd = {line(o) for (n, j), o in list(hi.items()) + list(lo.items()) if j >= 104}
    assert not (fwd & bwd) and len(fwd) == len(bwd) == 48                                                                       # no 128-byte line has two writers outside region C
    for d in (0, 1):                                       # writers: tile u >= 1 = tile 1 + 32 (u - 1), lo = hi + kY0PlaneAB; four units are 8 contiguous bytes
        for n in range(16):
            for q in range(4):
                for u in range(6):
                    j = d * 100 + 16 * u + 4 * q
                    assert [hi[(n, j + k)] for k in range(4)] == [hi[(n, j)] + 2 * k for k in range(4)]
                    if u >= 1:
                        assert hi[(n, j)] == hi[(n, d * 100 + 16 + 4 * q)] + 32 * (u - 1) and lo[(n, j)] == hi[(n, j)] + plane_ab
    for n in range(16):                                    # readers: lane (n, q) takes piece 4 sl + q of slab sl from vA / v3 / vB
        for q in range(4):
            vA, v3, vB = piece[(n, q)][0], piece[(n, 12 + q)][0], piece[(n, 16 + q)][0]
            for sl in range(6):
                want = piece[(n, 4 * sl + q)][0]
                assert want == (vA + 64 * sl if sl < 3 else (v3 if sl == 3 else vB + 64 * (sl - 4)))
            if q == 0:
                assert piece[(n, 24)][0] == vB + 128
