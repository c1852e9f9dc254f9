#!/usr/bin/env python
"""Static hazard audit of the built gfx950 code - no GPU, no recompilation: reads the code objects inside
turboae_amd/lib/libturboae_hip.so (or any hipcc object / library given on the command line).

    python tools/isa_audit.py [lib.so | obj.o ...]        exit code 1 on a violation

Rules, each from a defect this library has actually met on gfx950 with the ROCm 7.2 hipcc:
  (i)   ONE MFMA shape per kernel.  A v_mfma_f32_16x16x16_f16 that takes the result of a v_mfma_f32_16x16x32_f16 issued just before it
        as srcC reads a stale accumulator - the compiler inserts no wait states for that pair (DESIGN.md 3.5,
        tools/lab/probes/mfma_mixed_shape_hazard.hip, profiles/r03_mfma_mixed_shape_hazard.txt).  The set of v_mfma_* mnemonics of every
        kernel must have at most one element.
  (ii)  No scratch on the kernels the bench times (BENCH_KERNELS): private_segment_fixed_size = 0 and no spilled vector registers, read
        from the code object's own metadata (r02: 272 bytes of scratch per lane cost the decoder 1.2 %; r04: three spilled registers
        tripled the decoder's HBM writes).  Spilled SCALAR registers live in vector-register lanes and are reported, not failed.
        One exception (late r06): the kernels of COLD_SCRATCH_OK may hold up to COLD_SCRATCH_MAX bytes of scratch - the both-branch-head
        twin of the decoder parks seven loop-invariant registers before its stack loop and reloads them once per layer (60 times per
        workgroup, outside the K loops); bench.py reports what that costs (cfg1_head2_decoder_over_plain).  The audit prints how many
        scratch instructions of every kernel lie inside a loop (a backward branch spans them).
  (iii) No memory instruction inside inline assembly.  The compiler keeps the vmcnt / lgkmcnt bookkeeping of the loads IT issues; a
        hand-issued global_load / buffer_load / ds_read is invisible to it, so every s_waitcnt vmcnt(k) it computes afterwards is off by
        the number of hand-issued loads in flight - the likeliest reading of the run-to-run differences of r03's one-step-ahead GI
        fetch (unexplained then; that variant was never kept in the tree).  Checked on the SOURCES (asm statements of
        csrc/*.hip, *.hpp): the disassembly no longer knows which instructions came from an asm statement.
"""
from __future__ import annotations

import glob
import os
import re
import shutil
import subprocess
import sys
import tempfile
from typing import Dict, List

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
DEFAULT_LIB = os.path.join(ROOT, "turboae_amd", "lib", "libturboae_hip.so")
CSRC = os.path.join(ROOT, "turboae_amd", "csrc")

# demangled-name prefixes of the kernels bench.py times (headline line + roofline.other_configs), production instantiations only
BENCH_KERNELS = [r"tae::dec_kernel_h<100, 5, false", r"tae::enc_kernel_h<100, 5, 0", r"tae::seg_kernel_h<100, 5>",
                 r"tae::dec_kernel<100, 5, false>", r"tae::enc_kernel<100, 5>",
                 r"tae::gru_rec_h_kernel<true>", r"tae::gru_rec0u_kernel", r"tae::(anonymous namespace)::gru_l1f_kernel", r"tae::gru_head_part_kernel",
                 r"tae::(anonymous namespace)::rnn_rec_u_kernel<4, ", r"tae::(anonymous namespace)::rnn_proj_u_kernel<25>",      # LSTM decoder line
                 r"tae::(anonymous namespace)::rnn_l1f_u_kernel<4, 1>",                                                              # ... its fused layer 1 (r06)
                 # roofline.generic_configs (LSTM decoder, 256-wide CNN pair) and what bench.py times beside the kernels above
                 r"tae::(anonymous namespace)::gen_conv_mfma_kernel", r"tae::(anonymous namespace)::gen_proj_mfma_kernel",
                 r"tae::(anonymous namespace)::gen_rnn_mfma_kernel", r"tae::normalize_kernel", r"tae::count_errors_vec4_kernel"]
# (ii)'s exception: the both-branch-head twin of the decoder spills seven loop-invariant registers since the two-MFMA tail slabs
# (run_stack_h<.., T20>, late r06); the plain decoder - the headline kernel - has none
COLD_SCRATCH_OK = [r"tae::dec_kernel_h<100, 5, false, true, 3>", r"tae::seg_kernel_h<100, 5>"]      # ... and the long-block kernel: six (same-box A/B with them: -1.0 % per forward at L = 1000)
COLD_SCRATCH_MAX = 64
MEM_ASM = re.compile(r"\b(global_load|global_store|global_atomic|buffer_load|buffer_store|buffer_atomic|flat_load|flat_store|flat_atomic|"
                     r"scratch_load|scratch_store|ds_read|ds_write|ds_load|ds_store|ds_bpermute|ds_permute|s_load|s_buffer_load|"
                     r"tbuffer_load|tbuffer_store)", re.I)


def tools_available() -> bool:
    return all(os.path.isfile(os.path.join(LLVM, t)) for t in ("llvm-objdump", "llvm-readelf")) and shutil.which("c++filt") is not None


def _run(args: List[str], cwd=None) -> str:
    return subprocess.run(args, cwd=cwd, check=True, capture_output=True, text=True).stdout


def code_objects(path: str, tmp: str) -> List[str]:
    """gfx950 code objects inside a hipcc-built shared library / object file (extracted into `tmp`)."""
    local = os.path.join(tmp, os.path.basename(path))
    shutil.copy(path, local)
    _run([os.path.join(LLVM, "llvm-objdump"), "--offloading", os.path.basename(local)], cwd=tmp)     # writes <file>.<n>.<triple> next to the input
    return sorted(f for f in glob.glob(local + ".*gfx950*") if os.path.getsize(f) > 0)


def kernel_metadata(co: str) -> Dict[str, dict]:
    notes = _run([os.path.join(LLVM, "llvm-readelf"), "--notes", co])
    out: Dict[str, dict] = {}
    for block in notes.split("  - .agpr_count:")[1:]:
        def num(key, default=0):
            m = re.search(r"\." + key + r":\s+(\d+)", block)
            return int(m.group(1)) if m else default
        m = re.search(r"\.name:\s+(\S+)", block)
        if not m:
            continue
        out[m.group(1)] = {"vgpr": num("vgpr_count"), "sgpr": num("sgpr_count"), "scratch": num("private_segment_fixed_size"),
                           "vgpr_spill": num("vgpr_spill_count"), "sgpr_spill": num("sgpr_spill_count"), "lds_static": num("group_segment_fixed_size")}
    return out


def kernel_mfma(co: str, hot_scratch: Dict[str, int] = None) -> Dict[str, Dict[str, int]]:
    """symbol -> {v_mfma mnemonic: count} from the disassembly; `hot_scratch` (optional) receives symbol -> number of scratch
    instructions that lie INSIDE a loop (an offset spanned by a backward branch of the same function)."""
    dis = _run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", co])
    out: Dict[str, Dict[str, int]] = {}
    cur, start = None, 0
    scratch: Dict[str, List[int]] = {}
    back: Dict[str, List[tuple]] = {}
    for line in dis.splitlines():
        m = re.match(r"^([0-9a-f]+) <(.+)>:$", line)
        if m:
            cur, start = m.group(2), int(m.group(1), 16)
            out.setdefault(cur, {})
            continue
        if cur is None:
            continue
        m = re.match(r"^\s*(v_mfma_\w+|v_smfmac_\w+)\b", line)
        if m:
            op = re.sub(r"_e64$", "", m.group(1))
            out[cur][op] = out[cur].get(op, 0) + 1
            continue
        a = re.search(r"//\s*([0-9A-Fa-f]+):", line)
        if a is None:
            continue
        off = int(a.group(1), 16) - start
        if re.match(r"^\s*scratch_(load|store)", line):
            scratch.setdefault(cur, []).append(off)
        elif re.match(r"^\s*s_c?branch", line):
            t = re.search(r"<[^>]*\+0x([0-9a-fA-F]+)>\s*$", line)
            tgt = int(t.group(1), 16) if t else (0 if re.search(r"<[^>+]*>\s*$", line) else None)
            if tgt is not None and tgt <= off:
                back.setdefault(cur, []).append((tgt, off))
    if hot_scratch is not None:
        for sym, offs in scratch.items():
            hot_scratch[sym] = sum(1 for o in offs if any(lo <= o <= hi for lo, hi in back.get(sym, [])))
    return out


def demangle(names: List[str]) -> Dict[str, str]:
    if not names:
        return {}
    txt = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout
    return dict(zip(names, [re.sub(r"^void ", "", t) for t in txt.splitlines()]))


def audit_binary(path: str) -> dict:
    """{'kernels': {demangled: {...}}, 'violations': [...]} for one library / object file."""
    kernels, violations = {}, []
    with tempfile.TemporaryDirectory() as tmp:
        cos = code_objects(path, tmp)
        if not cos:
            return {"kernels": {}, "violations": [f"{path}: no gfx950 code object found"]}
        for co in cos:
            hot: Dict[str, int] = {}
            meta, mfma = kernel_metadata(co), kernel_mfma(co, hot)
            names = demangle(list(meta))
            for sym, md in meta.items():
                k = dict(md)
                k["mfma"] = mfma.get(sym, {})
                k["scratch_in_loops"] = hot.get(sym, 0)
                kernels[names[sym]] = k
    for name, k in sorted(kernels.items()):
        if len(k["mfma"]) > 1:
            violations.append(f"(i) mixed MFMA shapes in {name}: {k['mfma']}")
        cold_ok = any(name.startswith(p) for p in COLD_SCRATCH_OK) and 0 < k["scratch"] <= COLD_SCRATCH_MAX
        if any(name.startswith(p) for p in BENCH_KERNELS) and (k["scratch"] > 0 or k["vgpr_spill"] > 0) and not cold_ok:
            violations.append(f"(ii) bench kernel {name} uses scratch: {k['scratch']} bytes per lane, {k['vgpr_spill']} spilled vector registers")
    return {"kernels": kernels, "violations": violations}


def audit_sources(src_dir: str = CSRC) -> List[str]:
    """Rule (iii): memory mnemonics inside asm statements of the kernel sources."""
    bad = []
    for path in sorted(glob.glob(os.path.join(src_dir, "*.hip")) + glob.glob(os.path.join(src_dir, "*.hpp"))):
        text = open(path).read()
        for m in re.finditer(r"\basm\s*(?:volatile)?\s*\(", text):
            depth, i = 1, m.end()
            while i < len(text) and depth:           # the whole parenthesised statement
                depth += {"(": 1, ")": -1}.get(text[i], 0)
                i += 1
            stmt = text[m.start():i]
            strings = " ".join(re.findall(r'"((?:[^"\\]|\\.)*)"', stmt))
            hit = MEM_ASM.search(strings)
            if hit:
                line = text.count("\n", 0, m.start()) + 1
                bad.append(f"(iii) memory instruction '{hit.group(1)}' in inline assembly: {os.path.relpath(path, ROOT)}:{line}")
    return bad


def main(argv: List[str]) -> int:
    paths = argv or [DEFAULT_LIB]
    violations: List[str] = []
    for p in paths:
        res = audit_binary(p)
        print(f"== {p}: {len(res['kernels'])} kernels")
        print(f"{'kernel':78s} {'vgpr':>5s} {'sgpr':>5s} {'scratch':>8s} {'v-spill':>8s} {'s-spill':>8s}  mfma")
        for name, k in sorted(res["kernels"].items()):
            star = "*" if any(name.startswith(b) for b in BENCH_KERNELS) else " "
            mf = ", ".join(f"{op} x{n}" for op, n in sorted(k["mfma"].items())) or "-"
            print(f"{star}{name[:77]:77s} {k['vgpr']:5d} {k['sgpr']:5d} {k['scratch']:8d} {k['vgpr_spill']:8d} {k['sgpr_spill']:8d}  {mf}")
        violations += res["violations"]
    if not argv:
        violations += audit_sources()
    print("(* = timed by bench.py)")
    for v in violations:
        print("VIOLATION", v)
    print("audit:", "FAILED" if violations else "clean")
    return 1 if violations else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
