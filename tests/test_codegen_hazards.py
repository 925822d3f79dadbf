"""CPU: the compiler must not borrow an accumulator register of the hand-scheduled loops.

The K loop of the 256 x 256 GEMM keeps its 256 accumulators in a0-a255, the attention KV loop its O^T accumulators in a0-a95; both are
ONE inline-asm statement, and the epilogue reads the registers back with separate `v_accvgpr_read` statements.  To the compiler those
registers are clobbers - dead after the asm - so a register allocator that runs out of VGPRs in the epilogue is free to park a value in
one of them (gfx90a+ has a unified file).  Round 5 met exactly that in an experimental epilogue (`v_accvgpr_write_b32 a0, v2` in front
of the C staging loop: one accumulator element per lane silently replaced).  The shipped kernels do not do it; this
test keeps it that way: it compiles the two sources to device assembly with the library's flags and checks every compiler-generated
instruction (outside ASMSTART / ASMEND) of every kernel that contains one of the loops.  ~40 s of hipcc."""
import os
import re
import subprocess

import pytest

from regione_amd import build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _device_asm(src, tmp_path):
    out = os.path.join(str(tmp_path), src.replace(".hip", ".s"))
    cmd = [build.HIPCC, *build.FLAGS, *build.EXTRA.get(src, []), "--cuda-device-only", "-S", "-o", out, os.path.join(build.CSRC, src)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    return open(out).read()


def _kernels(asm):
    """name -> (compiler-generated lines, inline-asm lines) for every function of the translation unit."""
    res = {}
    for m in re.finditer(r"^(_Z\w+):\s*(?:;.*)?$", asm, re.M):
        end = asm.find(".Lfunc_end", m.end())
        own, inl, inside = [], [], False
        for line in asm[m.end():end].split("\n"):
            if "ASMSTART" in line:
                inside = True
            elif "ASMEND" in line:
                inside = False
            else:
                (inl if inside else own).append(line.split(";")[0].strip())
        res[m.group(1)] = (own, inl)
    return res


def _agpr_writes(lines):
    """AGPR numbers written by these instructions (first operand a<N> or a[N:M]; stores and compares have no AGPR destination)."""
    out = []
    for l in lines:
        m = re.match(r"(\S+)\s+a(?:(\d+)|\[(\d+):(\d+)\])\s*,", l)
        if m and not m.group(1).startswith(("global_store", "buffer_store", "scratch_store", "ds_write", "flat_store")):
            lo = int(m.group(2) if m.group(2) is not None else m.group(3))
            hi = int(m.group(2) if m.group(2) is not None else m.group(4))
            out.append((lo, hi, l))
    return out


@pytest.mark.parametrize("src,live", [("gemm.hip", 256), ("attn.hip", 96)])
def test_no_compiler_generated_write_to_an_accumulator_register(src, live, tmp_path):
    if not os.path.exists(build.HIPCC):
        pytest.skip("hipcc not available")
    kernels = _kernels(_device_asm(src, tmp_path))
    checked = 0
    for name, (own, inl) in kernels.items():
        if not any(l.startswith("v_mfma") for l in inl):          # only kernels built around a hand-scheduled loop
            continue
        checked += 1
        bad = [l for lo, hi, l in _agpr_writes(own) if lo < live]
        assert not bad, f"{name}: the compiler writes accumulator registers of the asm loop: {bad[:4]}"
        # and the checker sees the loop's own accumulator traffic (it would see the compiler's)
        assert _agpr_writes(inl), name
    assert checked >= (8 if src == "gemm.hip" else 4), (src, checked)


def test_the_checker_reports_a_borrowed_accumulator():
    asm = """
_ZN3rgn4demoEv:                         ; @demo
\tv_mov_b32_e32 v1, 0
\t;;#ASMSTART
\tv_accvgpr_write_b32 a0, 0
\tv_mfma_f32_16x16x32_bf16 a[0:3], v[128:131], v[160:163], a[0:3]
\t;;#ASMEND
\tv_accvgpr_write_b32 a0, v2
\tglobal_load_dwordx4 a[100:103], v[4:5], off
\tglobal_store_dwordx4 v[4:5], a[8:11], off
\tv_accvgpr_read_b32 v2, a0
.Lfunc_end0:
"""
    (own, inl), = _kernels(asm).values()
    assert any(l.startswith("v_mfma") for l in inl)
    w = _agpr_writes(own)
    assert [(lo, hi) for lo, hi, _ in w] == [(0, 0), (100, 103)]             # the store and the read are not writes
    assert [l for lo, hi, l in w if lo < 96] == ["v_accvgpr_write_b32 a0, v2"]
