"""Static check of the compiled Winograd kernels: gfx950 does not interlock a VALU write with an MFMA that reads the
register as SrcA/B in the next two issue slots (tools/probes/valu_mfma_hazard.hip, profiles/r02_valu_mfma_hazard_probe.txt).
The compiler adds those wait states for the code it generates, but the input transform of csrc/conv3d_wino.hip and
csrc/conv2d_wino.hip is inline asm it cannot see through, so this test compiles both files to ISA (hipcc cross-compiles
without a GPU) and verifies that no v_mfma reads a register that an inline-asm instruction wrote fewer than two wait
states earlier -- whatever schedule the compiler picked."""
import re
import subprocess
from pathlib import Path

import pytest

from diffuvolume_amd import _build

ROOT = Path(__file__).resolve().parents[1]
REG = re.compile(r"^v\[(\d+):(\d+)\]$|^v(\d+)$")


def regs(tok):
    m = REG.match(tok.strip())
    if not m:
        return []
    if m.group(3) is not None:
        return [int(m.group(3))]
    return list(range(int(m.group(1)), int(m.group(2)) + 1))


def lint(asm_text, m0_report=None):
    """-> (violations, number of inline-asm VALU producers seen, number of MFMAs seen).  ``m0_report`` (a dict) receives
    the second check: an LDS-DMA (``*_load_lds_*`` / ``buffer_load ... lds``) reads M0, and gfx9 needs one wait state
    between a scalar write of M0 and that read; the compiler's hazard recogniser does not look inside inline asm, so
    every DMA whose M0 write sits in an ASMSTART/ASMEND block must be at least one slot behind it."""
    in_asm, slot = False, 0
    written = {}            # vgpr -> slot index of the inline-asm instruction that wrote it last
    bad, n_prod, n_mfma = [], 0, 0
    m0_written_asm = None   # slot of the last inline-asm write of m0 (None once the compiler wrote it: that one it guards)
    m0_bad, n_dma = [], 0
    for raw in asm_text.splitlines():
        line = raw.strip()
        if line.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if line.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not line or line[0] in ";." or line.endswith(":"):
            continue
        op, _, rest = line.partition(" ")
        ops = [t for t in re.split(r"[ ,]+", rest.split(";")[0]) if t]
        if op == "s_nop":
            slot += int(ops[0], 0) + 1
            continue
        if "_load_lds_" in op or (op.startswith("buffer_load") and "lds" in ops):
            n_dma += 1
            if m0_written_asm is not None and slot - m0_written_asm - 1 < 1:
                m0_bad.append((line, slot - m0_written_asm - 1))
        if ops and ops[0] == "m0" and op.startswith("s_"):
            m0_written_asm = slot if in_asm else None
        if op.startswith("v_mfma"):
            n_mfma += 1
            for tok in ops[1:3]:                                  # SrcA, SrcB
                for r in regs(tok):
                    if r in written and slot - written[r] - 1 < 2:
                        bad.append((line, r, slot - written[r] - 1))
        if in_asm and op.startswith("v_") and ops:
            n_prod += 1
            for r in regs(ops[0]):
                written[r] = slot
        elif ops:                                                 # a compiler-generated write supersedes the asm one
            for r in regs(ops[0]):
                written.pop(r, None)
        slot += 1
    if m0_report is not None:
        m0_report.update(bad=m0_bad, n_dma=n_dma)
    return bad, n_prod, n_mfma


@pytest.mark.parametrize("name", ["conv3d_wino", "conv2d_wino"])
def test_no_asm_valu_result_is_read_by_an_mfma_too_early(name, tmp_path):
    src = ROOT / "diffuvolume_amd" / "csrc" / f"{name}.hip"
    out = tmp_path / f"{name}.s"
    flags = [f for f in _build.FLAGS if f != "-fPIC"]
    r = subprocess.run([_build._hipcc(), *flags, "-S", "--cuda-device-only", str(src), "-o", str(out)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    m0 = {}
    bad, n_prod, n_mfma = lint(out.read_text(), m0)
    assert n_prod >= 16 and n_mfma >= 64, (n_prod, n_mfma)        # the transform and the MFMA stream were really seen
    assert not bad, bad[:5]
    assert m0["n_dma"] >= 6, m0                                   # the weight DMA was really seen
    assert not m0["bad"], m0["bad"][:5]


def test_no_asm_valu_result_is_read_by_an_mfma_too_early_in_the_3d_winograd_kernel(tmp_path):
    """csrc/conv3d_wino3.hip: the two 14-instruction transform bursts (depth combination + rows + columns) are inline asm
    whose results are MFMA A operands; it has no LDS-DMA (weights and bricks travel through registers)."""
    src = ROOT / "diffuvolume_amd" / "csrc" / "conv3d_wino3.hip"
    out = tmp_path / "conv3d_wino3.s"
    flags = [f for f in _build.FLAGS if f != "-fPIC"]
    r = subprocess.run([_build._hipcc(), *flags, "-S", "--cuda-device-only", str(src), "-o", str(out)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    m0 = {}
    bad, n_prod, n_mfma = lint(out.read_text(), m0)
    assert n_prod >= 3 * 2 * 14 and n_mfma >= 3 * 64, (n_prod, n_mfma)
    assert not bad, bad[:5]
    assert m0["n_dma"] == 0, m0


def test_m0_wait_state_in_the_persistent_deconv(tmp_path):
    """csrc/deconv3d_pl.hip issues its skip-tile LDS-DMA from inline asm too: the M0 rule on its compiled stream (the loader
    waves' DMAs come from the builtin, whose M0 writes the compiler guards itself)."""
    src = ROOT / "diffuvolume_amd" / "csrc" / "deconv3d_pl.hip"
    out = tmp_path / "deconv3d_pl.s"
    flags = [f for f in _build.FLAGS if f != "-fPIC"]
    r = subprocess.run([_build._hipcc(), *flags, "-S", "--cuda-device-only", str(src), "-o", str(out)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    m0 = {}
    lint(out.read_text(), m0)
    assert m0["n_dma"] >= 6 * 2 + 2 * 3, m0["n_dma"]     # skip-tile pieces of the three SKIP instantiations + the loaders' pieces
    assert not m0["bad"], m0["bad"][:5]


def test_lint_catches_the_hazard():
    text = """
	;;#ASMSTART
	v_pk_add_f32 v[10:11], v[2:3], v[4:5]
	;;#ASMEND
	v_mov_b32 v20, 0
	v_mfma_f32_16x16x4_f32 v[30:33], v10, v7, v[30:33]
	;;#ASMSTART
	v_pk_add_f32 v[12:13], v[2:3], v[4:5]
	s_nop 1
	;;#ASMEND
	v_mfma_f32_16x16x4_f32 v[30:33], v12, v7, v[30:33]
"""
    bad, n_prod, n_mfma = lint(text)
    assert n_prod == 2 and n_mfma == 2
    assert len(bad) == 1 and bad[0][1] == 10 and bad[0][2] == 1


def test_lint_catches_the_m0_hazard():
    text = """
	;;#ASMSTART
	s_mov_b32 s5, m0
	s_mov_b32 m0, s7
	global_load_lds_dwordx4 v[2:3], off
	s_mov_b32 m0, s5
	;;#ASMEND
	;;#ASMSTART
	s_mov_b32 s5, m0
	s_mov_b32 m0, s7
	s_nop 0
	global_load_lds_dwordx4 v[2:3], off
	s_mov_b32 m0, s5
	;;#ASMEND
	s_mov_b32 m0, s9
	global_load_lds_dwordx4 v[2:3], off
"""
    m0 = {}
    lint(text, m0)
    assert m0["n_dma"] == 3 and len(m0["bad"]) == 1 and m0["bad"][0][1] == 0
