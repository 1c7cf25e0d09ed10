"""Build audits of the looped two-seeds-per-lane kernel (CPU only: hipcc cross-compiles gfx950; ~30 s).

The kernel's design leans on three properties of the generated code that no functional test sees:
  * its AccVGPRs are its own: the compiler allocates none and spills nothing;
  * the block loops sit on the 8-byte instruction grid (one dword of shift cost 7 %, profiles/r3/fetch_alignment.txt);
  * the N-free block loops carry no test per block: ~15 scalar instructions each.
"""
import os
import re
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
SRC = os.path.join(ROOT, "bwa-mem-sw_amd", "csrc", "bsw_lane2l_kernel.hip")
FLAGS = ["-mllvm", "-amdgpu-sched-strategy=max-ilp"]          # the Makefile's L2L_SCHED

pytestmark = pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")


@pytest.fixture(scope="module")
def listing():
    import isa_align
    return isa_align, isa_align.disassemble(SRC, FLAGS)


def test_makefile_builds_the_kernel_with_the_audited_flags():
    mk = open(os.path.join(ROOT, "bwa-mem-sw_amd", "csrc", "Makefile")).read()
    assert "L2L_SCHED ?= " + " ".join(FLAGS) in mk


def test_block_loops_sit_on_the_instruction_grid(listing):
    isa_align, text = listing
    for kern in ("Li29ELi1ELb0ELb1", "Li29ELi1ELb0ELb0", "Li29ELi1ELb1ELb1", "Li29ELi1ELb1ELb0"):
        ls = isa_align.loops(text, kern)
        # the block loops: one or two block bodies (>= 100 64-bit instructions) and little else
        block = [l for l in ls if 100 <= l["wide"] and l["instructions"] <= 640]
        assert len(block) >= 4, (kern, ls)          # N-free dense / edge, general dense / edge (the latter with two back edges)
        for l in block:
            assert l["wide_off_grid"] <= 4, (kern, l)
        # the test-free loops: a block body, the folds, the swap statement and its 15 scalar instructions
        lean = [l for l in block if l["scalar"] <= 24]
        assert len(lean) >= 2, (kern, block)


def test_compiler_leaves_the_accumulator_registers_alone():
    """No scratch, no compiler-generated AccVGPR traffic: every v_accvgpr_* of the listing comes from the row accessors'
    asm statements (which name a0..a15 or the match-word registers literally)."""
    out = subprocess.check_output(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-inline-asm", "-Wno-unused-function",
                                   "-I" + os.path.join(ROOT, "include"), "--cuda-device-only", "-S", SRC, "-o", "-"] + FLAGS,
                                  stderr=subprocess.DEVNULL, text=True)
    assert re.findall(r"ScratchSize: (\d+)", out) and all(int(x) == 0 for x in re.findall(r"ScratchSize: (\d+)", out))
    # the accessors name a0..a15 (block-relative, offset by M0) and the match words behind the row: a[QMAX+8 .. QMAX+8+2NW)
    # = a144..a153 for the 136-column instantiation, a240..a255 for the 232-column one; a compiler spill would sit elsewhere
    allowed = set(range(0, 16)) | set(range(144, 154)) | set(range(240, 256))
    stray = [l.strip() for l in out.splitlines() if "v_accvgpr" in l and not l.lstrip().startswith(";")
             and int(re.search(r"\ba\[?(\d+)", l).group(1)) not in allowed]
    assert not stray, stray[:5]
    assert out.count("v_accvgpr_read_b32") > 100 and out.count("v_accvgpr_write_b32") > 100
