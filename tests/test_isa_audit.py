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
        # (a back edge that also spans a row's wave reductions is a short path of the ROW loop — a row of one block — not a block loop)
        block = [l for l in ls if 100 <= l["wide"] and l["instructions"] <= 640 and not l["row_head"]]
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


def test_three_wave_kernel_spills_only_outside_its_row_loop():
    """bsw_lane2_kernel<9, 3, ...> (the 72-column class) is held to 168 VGPRs so that three waves share a SIMD; the compiler
    pays for that with a dozen spills.  They must sit in the prologue and in the cold code behind the row loop (target
    staging one row in 64, query-N bodies) — a scratch access inside the row loop would cost every row a memory round trip."""
    src = os.path.join(ROOT, "bwa-mem-sw_amd", "csrc", "bsw_lane2_kernel.hip")
    out = subprocess.check_output(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-DBSW_L2_ASM_BODY=1", "-Wno-unused-function",
                                   "-I" + os.path.join(ROOT, "include"), "--cuda-device-only", "-S", src, "-o", "-"], stderr=subprocess.DEVNULL, text=True)
    seen = 0
    for m in re.finditer(r"^(_ZN3bsw16bsw_lane2_kernelILi9ELi3E\w+):.*?s_endpgm", out, re.S | re.M):
        body = m.group(0).split("\n")
        labels = {l.split(":")[0]: i for i, l in enumerate(body) if re.match(r"^\.LBB\d+_\d+:", l)}
        loops = []
        for i, l in enumerate(body):
            b = re.search(r"s_c?branch\w* (\.LBB\d+_\d+)", l)
            if b and labels.get(b.group(1), 1 << 30) < i:
                lo = labels[b.group(1)]
                loops.append((sum(1 for x in body[lo:i] if "v_pk_" in x), lo, i))
        # the row loop: the first back edge that closes around the hot block bodies (the cold code — target staging, query-N
        # bodies — is laid out BEHIND it and jumps back into it: those later back edges span it too)
        npk, lo, hi = min((x for x in loops if x[0] > 1000 and x[2] - x[1] > 3000), key=lambda x: x[2])
        inside = [x.strip() for x in body[lo:hi] if "scratch_" in x]
        assert not inside, (m.group(1), inside[:4])
        vg = re.search(r"\.vgpr_count:\s+(\d+)", out[out.index(".name:           " + m.group(1)):])
        assert vg and int(vg.group(1)) <= 168, vg and vg.group(1)
        seen += 1
    assert seen == 4                                           # variant H / M x shared / separate gap penalties


def _quad_listing():
    src = os.path.join(ROOT, "bwa-mem-sw_amd", "csrc", "bsw_quad_kernel.hip")
    return subprocess.check_output(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-function",
                                    "-I" + os.path.join(ROOT, "include"), "--cuda-device-only", "-S", src, "-o", "-"], stderr=subprocess.DEVNULL, text=True)


def test_quad_kernel_dpp_reads_keep_their_wait_states_and_its_spills_stay_out_of_the_rows():
    """bsw_quad_kernel's hand-written DPP sequences (row_scan_max, shr1_max, bcast15_max, row_max4, the packed min / max
    reduction) sit in asm statements the hazard recogniser does not look into: a DPP read of a VGPR needs TWO wait states
    behind the VALU instruction that wrote it, and only the statements' own s_nops / instruction order provide them.  A
    compiler or flag change that puts a VALU write of the source directly in front of one of them would break the kernel
    silently (round 4's advisor).  Audit: for every v_*_dpp of every instantiation, no VALU instruction among the two wait
    states in front of it writes its DPP source.  And the spills the launch bounds cost (3 / 4 waves per SIMD) must sit in the
    rare try / side bookkeeping loop, never on the path every DP row takes."""
    out = _quad_listing()
    kernels = list(re.finditer(r"^(_ZN3bsw15bsw_quad_kernelILi(\d+)ELi(\d+)E\w+):.*?s_endpgm", out, re.S | re.M))
    assert len(kernels) == 8                                    # 2 / 4 / 6 / 8 stripes x variant H / M
    for m in kernels:
        body = [l for l in m.group(0).split("\n")]
        ins = []                                                # (line index, mnemonic, operands)
        for i, l in enumerate(body):
            t = l.strip()
            if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
                continue
            mm = re.match(r"([a-z_0-9]+)\s*(.*)", t)
            if mm:
                ins.append((i, mm.group(1), mm.group(2)))
        ndpp = 0
        for k, (i, op, args) in enumerate(ins):
            if not (op.startswith("v_") and ("_dpp" in op or "row_" in args or "quad_perm" in args)):
                continue
            ndpp += 1
            regs = [a.strip() for a in args.split(",")]
            src = regs[1].split()[0]                            # vDST, vSRC0 (the operand the DPP control applies to), ...
            if not src.startswith("v"):
                continue
            wait, j = 0, k - 1
            while j >= 0 and wait < 2:
                _, pop, pargs = ins[j]
                if pop == "s_nop":
                    wait += int(pargs.split()[0]) + 1
                else:
                    pdst = pargs.split(",")[0].strip()
                    assert not (pop.startswith("v_") and pdst == src), (m.group(1)[:40], body[i].strip(), "written by", pop, pargs)
                    wait += 1
                j -= 1
        assert ndpp > 50, (m.group(1), ndpp)
        # spills: none in a basic block of the row path — the blocks that hold a stripe's F scan (row_shr:8), the row head's
        # target fetch (ds_bpermute) or the row tail's four reductions (row_ror:8); the bookkeeping's blocks may have them
        blocks, cur = [], []
        for l in body:
            if re.match(r"^\.LBB\d+_\d+:", l) or l.startswith("; %bb."):
                blocks.append(cur); cur = []
            cur.append(l)
        blocks.append(cur)
        row_blocks = [b for b in blocks if any(("row_shr:8" in x or "ds_bpermute" in x or "row_ror:8" in x) and not x.lstrip().startswith(";") for x in b)]
        assert len(row_blocks) >= 3, (m.group(1), len(row_blocks))
        for b in row_blocks:
            bad = [x.strip() for x in b if "scratch_" in x]
            assert not bad, (m.group(1)[:40], bad[:4])
