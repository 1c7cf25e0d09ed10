#!/usr/bin/env python3
"""How the 64-bit instructions of a kernel sit on the instruction stream's 8-byte grid.

Measured on gfx950 (profiles/r3/fetch_alignment.txt): shifting the looped two-seeds-per-lane kernel by ONE dword (a single
s_nop in the prologue) costs 7 % — a 64-bit instruction that starts at an address = 4 (mod 8) is dearer to fetch than one
that starts on the grid, and the block bodies are ~150 64-bit instructions in a row.  The asm statements therefore align
themselves (.p2align 3) and pair their 32-bit scalar instructions; this tool audits the result:

    python3 tools/isa_align.py bwa-mem-sw_amd/csrc/bsw_lane2l_kernel.hip [-D...] [--kernel SUBSTR] [--loops]

prints, per kernel, the number of 64-bit instructions on / off the grid, and the longest off-grid runs with their addresses.
"""
import os, re, subprocess, sys, tempfile

HIPCC = "/opt/rocm/bin/hipcc"
LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def disassemble(src, extra):
    with tempfile.TemporaryDirectory() as d:
        o, elf = os.path.join(d, "k.o"), os.path.join(d, "k.elf")
        subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-inline-asm", "-Wno-unused-function",
                               "-I" + os.path.join(ROOT, "include"), "--cuda-device-only", "-c", src, "-o", o] + extra,
                              stderr=subprocess.DEVNULL)
        subprocess.check_call([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--input=" + o,
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + elf])
        return subprocess.check_output([LLVM + "/llvm-objdump", "-d", elf], text=True)


def audit(text, want):
    out, cur = {}, None
    ins = re.compile(r"^\s+(\S+).*//\s*([0-9A-F]+):((?:\s[0-9A-F]{8})+)")
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
            cur = m.group(1)
            out[cur] = []
            continue
        m = ins.match(line)
        if m and cur:
            out[cur].append((int(m.group(2), 16), len(m.group(3).split()), m.group(1)))
    res = {}
    for k, lst in out.items():
        if want and want not in k:
            continue
        on = sum(1 for a, n, _ in lst if n >= 2 and a % 8 == 0)
        off = sum(1 for a, n, _ in lst if n >= 2 and a % 8 != 0)
        runs, run = [], None
        for a, n, op in lst:
            if n >= 2 and a % 8:
                run = [a, 1] if run is None else [run[0], run[1] + 1]
            elif n >= 2 or run is None:
                if run:
                    runs.append(tuple(run))
                run = None
        if run:
            runs.append(tuple(run))
        runs.sort(key=lambda r: -r[1])
        res[k] = dict(on_grid=on, off_grid=off, longest_off_runs=[(hex(a), n) for a, n in runs[:6]])
    return res


def loops(text, want):
    """Per backward branch of a kernel (= per loop, spans > 600 bytes): instructions, 64-bit ones, how many of those
    start off the 8-byte grid."""
    out, cur, rows = [], None, {}
    ins = re.compile(r"^\s+(\S+)(.*)//\s*([0-9A-F]+):((?:\s[0-9A-F]{8})+)")
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
            cur = m.group(1)
            rows[cur] = []
            continue
        m = ins.match(line)
        if m and cur:
            rows[cur].append((int(m.group(3), 16), len(m.group(4).split()), m.group(1), m.group(2).strip()))
    for k, lst in rows.items():
        if want and want not in k:
            continue
        for a, n, op, rest in lst:
            if op.startswith("s_cbranch") or op == "s_branch":
                imm = int(rest.split()[0])
                if imm >= 32768:
                    imm -= 65536
                tgt = a + 4 + imm * 4
                if tgt < a and a - tgt > 600:
                    seg = [r for r in lst if tgt <= r[0] <= a]
                    out.append(dict(kernel=k, start=tgt, end=a, instructions=len(seg), scalar=sum(1 for x in seg if x[2].startswith("s_")),
                                    wide=sum(1 for x in seg if x[1] >= 2), wide_off_grid=sum(1 for x in seg if x[1] >= 2 and x[0] % 8),
                                    row_head=sum(1 for x in seg if "row_bcast:31" in x[3])))     # > 0: the span holds a row's wave reductions — a path of the ROW loop, not a block loop
    return out


if __name__ == "__main__":
    args = sys.argv[1:]
    want = None
    if "--kernel" in args:
        i = args.index("--kernel")
        want = args[i + 1]
        del args[i:i + 2]
    show_loops = "--loops" in args
    if show_loops:
        args.remove("--loops")
    src, extra = args[0], args[1:]
    text = disassemble(src, extra)
    for k, v in audit(text, want).items():
        print(k[:70], v)
    if show_loops:
        for l in loops(text, want):
            print("  loop %x..%x: %d instructions (%d scalar), %d 64-bit, %d of them off the grid" % (
                l["start"], l["end"], l["instructions"], l["scalar"], l["wide"], l["wide_off_grid"]))
