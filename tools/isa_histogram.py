#!/usr/bin/env python3
"""ISA histogram of the dense 8-column block of bsw_lane2_kernel and the opcode-weighted VALU roof it implies.

gfx950 issues a wave64 VALU instruction in 2 cycles per SIMD for the plain VOP1/VOP2 integer ops (v_add/sub/and/or/xor/
lshr/ashr, 16-bit VOP2) and in 4 cycles for everything else the DP cell needs (v_max/min, v_lshl*, every VOP3 / VOP3P /
SDWA / DPP encoding) — measured in profiles/r1/ubench_valu_rate_* and profiles/r2/ubench3_valu_rates.txt.  The roof
for THIS instruction mix: the dense block issued back to back at those nominal cycles on all 1024 SIMDs at 2.4 GHz.
Usage: isa_histogram.py [lane2.s]   (compiles csrc/bsw_lane2_kernel.hip to ISA when no file is given)"""
import collections, json, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAST = {"v_add_u32_e32", "v_sub_u32_e32", "v_subrev_u32_e32", "v_and_b32_e32", "v_or_b32_e32", "v_xor_b32_e32", "v_lshrrev_b32_e32",
        "v_ashrrev_i32_e32", "v_mov_b32_e32", "v_add_u16_e32", "v_sub_u16_e32", "v_max_u16_e32", "v_max_i16_e32", "v_min_u16_e32",
        "v_lshrrev_b16_e32", "v_lshlrev_b16_e32", "v_mul_lo_u16_e32", "v_bitop3_b32"}
def main():
    if len(sys.argv) > 1:
        text = open(sys.argv[1]).read()
    else:
        out = os.path.join(tempfile.mkdtemp(), "lane2.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-DBSW_L2_ASM_BODY=1", "-I", os.path.join(ROOT, "include"), "-o", out,
                               os.path.join(ROOT, "bwa-mem-sw_amd", "csrc", "bsw_lane2_kernel.hip")], stderr=subprocess.DEVNULL)
        text = open(out).read()
    lines = text.split("\n")
    k0 = [i for i, l in enumerate(lines) if l.startswith("_ZN3bsw16bsw_lane2_kernelILi17ELi2ELb0ELb1E")][0]   # the headline instantiation
    hdr = [i for i, l in enumerate(lines) if i > k0 and "Loop Header: Depth=1" in l][0]
    blocks, cur = [], None
    for i in range(hdr, len(lines)):
        l = lines[i]
        if re.match(r"^\.LBB", l) or re.match(r"^; %bb\.", l):
            cur = [i, []]; blocks.append(cur)
        elif cur is not None:
            m = re.match(r"\s+([vs]_\S+)", l)
            if m: cur[1].append(m.group(1))
        if "codeLenInByte" in l: break
    valu = lambda ops: [o for o in ops if o.startswith("v_")]
    # the block bodies are one asm statement each (bsw_lane2_body_asm.inc), so a basic block that holds a body holds the whole
    # path of one 8-column block: the match-byte extraction in front, the body, the folds behind.  Four variants per column
    # block (dense / with query Ns / edge / edge with Ns); the dense one is the smallest and by far the most frequent.
    cnt = collections.Counter(len(valu(b[1])) for b in blocks if len(valu(b[1])) > 100)
    nd = min(sz for sz, c in cnt.items() if c >= 8)
    di = [k for k, b in enumerate(blocks) if len(valu(b[1])) == nd][3]
    ops = blocks[di][1]
    ext = mrg = di
    has_mrg = False
    hist = collections.Counter(ops)
    v = valu(ops)
    cyc = sum(2 if o in FAST else 4 for o in v)
    pair_cells = 8
    cells_per_cycle_per_simd = 2 * 64 * pair_cells / cyc                 # two seeds per lane
    peak_gcups = cells_per_cycle_per_simd * 1024 * 2.4
    res = {"kernel": "bsw_lane2_kernel<17,2>", "dense_path_valu_insts": len(v), "valu_insts_per_pair_cell": round(len(v) / pair_cells, 2),
           "full_rate_insts": sum(1 for o in v if o in FAST), "half_rate_insts": sum(1 for o in v if o not in FAST),
           "s_nop": hist.get("s_nop", 0), "salu_other": sum(c for o, c in hist.items() if o.startswith("s_") and o != "s_nop"),
           "nominal_cycles_per_block": cyc, "peak_gcups_dense_body_back_to_back": round(peak_gcups, 1),
           "peak_opcode_weighted_tops": round(peak_gcups * 15 / 1000, 2), "histogram": dict(hist.most_common())}
    print(json.dumps(res))
if __name__ == "__main__":
    main()
