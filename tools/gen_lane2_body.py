#!/usr/bin/env python3
"""Generates bwa-mem-sw_amd/csrc/bsw_lane2_body_asm.inc: the 8-column block bodies of the two-seeds-per-lane kernels as ONE
hand-scheduled inline-asm statement each.

The per-cell arithmetic is lane2::cell() of bsw_lane2_core.h (sw_pe_array_sw_extend.v:1797-1816,1863-1866; the CPU model of
the tests runs that C++ function, the GPU runs these statements: the GPU parity suite compares them with the oracle).  hipcc
sees every v_pk_* primitive of cell() as an opaque `asm` statement of unknown latency and orders the 128+ instructions of a
block almost serially (half of them directly behind their producer, each such pair with an s_nop for the packed-result
forwarding hazard).  With two waves per SIMD the partner wave covers that; at ONE wave per SIMD (the 232-column class)
nothing does.  Here the eight columns are list-scheduled together: priority = longest path to the end of the block, a
consumer is placed >= LAT issue slots behind its producer whenever anything else is ready, registers are assigned by linear
scan over the final order, and an s_nop 0 is emitted only where a packed result is read by the very next instruction.

Variants: EDGE (the block holds some seed's `end`: per-column masks), NQ (some query has an N in the block), VM (variant M),
SYM (one shared gap-open term).  Usage: gen_lane2_body.py > bsw_lane2_body_asm.inc"""
import itertools, re
import sys

import os
LAT_PK = int(os.environ.get("L2GEN_LAT_PK", "3"))      # issue slots between a VOP3/VOP3P producer and its consumer (dependent issue ~9 cycles, slot ~4.4)
LAT_V2 = int(os.environ.get("L2GEN_LAT_V2", "2"))      # plain VOP2


class Op:
    def __init__(self, name, fmt, dst, srcs, packed=True, lat=None):
        self.name, self.fmt, self.dst, self.srcs, self.packed = name, fmt, dst, srcs, packed
        self.lat = lat if lat is not None else (LAT_PK if packed else LAT_V2)
        self.preds, self.succs = set(), set()


def build(edge, nq, vm, sym):
    """The ops of one 8-column block in program order (SSA value names), plus which values are inputs / outputs."""
    ops = []
    ins = {"Wc", "B", "HI", "h1_in", "f_in"} | {"P%d" % c for c in range(8)}
    if nq:
        ins |= {"WN", "D2"}
    if edge:
        ins |= {"END", "mi_in"}

    def op(name, fmt, dst, srcs, **kw):
        ops.append(Op(name, fmt, dst, srcs, **kw))
        return dst

    f, h1, mi_prev = "f_in", "h1_in", "mi_in"
    mk = nz = None
    for c in range(8):
        P = "P%d" % c
        bit = (1 << c) * 0x00010001
        jj = "0" if c == 0 else "%%[JJ%d]" % c
        t = op("t", "v_and_b32_e32 {d}, 0x%x, {0}" % bit, "t%d" % c, ["Wc"], packed=False)
        hd = op("hd", "v_pk_lshlrev_b16 {d}, 8, {0} op_sel_hi:[0,1]", "hd%d" % c, [P])
        X = op("X", "v_pk_mad_u16 {d}, {0}, %%[MC%d], {1}" % c, "X%d" % c, [t, hd])
        if nq:
            if c:
                n0 = op("n0", "v_lshrrev_b32_e32 {d}, %d, {0}" % c, "ns%d" % c, ["WN"], packed=False)
            else:
                n0 = "WN"
            n1 = op("n1", "v_and_b32_e32 {d}, 0x10001, {0}", "n%d" % c, [n0], packed=False)
            X = op("Xn", "v_pk_mad_u16 {d}, {0}, {1}, {2}", "Xn%d" % c, [n1, "D2", X])
        M = op("M", "v_pk_sub_u16 {d}, {0}, {1} clamp", "M%d" % c, [X, "B"])
        if vm:
            z = op("z", "v_pk_mad_u16 {d}, {0}, -1, 0 op_sel_hi:[1,0,0] clamp", "z%d" % c, [hd])
            M = op("Mz", "v_and_b32_e32 {d}, {0}, {1}", "Mz%d" % c, [M, z], packed=False)
        me = op("me", "v_pk_max_u16 {d}, {0}, {1}", "me%d" % c, [M, P])
        h = op("h", "v_pk_max_u16 {d}, {0}, {1}", "h%d" % c, [me, f])
        g = M if vm else h
        tD = op("tD", "v_pk_sub_u16 {d}, {0}, %[OED] clamp", "tD%d" % c, [g])
        tI = tD if sym else op("tI", "v_pk_sub_u16 {d}, {0}, %[OEI] clamp", "tI%d" % c, [g])
        es = op("es", "v_pk_sub_u16 {d}, {0}, %[ED] clamp", "es%d" % c, [P])
        en = op("en", "v_pk_max_u16 {d}, {0}, {1}", "en%d" % c, [es, tD])
        fs = op("fs", "v_pk_sub_u16 {d}, {0}, %%[%s] clamp" % ("ED" if sym else "EI"), "fs%d" % c, [f])
        f = op("f", "v_pk_max_u16 {d}, {0}, {1}", "f%d" % c, [fs, tI])
        if not edge:
            key = op("key", "v_and_or_b32 {d}, {0}, {1}, %s" % jj, "key%d" % c, [h, "HI"])
            mk = key if c == 0 else op("mk", "v_pk_max_u16 {d}, {0}, {1}", "mk%d" % c, [mk, key])
            np_ = op("np", "v_perm_b32 {d}, {0}, {1}, %[PERM]", "np%d" % c, [en, h1])
            nb = op("nb", "v_pk_min_u16 {d}, {0}, %[ONE]", "nb%d" % c, [np_])
            nz = nb if c == 0 else op("nz", "v_lshl_or_b32 {d}, {0}, %d, {1}" % c, "nz%d" % c, [nb, nz])
            newP, h1 = np_, h
        else:
            d = "END" if c == 0 else op("d", "v_pk_sub_u16 {d}, {0}, %s clamp" % jj, "d%d" % c, ["END"])
            mi = op("mi", "v_pk_mad_u16 {d}, {0}, -1, 0 op_sel_hi:[1,0,0] clamp", "mi%d" % c, [d])
            mw = mi_prev
            mi_prev = mi
            hm = op("hm", "v_and_b32_e32 {d}, {0}, {1}", "hm%d" % c, [h, mi], packed=False)
            key = op("key", "v_and_or_b32 {d}, {0}, {1}, %s" % jj, "key%d" % c, [hm, "HI"])
            mk = key if c == 0 else op("mk", "v_pk_max_u16 {d}, {0}, {1}", "mk%d" % c, [mk, key])
            enm = op("enm", "v_and_b32_e32 {d}, {0}, {1}", "enm%d" % c, [en, mi], packed=False)
            np0 = op("np0", "v_perm_b32 {d}, {0}, {1}, %[PERM]", "npr%d" % c, [enm, h1])
            np_ = op("np", "v_and_b32_e32 {d}, {0}, {1}", "np%d" % c, [np0, mw], packed=False)
            nb = op("nb", "v_pk_min_u16 {d}, {0}, %[ONE]", "nb%d" % c, [np_])
            nz = nb if c == 0 else op("nz", "v_lshl_or_b32 {d}, {0}, %d, {1}" % c, "nz%d" % c, [nb, nz])
            newP = op("Pn", "v_bfi_b32 {d}, {0}, {1}, {2}", "Pn%d" % c, [mw, np_, P])
            h1 = op("h1", "v_bfi_b32 {d}, {0}, {1}, {2}", "h1n%d" % c, [mi, h, h1])
        ops[-1].__dict__.setdefault("x", None)
        # remember which SSA value becomes the new P of this column
        build.newP[c] = newP
    outs = {"h1": h1, "f": f, "mk": mk, "nz": nz}
    for c in range(8):
        outs["P%d" % c] = build.newP[c]
    return ops, ins, outs


build.newP = {}


def schedule(ops, ins, outs):
    producer = {o.dst: o for o in ops}
    readers = {}
    for o in ops:
        for s in o.srcs:
            readers.setdefault(s, []).append(o)
    for o in ops:
        for s in o.srcs:
            if s in producer:
                o.preds.add(producer[s]); producer[s].succs.add(o)
    # anti-dependencies: an output value is written into its operand register, whose old content (the matching input)
    # must have been read by then
    reg_in = {"h1": "h1_in", "f": "f_in"}
    reg_in.update({"P%d" % c: "P%d" % c for c in range(8)})
    for r, v in outs.items():
        if r in reg_in and v in producer:
            for rd in readers.get(reg_in[r], []):
                if rd is not producer[v]:
                    producer[v].preds.add(rd); rd.succs.add(producer[v])
    # priority: longest latency-weighted path to the end
    prio = {}
    for o in reversed(ops):
        prio[o] = o.lat + max([prio[s] for s in o.succs], default=0)
    done, order, slot = {}, [], 0
    remaining = list(ops)
    while remaining:
        ready = [o for o in remaining if all(p in done for p in o.preds)]
        def wait(o):
            return max([done[p] + p.lat - slot for p in o.preds if p.dst in o.srcs] + [0])
        ok = [o for o in ready if wait(o) <= 0]
        pick = max(ok, key=lambda o: prio[o]) if ok else min(ready, key=lambda o: (wait(o), -prio[o]))
        # a packed result read by the very next instruction: one wait state (dst forwarding hazard)
        if order and order[-1] != "nop" and order[-1].packed and order[-1].dst in pick.srcs:
            order.append("nop"); slot += 1
        order.append(pick); done[pick] = slot; slot += 1
        remaining.remove(pick)
    return order


def allocate(order, ins, outs):
    """Linear scan over the final order: SSA value -> asm operand name."""
    real = [o for o in order if o != "nop"]
    last_use = {}
    for i, o in enumerate(real):
        for s in o.srcs:
            last_use[s] = i
    out_of = {v: r for r, v in outs.items()}
    reg_in = {"h1_in": "h1", "f_in": "f"}
    loc = {}
    for v in sorted(ins):
        loc[v] = reg_in.get(v, v)                       # input operands keep their own names (P0.., Wc, B, ...)
    busy = {loc[v]: v for v in ins}
    free_tmp, ntmp = [], 0
    inplace = {"mk", "nz"}
    for i, o in enumerate(real):
        # registers whose value dies at this instruction are free for its destination (read-before-write in one op)
        for s in sorted(set(o.srcs)):                    # (sorted: the output must not depend on hash order)
            if last_use.get(s) == i and s not in outs.values():
                r = loc[s]
                if busy.get(r) == s:
                    del busy[r]
                    if r.startswith("T"):
                        free_tmp.append(r)
        v = o.dst
        want = out_of.get(v)
        if want is None and v[:2] in ("mk", "nz") and v[2:].isdigit():
            want = v[:2]                                 # the mk / nz chains run in place in their output registers
        if want is not None and want not in busy:
            r = want
        elif want is not None:
            raise SystemExit("output register %s still busy with %s when %s is defined" % (want, busy[want], v))
        else:
            if free_tmp:
                r = free_tmp.pop()
            else:
                r = "T%d" % ntmp; ntmp += 1
        loc[v] = r
        busy[r] = v
    return loc, ntmp


def on_grid(lines):
    """Two encodings of one body, chosen where the .inc is included (BSW_L2_GRID):
    * on the grid, for the looped kernel (one wave per SIMD): every instruction as a 64-bit encoding and the statement
      aligned to 8 bytes — a 64-bit instruction that starts 4 (mod 8) is dearer to fetch (tools/isa_align.py; one dword of
      shift cost that kernel 7 %).  VOP1/VOP2 forms without a literal take their VOP3 (_e64) encoding, the wait state is
      a 64-bit v_nop, scalar instructions come in pairs (s_cmp + s_cbranch);
    * compact, for the unrolled kernel (two waves per SIMD hide the fetch, the instruction cache is what it is short of,
      and a v_nop would take a VALU slot from the other wave): _e32 forms, s_nop, no alignment.
    The choice is made by three string macros the lines are written with: L2A, L2E, L2W."""
    out, nsalu = ["\x03"], 0
    for l in lines:
        if l.endswith(":"):
            assert nsalu % 2 == 0
            out.append(l)
            continue
        if l == "s_nop 0":
            out.append("\x02")
            continue
        if l.startswith("s_"):
            nsalu += 1
            out.append(l)
            continue
        assert nsalu % 2 == 0, l
        m = l.split()[0]
        if m.endswith("_e32"):
            lit = any(re.fullmatch(r"0x[0-9a-f]+", t) and int(t, 16) > 64 for t in l.replace(",", " ").split()[1:])
            if not lit:
                l = l.replace("_e32", "\x01", 1)
        out.append(l)
    assert nsalu % 2 == 0
    return out


def c_string(l, last=False):
    """One asm line as a C string literal (the encoding macros spliced in)."""
    end = "" if last else "\\n\\t"
    if l == "\x03":
        return "L2A"
    if l == "\x02":
        return 'L2W "%s"' % end
    if "\x01" in l:
        h, t = l.split("\x01")
        return '"%s" L2E "%s%s"' % (h, t, end)
    return '"%s%s"' % (l, end)


def emit(edge, nq, vm, sym):
    ops, ins, outs = build(edge, nq, vm, sym)
    order = schedule(ops, ins, outs)
    # key0 / nb0 are the first elements of the mk / nz chains: give them the chain's register
    ren = {}
    loc, ntmp = allocate(order, ins, outs)
    lines = []
    for o in order:
        if o == "nop":
            lines.append("s_nop 0")
            continue
        d = "%%[%s]" % loc[o.dst]
        srcs = ["%%[%s]" % loc[s] for s in o.srcs]
        lines.append(o.fmt.replace("{d}", d).format(*srcs) if "{0}" in o.fmt else o.fmt.replace("{d}", d))
    nvalu = sum(1 for o in order if o != "nop")
    nnop = sum(1 for o in order if o == "nop")
    lines = on_grid(lines)
    name = "block8_asm"
    sig = ["uint32_t (&P)[8]", "uint32_t Wc"]
    if nq:
        sig += ["uint32_t WN", "uint32_t D2"]
    sig += ["uint32_t B", "const consts &k"]
    if edge:
        sig += ["uint32_t END", "uint32_t mi_in"]
    sig += ["uint32_t &h1", "uint32_t &f", "uint32_t &mk", "uint32_t &nz"]
    outs_c = ['[P%d] "+v"(P[%d])' % (c, c) for c in range(8)] + ['[h1] "+v"(h1)', '[f] "+v"(f)', '[mk] "=&v"(mk)', '[nz] "=&v"(nz)']
    outs_c += ['[T%d] "=&v"(t%d)' % (i, i) for i in range(ntmp)]
    ins_c = ['[Wc] "v"(Wc)', '[B] "v"(B)', '[HI] "v"(k.HI2)']
    if nq:
        ins_c += ['[WN] "v"(WN)', '[D2] "v"(D2)']
    if edge:
        ins_c += ['[END] "v"(END)', '[mi_in] "v"(mi_in)']
    ins_c += ['[MC%d] "s"(k.MC[%d])' % (c, c) for c in range(8)]
    ins_c += ['[JJ%d] "s"(0x%xu)' % (c, c * 0x00010001) for c in range(1, 8)]
    ins_c += ['[OED] "s"(k.OED2s)', '[ED] "s"(k.ED2s)', '[ONE] "s"(k.ONE2)', '[PERM] "s"(0x07030501u)']
    if not sym:
        ins_c += ['[OEI] "s"(k.OEI2s)', '[EI] "s"(k.EI2s)']
    body = []
    body.append("/* EDGE=%d NQ=%d VM=%d SYM=%d: %d instructions + %d s_nop, %d temporaries */" % (edge, nq, vm, sym, nvalu, nnop, ntmp))
    body.append("__device__ __forceinline__ void %s<%s, %s, %s, %s>::run(%s)" % (
        name, *("true" if x else "false" for x in (edge, nq, vm, sym)), ", ".join(sig)))
    body.append("{")
    if ntmp:
        body.append("    uint32_t %s;" % ", ".join("t%d" % i for i in range(ntmp)))
    body.append("    asm volatile(")
    for i, l in enumerate(lines):
        body.append("        " + c_string(l, i == len(lines) - 1))
    body.append("        : " + ", ".join(outs_c))
    body.append("        : " + ", ".join(ins_c) + ");")
    body.append("}")
    return "\n".join(body), (nvalu, nnop, ntmp)


def emit_seq(edge, vm, sym):
    """The ragged first (dense) / last (edge) block of a row: columns in order, a scalar guard between them — entry guards
    (skip column c while c < first) for the dense body, exit guards (leave once c > last) for the edge body.  Everything
    that flows from column to column sits in a fixed register (h1, f, mk, nz, the edge mask); temporaries are per column."""
    lines, ntmp_max = [], 0
    lines.append("v_mov_b32_e32 %[mk], 0")
    lines.append("v_mov_b32_e32 %[nz], 0")
    for c in range(8):
        if edge and c:
            lines += ["s_cmp_gt_i32 %d, %%[G]" % c, "s_cbranch_scc1 9f"]           # c > last: done
        if not edge and c < 7:
            lines += ["s_cmp_lt_i32 %d, %%[G]" % c, "s_cbranch_scc1 %df" % (10 + c)]   # c < first: next column
        P = "%%[P%d]" % c
        bit = (1 << c) * 0x00010001
        jj = "0" if c == 0 else "%%[JJ%d]" % c
        T = lambda i: "%%[T%d]" % i
        col = []
        # (temporaries: T0 t/X/M/h-chain, T1 hd, T2 tD, T3 es/en, T4 fs/nb, T5 key (tI before it), T6 edge mask)
        # One column is a serial chain (hd -> X -> M -> me -> h -> tD -> en); what does not depend on it is placed INTO its
        # gaps (a packed result read by the next instruction costs a wait state, which at one wave per SIMD is an issue slot)
        col.append("v_and_b32_e32 %s, 0x%x, %%[Wc]" % (T(0), bit))
        col.append("v_pk_lshlrev_b16 %s, 8, %s op_sel_hi:[0,1]" % (T(1), P))
        col.append("v_pk_sub_u16 %s, %s, %%[ED] clamp" % (T(3), P))                                   # es
        col.append("v_pk_mad_u16 %s, %s, %%[MC%d], %s" % (T(0), T(0), c, T(1)))
        if edge:
            if c:
                col.append("v_pk_sub_u16 %s, %%[END], %s clamp" % (T(6), jj))
        else:
            col.append("v_pk_sub_u16 %s, %%[f], %%[%s] clamp" % (T(4), "ED" if sym else "EI"))      # fs
        col.append("v_pk_sub_u16 %s, %s, %%[B] clamp" % (T(0), T(0)))                                  # M
        if edge:
            col.append("v_pk_mad_u16 %s, %s, -1, 0 op_sel_hi:[1,0,0] clamp" % (T(6), T(6) if c else "%[END]"))   # mi
        if vm:
            col.append("v_pk_mad_u16 %s, %s, -1, 0 op_sel_hi:[1,0,0] clamp" % (T(1), T(1)))
            if edge:
                col.append("v_pk_sub_u16 %s, %%[f], %%[%s] clamp" % (T(4), "ED" if sym else "EI"))  # fs
            col.append("v_and_b32_e32 %s, %s, %s" % (T(0), T(0), T(1)))
            col.append("v_pk_sub_u16 %s, %s, %%[OED] clamp" % (T(2), T(0)))                            # gaps open from M
            if not sym:
                col.append("v_pk_sub_u16 %s, %s, %%[OEI] clamp" % (T(5), T(0)))
        elif edge:
            col.append("v_pk_sub_u16 %s, %%[f], %%[%s] clamp" % (T(4), "ED" if sym else "EI"))      # fs
        col.append("v_pk_max_u16 %s, %s, %s" % (T(0), T(0), P))                                        # me
        col.append("v_pk_max_u16 %s, %s, %%[f]" % (T(0), T(0)))                                        # h
        if not vm:
            col.append("v_pk_sub_u16 %s, %s, %%[OED] clamp" % (T(2), T(0)))
            if not sym:
                col.append("v_pk_sub_u16 %s, %s, %%[OEI] clamp" % (T(5), T(0)))
        if not edge:
            if sym:
                col.append("v_mov_b32_e32 %s, %%[h1]" % T(6))                                          # H(i,j-1) for the stored pair
                col.append("v_mov_b32_e32 %%[h1], %s" % T(0))
            col.append("v_pk_max_u16 %s, %s, %s" % (T(3), T(3), T(2)))                                 # en
            col.append("v_pk_max_u16 %%[f], %s, %s" % (T(4), T(2) if sym else T(5)))                   # f
            col.append("v_and_or_b32 %s, %s, %%[HI], %s" % (T(5), T(0), jj))                           # key
            if sym:
                col.append("v_perm_b32 %s, %s, %s, %%[PERM]" % (P, T(3), T(6)))
            else:
                col.append("v_perm_b32 %s, %s, %%[h1], %%[PERM]" % (P, T(3)))
                col.append("v_mov_b32_e32 %%[h1], %s" % T(0))
            col.append("v_pk_max_u16 %%[mk], %%[mk], %s" % T(5))
            col.append("v_pk_min_u16 %s, %s, %%[ONE]" % (T(4), P))
            col.append("v_lshl_or_b32 %%[nz], %s, %d, %%[nz]" % (T(4), c))
        else:
            # T6 = mi (c < end), %[mi] = mw (c <= end) from the previous column
            if sym:
                col.append("v_and_b32_e32 %s, %s, %s" % (T(5), T(0), T(6)))                             # h & mi
            col.append("v_pk_max_u16 %s, %s, %s" % (T(3), T(3), T(2)))                                 # en
            col.append("v_pk_max_u16 %%[f], %s, %s" % (T(4), T(2) if sym else T(5)))                   # f
            if not sym:
                col.append("v_and_b32_e32 %s, %s, %s" % (T(5), T(0), T(6)))                             # h & mi (T5 held tI until f)
            col.append("v_and_or_b32 %s, %s, %%[HI], %s" % (T(5), T(5), jj))                           # key
            col.append("v_and_b32_e32 %s, %s, %s" % (T(3), T(3), T(6)))                                 # en & mi
            col.append("v_pk_max_u16 %%[mk], %%[mk], %s" % T(5))
            col.append("v_perm_b32 %s, %s, %%[h1], %%[PERM]" % (T(3), T(3)))
            col.append("v_bfi_b32 %%[h1], %s, %s, %%[h1]" % (T(6), T(0)))
            col.append("v_and_b32_e32 %s, %s, %%[mi]" % (T(3), T(3)))
            col.append("v_pk_min_u16 %s, %s, %%[ONE]" % (T(4), T(3)))
            col.append("v_bfi_b32 %s, %%[mi], %s, %s" % (P, T(3), P))
            col.append("v_lshl_or_b32 %%[nz], %s, %d, %%[nz]" % (T(4), c))
            col.append("v_mov_b32_e32 %%[mi], %s" % T(6))
        lines += [l for l in col if l != "s_nop 0"]
        if not edge and c < 7:
            lines.append("%d:" % (10 + c))
    lines.append("9:")
    # one wait state wherever a packed (VOP3P) result is read by the very next instruction (dst forwarding hazard);
    # a label does not separate two instructions, a scalar instruction does
    fixed, prev = [], None
    for l in lines:
        if l.endswith(":"):
            fixed.append(l)
            prev = None                                   # (a join: the predecessor is not known statically; see below)
            continue
        ops = l.replace(",", " ").split()
        if prev is not None and prev[0].startswith("v_pk_") and l.startswith("v_") and prev[1] in ops[2:]:
            fixed.append("s_nop 0")
        fixed.append(l)
        prev = ops
    # at a label the fall-through predecessor is the instruction above it: check that pair too
    out2 = []
    for i, l in enumerate(fixed):
        if l.endswith(":") and i > 0 and i + 1 < len(fixed):
            a, b = fixed[i - 1].replace(",", " ").split(), fixed[i + 1].replace(",", " ").split()
            if a[0].startswith("v_pk_") and fixed[i + 1].startswith("v_") and a[1] in b[2:]:
                out2.append("s_nop 0")
        out2.append(l)
    lines = on_grid(out2)
    ntmp = 7
    sig = ["uint32_t (&P)[8]", "uint32_t Wc", "uint32_t B", "const consts &k"]
    if edge:
        sig += ["uint32_t END", "uint32_t mi_in"]
    sig += ["int guard", "uint32_t &h1", "uint32_t &f", "uint32_t &mk", "uint32_t &nz"]
    outs_c = ['[P%d] "+v"(P[%d])' % (c, c) for c in range(8)] + ['[h1] "+v"(h1)', '[f] "+v"(f)', '[mk] "=&v"(mk)', '[nz] "=&v"(nz)']
    if edge:
        outs_c += ['[mi] "+v"(mi)']
    outs_c += ['[T%d] "=&v"(t%d)' % (i, i) for i in range(ntmp)]
    ins_c = ['[Wc] "v"(Wc)', '[B] "v"(B)', '[HI] "v"(k.HI2)', '[G] "s"(guard)']
    if edge:
        ins_c += ['[END] "v"(END)']
    ins_c += ['[MC%d] "s"(k.MC[%d])' % (c, c) for c in range(8)]
    ins_c += ['[JJ%d] "s"(0x%xu)' % (c, c * 0x00010001) for c in range(1, 8)]
    ins_c += ['[OED] "s"(k.OED2s)', '[ED] "s"(k.ED2s)', '[ONE] "s"(k.ONE2)', '[PERM] "s"(0x07030501u)']
    if not sym:
        ins_c += ['[OEI] "s"(k.OEI2s)', '[EI] "s"(k.EI2s)']
    body = ["/* ragged %s block, VM=%d SYM=%d: columns in order behind scalar guards */" % ("last (edge)" if edge else "first (dense)", vm, sym)]
    body.append("__device__ __forceinline__ void block8_seq_asm<%s, %s, %s>::run(%s)" % (*("true" if x else "false" for x in (edge, vm, sym)), ", ".join(sig)))
    body.append("{")
    body.append("    uint32_t %s;" % ", ".join("t%d" % i for i in range(ntmp)))
    if edge:
        body.append("    uint32_t mi = mi_in;")
    body.append("    asm volatile(")
    for i, l in enumerate(lines):
        body.append("        " + c_string(l, i == len(lines) - 1))
    body.append("        : " + ", ".join(outs_c))
    body.append("        : " + ", ".join(ins_c) + " : \"scc\");")
    body.append("}")
    return "\n".join(body)


def main():
    print("/* GENERATED by tools/gen_lane2_body.py — do not edit.  The 8-column block bodies of the two-seeds-per-lane kernels as")
    print(" * one list-scheduled inline-asm statement per variant; arithmetic = lane2::cell() of bsw_lane2_core.h. */")
    print("#if defined(BSW_L2_GRID)")
    print('#define L2A ".p2align 3\\n\\t"')
    print('#define L2E "_e64"')
    print('#define L2W "v_nop_e64"')
    print("#else")
    print('#define L2A ""')
    print('#define L2E "_e32"')
    print('#define L2W "s_nop 0"')
    print("#endif")
    print("template <bool EDGE, bool NQ, bool VM, bool SYM> struct block8_asm;")
    for edge, nq, vm, sym in itertools.product((0, 1), repeat=4):
        sig = ["uint32_t (&P)[8]", "uint32_t Wc"]
        if nq:
            sig += ["uint32_t WN", "uint32_t D2"]
        sig += ["uint32_t B", "const consts &k"]
        if edge:
            sig += ["uint32_t END", "uint32_t mi_in"]
        sig += ["uint32_t &h1", "uint32_t &f", "uint32_t &mk", "uint32_t &nz"]
        print("template <> struct block8_asm<%s, %s, %s, %s> { static __device__ __forceinline__ void run(%s); };" % (
            *("true" if x else "false" for x in (edge, nq, vm, sym)), ", ".join(sig)))
    stats = {}
    for edge, nq, vm, sym in itertools.product((0, 1), repeat=4):
        text, st = emit(edge, nq, vm, sym)
        stats[(edge, nq, vm, sym)] = st
        print(text)
    print("template <bool EDGE, bool VM, bool SYM> struct block8_seq_asm;")
    for edge, vm, sym in itertools.product((0, 1), repeat=3):
        sig = ["uint32_t (&P)[8]", "uint32_t Wc", "uint32_t B", "const consts &k"]
        if edge:
            sig += ["uint32_t END", "uint32_t mi_in"]
        sig += ["int guard", "uint32_t &h1", "uint32_t &f", "uint32_t &mk", "uint32_t &nz"]
        print("template <> struct block8_seq_asm<%s, %s, %s> { static __device__ __forceinline__ void run(%s); };" % (
            *("true" if x else "false" for x in (edge, vm, sym)), ", ".join(sig)))
    for edge, vm, sym in itertools.product((0, 1), repeat=3):
        print(emit_seq(edge, vm, sym))
    print("/* instruction counts (EDGE, NQ, VM, SYM) -> (instructions, s_nop, temporaries): %s */" % stats)


if __name__ == "__main__":
    main()
