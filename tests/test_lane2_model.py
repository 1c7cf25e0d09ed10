"""The two-seeds-per-lane algebra of bsw_lane2_kernel, verified on the CPU: tests/lane2_model.cpp drives the very
header the GPU kernel is compiled from (csrc/bsw_lane2_core.h) with the wave-level glue restated in loops, and must
agree with the oracle on every output of every seed — including the exact cell counts."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import _gen

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXTF = ["score", "qle", "tle", "gtle", "gscore", "max_off", "aw", "cells"]


@pytest.fixture(scope="module")
def model_lib(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("l2") / "lane2_model.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-Wall", "-I", os.path.join(ROOT, "include"),
                           "-o", so, os.path.join(ROOT, "tests", "lane2_model.cpp")])
    L = C.CDLL(so)
    L.lane2_model_run.restype = C.c_int
    L.lane2_model_run.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    L.lane2_model_run_qb.restype = C.c_int
    L.lane2_model_run_qb.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_int]
    L.lane2l_model_run.restype = C.c_int
    L.lane2l_model_run.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_int]
    L.lane2g_model_run.restype = C.c_int
    L.lane2g_model_run.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_int]
    return L


class Model:
    """unrolled / unrolled9: bsw_lane2_kernel<17> / <9> (blocks unrolled, row in VGPRs: the 136-column class and the
    72-column class of round 4 that runs at three waves per SIMD); loop17 / loop29: bsw_lane2l_kernel (blocks walked by a
    run-time loop, row behind an accessor) for the 136- and the 232-column class"""
    def __init__(self, lib, kind):
        self.lib, self.kind = lib, kind
        self.qcap = {"loop29": 231, "unrolled9": 71, "group4": 231}.get(kind, 135)

    def lane2_model_run(self, *a):
        if self.kind == "unrolled":
            return self.lib.lane2_model_run(*a)
        if self.kind == "unrolled9":
            return self.lib.lane2_model_run_qb(*a, 9)
        if self.kind in ("group3", "group4"):        # bsw_lane2g_kernel: a seed pair per group of eight lanes, 3 / 4 stripes of 64 columns
            return self.lib.lane2g_model_run(*a, 3 if self.kind == "group3" else 4)
        return self.lib.lane2l_model_run(*a, 29 if self.kind == "loop29" else 17)


@pytest.fixture(scope="module", params=["unrolled", "unrolled9", "loop17", "loop29", "group3", "group4"])
def model(model_lib, request):
    return Model(model_lib, request.param)


def run_side(model, host, p, tasks, side, h0s=None):
    qf = "rqlen" if side else "lqlen"
    idx = np.nonzero((tasks[qf] > 0) & (tasks[qf] <= model.qcap))[0]      # (the sides this class holds: the bins send the rest elsewhere)
    order = idx[np.argsort(-tasks[qf][idx], kind="stable")].astype(np.uint32)     # longest queries first, as the device bins
    out = np.zeros(len(tasks), dtype=host.EXT)
    rc = model.lane2_model_run(p.ctypes.data, tasks.ctypes.data, side, order.ctypes.data, len(order),
                               h0s.ctypes.data if h0s is not None else None, out.ctypes.data)
    assert rc == 0
    return out, idx


def oracle_side(host, oracle, p, tasks, side, h0s=None):
    et = np.zeros(len(tasks), dtype=host.EXT_TASK)
    pre = "r" if side else "l"
    et["query"], et["target"] = tasks[pre + "query"], tasks[pre + "target"]
    et["qlen"], et["tlen"] = tasks[pre + "qlen"], tasks[pre + "tlen"]
    et["w"], et["end_bonus"] = int(p["w"][0]), int(p["pen_clip3" if side else "pen_clip5"][0])
    et["h0"] = tasks["h0"] if h0s is None else h0s
    return oracle.ext_batch(p, et, nthreads=4)


def check(model, host, oracle, p, tasks):
    for side in (0, 1):
        got, idx = run_side(model, host, p, tasks, side)
        if len(idx) == 0:
            continue
        want = oracle_side(host, oracle, p, tasks, side)
        for f in EXTF:
            bad = np.nonzero(got[f][idx] != want[f][idx])[0]
            assert bad.size == 0, "side %d field %s: task %s got %s want %s" % (side, f, idx[bad[:4]], got[f][idx[bad[:4]]], want[f][idx[bad[:4]]])


@pytest.mark.parametrize("over", [
    dict(), dict(zdrop=0), dict(w=5), dict(w=17, zdrop=0), dict(w=1), dict(w=40, zdrop=30),
    dict(o_del=4, e_del=2, o_ins=4, e_ins=2), dict(o_del=0, e_del=1, o_ins=0, e_ins=1), dict(o_del=11, e_del=3, o_ins=11, e_ins=3, w=25),
    # variant M (bwa >= 0.7.9) and separate deletion / insertion penalties (the RTL's four-penalty datapath,
    # sw_pe_array_proc_element.v:816-819), alone and together
    dict(variant=1), dict(variant=1, zdrop=0), dict(variant=1, w=7), dict(variant=1, o_del=4, e_del=2, o_ins=4, e_ins=2, zdrop=40),
    dict(o_del=6, e_del=1, o_ins=4, e_ins=2), dict(o_del=2, e_del=3, o_ins=9, e_ins=1, zdrop=25), dict(o_del=0, e_del=2, o_ins=5, e_ins=1, w=12),
    dict(variant=1, o_del=6, e_del=1, o_ins=4, e_ins=2), dict(variant=1, o_del=3, e_del=2, o_ins=8, e_ins=1, zdrop=20, w=30),
    dict(variant=1, o_del=9, e_del=1, o_ins=1, e_ins=4, zdrop=0),
])
def test_random_seeds_match_the_oracle(model, host, oracle, over):
    rng = np.random.default_rng(abs(hash(str(sorted(over.items())))) % (2 ** 31))
    seeds = _gen.random_seeds(rng, 700, qmin=1, qmax=model.qcap, tfac=2.2, sub=0.03, indel=0.02, junk=0.15, nrate=0.004, h0max=60)
    for s in seeds:                                       # 8-bit score range of the kernel class
        tot = len(s.get("lq", ())) + len(s.get("rq", ()))
        s["h0"] = max(1, min(s["h0"], 255 - 4 - tot))     # the class bound: h0 + qlen*a + b <= 255
    tasks, arena = host.make_tasks(seeds)
    p = host.default_params(**over)
    check(model, host, oracle, p, tasks)


@pytest.mark.parametrize("variant", [0, 1])
@pytest.mark.parametrize("ab_n", [(1, 4, -1), (2, 3, -1), (1, 1, 0), (1, 6, -2), (3, 5, -5)])
def test_scoring_matrices(model, host, oracle, ab_n, variant):
    a, b, nn = ab_n
    rng = np.random.default_rng(a * 100 + b)
    seeds = _gen.random_seeds(rng, 400, qmin=1, qmax=model.qcap // a, tfac=2.0, sub=0.05, indel=0.02, junk=0.1, nrate=0.01, h0max=40)
    for s in seeds:
        tot = len(s.get("lq", ())) + len(s.get("rq", ()))
        s["h0"] = max(1, min(s["h0"], 255 - b - tot * a))
    tasks, arena = host.make_tasks(seeds)
    p = host.default_params(variant=variant)
    if variant:
        p["o_ins"], p["e_ins"] = 3, 2
    p["mat"][0] = host.bwa_matrix(a, b, nn)
    check(model, host, oracle, p, tasks)


@pytest.mark.parametrize("over", [dict(), dict(variant=1), dict(o_del=5, e_del=2, o_ins=7, e_ins=1), dict(variant=1, o_del=5, e_del=2, o_ins=7, e_ins=1)])
def test_bench_workload_shapes(model, host, oracle, over):
    """The synthetic workloads bench.py times: single bin (qlen 131 / tlen 257) and mixed PE bins with Ns."""
    p = host.default_params(**over)
    tasks, arena = host.synth_tasks(1500, seed=3)
    check(model, host, oracle, p, tasks)
    tasks, arena = host.synth_tasks(1500, seed=4, seed_len_min=19, seed_len_max=60, seed_at_start=0, junk_frac=0.1, n_rate=0.002)
    check(model, host, oracle, p, tasks)


def test_variants_differ_where_they_should(model, host, oracle):
    """Variant H lifts a zero H(i-1,j-1) by a match, variant M does not: on indel-rich seeds the two must disagree
    somewhere (or the M path is not exercised) and each must equal the oracle's own variant."""
    rng = np.random.default_rng(77)
    seeds = _gen.random_seeds(rng, 1500, qmin=20, qmax=model.qcap, tfac=2.2, sub=0.06, indel=0.05, junk=0.2, nrate=0.0, h0max=30)
    for s in seeds:
        tot = len(s.get("lq", ())) + len(s.get("rq", ()))
        s["h0"] = max(1, min(s["h0"], 255 - 4 - tot))
    tasks, arena = host.make_tasks(seeds)
    pH, pM = host.default_params(zdrop=0), host.default_params(zdrop=0, variant=1)
    gotH, idx = run_side(model, host, pH, tasks, 1)
    gotM, _ = run_side(model, host, pM, tasks, 1)
    assert gotH[idx].tobytes() != gotM[idx].tobytes()
    check(model, host, oracle, pH, tasks)
    check(model, host, oracle, pM, tasks)


def test_right_side_starts_from_the_left_score(model, host, oracle):
    """h0 of the right extension = score after the left one (sw_pe_array_proc_element.v:1671)."""
    p = host.default_params()
    tasks, arena = host.synth_tasks(900, seed=5, seed_len_min=19, seed_len_max=50, seed_at_start=0, indel_rate=0.01)
    full = oracle.pair_batch(p, tasks, nthreads=4)
    h0s = np.where(tasks["lqlen"] > 0, full["left"]["score"], tasks["h0"]).astype(np.int32)
    first_try = full["right"]["aw"] == int(p["w"][0])      # seeds whose right side needed no band retry
    got, idx = run_side(model, host, p, tasks, 1, h0s)
    sel = idx[first_try[idx]]
    for f in EXTF:
        assert (got[f][sel] == full["right"][f][sel]).all(), f


def test_ragged_and_tiny_waves(model, host, oracle):
    p = host.default_params()
    for n in (1, 2, 63, 64, 65, 127, 129, 200):
        tasks, arena = host.synth_tasks(n, seed=10 + n, seed_len_min=19, seed_len_max=60, seed_at_start=0, junk_frac=0.2, n_rate=0.003)
        check(model, host, oracle, p, tasks)


@pytest.mark.parametrize("ab", [(1, 4), (2, 3), (1, 1), (3, 7)])
def test_top_of_the_score_range(model, host, oracle, ab):
    """Perfect matches whose final score is EXACTLY the class bound 255 - b: the scaled cell forms H + a + b in the
    high byte of a 16-bit half before it subtracts b, so this is where an overflow would show."""
    a, b = ab
    rng = np.random.default_rng(a * 10 + b)
    seeds = []
    for k in range(300):
        lq = int(rng.integers(0, 60))
        rq = int(rng.integers(1, min(model.qcap, (255 - b - 1) // a - lq)))
        if model.qcap < 100 and k % 3 == 0 and (255 - b - 1) // a - lq > model.qcap:
            rq = model.qcap                                              # the class's widest row at the top of the range
        h0 = 255 - b - (lq + rq) * a - int(rng.integers(0, 2))          # top = 255 - b, or one below
        if h0 < 1:
            continue
        qL, qR = rng.integers(0, 4, lq).astype(np.uint8), rng.integers(0, 4, rq).astype(np.uint8)
        tL = np.concatenate([qL, rng.integers(0, 4, 20).astype(np.uint8)])  # the query matches its target end to end
        tR = np.concatenate([qR, rng.integers(0, 4, 20).astype(np.uint8)])
        seeds.append(dict(lq=qL, lt=tL, rq=qR, rt=tR, h0=h0))
    tasks, arena = host.make_tasks(seeds)
    p = host.default_params()
    p["mat"][0] = host.bwa_matrix(a, b, -1)
    full = oracle.pair_batch(p, tasks, nthreads=4)
    assert full["right"]["score"].max() == 255 - b                        # the bound is really reached
    h0s = np.where(tasks["lqlen"] > 0, full["left"]["score"], tasks["h0"]).astype(np.int32)
    got, idx = run_side(model, host, p, tasks, 0)
    want = oracle_side(host, oracle, p, tasks, 0)
    for f in EXTF:
        assert (got[f][idx] == want[f][idx]).all(), f
    got, idx = run_side(model, host, p, tasks, 1, h0s)
    want = oracle_side(host, oracle, p, tasks, 1, h0s)
    for f in EXTF:
        assert (got[f][idx] == want[f][idx]).all(), f


def test_generated_block_bodies_are_current():
    """bsw_lane2_body_asm.inc is the output of tools/gen_lane2_body.py: the committed file must be what the generator
    prints (a hand edit, or a generator change without regenerating, would put unreviewed asm into the kernels)."""
    import sys
    want = subprocess.check_output([sys.executable, os.path.join(ROOT, "tools", "gen_lane2_body.py")],
                                   env={k: v for k, v in os.environ.items() if not k.startswith("L2GEN_")}).decode()
    have = open(os.path.join(ROOT, "bwa-mem-sw_amd", "csrc", "bsw_lane2_body_asm.inc")).read()
    assert have == want
    # every variant: EDGE x NQ x VM x SYM interleaved bodies + EDGE x VM x SYM ragged bodies
    assert have.count("block8_asm<") >= 32 and have.count("block8_seq_asm<") >= 16
    # a packed result is never read by the very next instruction without a wait state (dst forwarding hazard)
    # (lines are written with the encoding macros L2E = _e32 / _e64, L2W = the wait state, L2A = the alignment)
    lines = [l.strip().replace('" L2E "', "_e32").replace("L2W ", '"s_nop 0" ').replace('" "', "").strip('"').replace("\\n\\t", "")
             for l in have.splitlines() if l.strip().startswith(('"', "L2W"))]
    assert sum(1 for l in lines if l == "s_nop 0") > 50
    for a, b in zip(lines, lines[1:]):
        if a.startswith("v_pk_") and not b.startswith("s_") and not b.endswith(":"):
            dst = a.split()[1].rstrip(",")
            srcs = b.split(None, 2)[2] if len(b.split(None, 2)) > 2 else ""
            assert dst not in [x.strip().rstrip(",") for x in srcs.replace(",", " ").split()], (a, b)
