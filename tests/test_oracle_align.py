"""ksw_align2 oracle (oracle/ksw_align_ref.c, the literal emulation of bwa's striped local alignment): analytic known
answers, agreement with an independent numpy local DP, and re-scoring of the reported end points.  CPU only."""
import os
import sys

import numpy as np
import pytest

import _gen

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle", "py"))
import full_dp  # noqa: E402


def bwa_mat(a=1, b=4, n=-1):
    m = np.full((5, 5), -b, dtype=np.int8)
    np.fill_diagonal(m, a)
    m[4, :] = n
    m[:, 4] = n
    return m


XB, XSTOP, XSUBO, XSTART = 0x10000, 0x20000, 0x40000, 0x80000


@pytest.mark.parametrize("byte", [True, False])
def test_known_answers(oracle, byte):
    rng = np.random.default_rng(3)
    mat = bwa_mat()
    xb = XB if byte else 0
    q = rng.integers(0, 4, 60).astype(np.uint8)
    t = np.concatenate([rng.integers(0, 4, 100), q, rng.integers(0, 4, 80)]).astype(np.uint8)
    r = oracle.align2(q, t, mat, 6, 1, 6, 1, xb | XSTART)
    assert (r["score"], r["te"], r["qe"], r["tb"], r["qb"]) == (60, 159, 59, 100, 0)
    # three mismatches inside: 60 - 3*(1+4) = 45 for the whole read is beaten by the best clipped piece or not; re-derive with numpy
    q2 = q.copy()
    q2[[10, 30, 50]] = (q2[[10, 30, 50]] + 1) % 4
    r = oracle.align2(q2, t, mat, 6, 1, 6, 1, xb | XSTART)
    want = full_dp.local_dp(q2, t, mat, 6, 1, 6, 1)
    assert r["score"] == want["score"] == 45 and (r["te"], r["qe"]) in want["ends"]
    # one base deleted from the read: 59 matches and a 1-base gap along the query
    q3 = np.delete(q, 25)
    r = oracle.align2(q3, t, mat, 6, 1, 6, 1, xb | XSTART)
    assert r["score"] == 59 - 7 and (r["te"], r["qe"], r["tb"], r["qb"]) == (159, 58, 100, 0)
    # nothing aligns: score 0, te -1, qe 0 (first maximum of an all-zero column), start pass gives 0/0
    r = oracle.align2(np.zeros(20, np.uint8), np.full(50, 3, np.uint8), mat, 6, 1, 6, 1, xb | XSTART)
    assert (r["score"], r["te"], r["qe"], r["score2"], r["te2"]) == (0, -1, 0, -1, -1)
    # empty target
    r = oracle.align2(q, np.zeros(0, np.uint8), mat, 6, 1, 6, 1, xb)
    assert (r["score"], r["te"]) == (0, -1)


def test_byte_version_saturates_and_word_version_does_not(oracle):
    rng = np.random.default_rng(4)
    mat = bwa_mat()
    q = rng.integers(0, 4, 255).astype(np.uint8)
    t = np.concatenate([rng.integers(0, 4, 30), q, rng.integers(0, 4, 30)]).astype(np.uint8)
    assert oracle.align2(q, t, mat, 6, 1, 6, 1, XB)["score"] == 255                 # gmax + shift reached 255: "try 16 bits"
    r = oracle.align2(q, t, mat, 6, 1, 6, 1, XSTART)
    assert (r["score"], r["te"], r["qe"], r["tb"], r["qb"]) == (255, 284, 254, 30, 0)
    r = oracle.align2(q[:250], t, mat, 6, 1, 6, 1, XB | XSTART)                     # 250 + shift 4 = 254 < 255: still exact
    assert (r["score"], r["qe"], r["qb"]) == (250, 249, 0)


def test_second_best_and_stop(oracle):
    rng = np.random.default_rng(5)
    mat = bwa_mat()
    q = rng.integers(0, 4, 80).astype(np.uint8)
    q_bad = q.copy()
    q_bad[[5, 40, 70]] = (q_bad[[5, 40, 70]] + 2) % 4                               # a second, worse copy of the read
    t = np.concatenate([rng.integers(0, 4, 50), q_bad, rng.integers(0, 4, 120), q, rng.integers(0, 4, 40)]).astype(np.uint8)
    r = oracle.align2(q, t, mat, 6, 1, 6, 1, XB | XSUBO | XSTART | 19)
    assert (r["score"], r["te"], r["qe"], r["tb"], r["qb"]) == (80, 50 + 80 + 120 + 79, 79, 250, 0)
    want2 = full_dp.local_dp(q, t[:140], mat, 6, 1, 6, 1)                           # the worse copy alone
    assert r["score2"] == want2["score"] and 50 <= r["te2"] < 140
    # without KSW_XSUBO there is no list: score2 stays -1
    assert oracle.align2(q, t, mat, 6, 1, 6, 1, XB | XSTART)["score2"] == -1
    # KSW_XSUBO with a threshold above the score: no start pass (tb, qb stay -1)
    r = oracle.align2(q, t, mat, 6, 1, 6, 1, XB | XSUBO | XSTART | 200)
    assert (r["score"], r["tb"], r["qb"]) == (80, -1, -1)
    # KSW_XSTOP ends the scan at the first row that reaches the threshold
    r = oracle.align2(q, t, mat, 6, 1, 6, 1, XB | XSTOP | 30)
    assert 30 <= r["score"] <= 31 and r["te"] < 140


@pytest.mark.parametrize("seed", range(6))
def test_random_pairs_against_the_numpy_local_dp(oracle, seed):
    rng = np.random.default_rng(100 + seed)
    a, b = [(1, 4), (1, 4), (2, 3), (1, 1), (3, 5), (1, 6)][seed]
    o, e = [(6, 1), (4, 2), (5, 1), (3, 1), (10, 2), (0, 1)][seed]
    mat = bwa_mat(a, b, -1)
    n_eq = n = 0
    for k in range(60):
        ql, tl = int(rng.integers(1, 200)), int(rng.integers(1, 400))
        t = rng.integers(0, 4, tl).astype(np.uint8)
        if k % 3:
            src = t[int(rng.integers(0, max(1, tl - ql))):][:ql]
            q = _gen.mutate(rng, src, len(src), 0.05, 0.03)
        else:
            q = rng.integers(0, 5, ql).astype(np.uint8)
        if len(q) == 0:
            continue
        want = full_dp.local_dp(q, t, mat, o, e, o, e)
        for xb in (XB, 0):
            if xb and want["score"] + b >= 255:
                continue
            r = oracle.align2(q, t, mat, o, e, o, e, xb | XSTART)
            n += 1
            assert r["score"] <= want["score"]                                      # never above the unrestricted optimum
            if r["score"] == want["score"]:
                n_eq += 1
                if r["score"] > 0:
                    assert (r["te"], r["qe"]) in want["ends"]
            if r["score"] > 0:
                # the reported box [qb, qe] x [tb, te] really holds an alignment of that score
                assert 0 <= r["tb"] <= r["te"] and 0 <= r["qb"] <= r["qe"]
                box = full_dp.local_dp(q[r["qb"]:r["qe"] + 1], t[r["tb"]:r["te"] + 1], mat, o, e, o, e)
                assert box["score"] >= r["score"]
    assert n_eq >= 0.97 * n                                                          # the I->D restriction almost never binds


def test_batch_equals_single_calls(oracle, host):
    rng = np.random.default_rng(9)
    mat = bwa_mat()
    seqs, at = [], np.zeros(40, dtype=host.ATASK)
    for i in range(40):
        t = rng.integers(0, 4, int(rng.integers(20, 300))).astype(np.uint8)
        q = _gen.mutate(rng, t[5:5 + int(rng.integers(10, 120))], 60, 0.04, 0.02)
        seqs.append((q, t))
        at[i]["query"], at[i]["target"], at[i]["qlen"], at[i]["tlen"] = q.ctypes.data, t.ctypes.data, len(q), len(t)
        at[i]["xtra"] = XB | XSUBO | XSTART | 19
    out, cells = oracle.align2_batch(mat, 6, 1, 6, 1, at, nthreads=3)
    assert cells > 0
    for i, (q, t) in enumerate(seqs):
        r = oracle.align2(q, t, mat, 6, 1, 6, 1, int(at[i]["xtra"]))
        assert [r[k] for k in oracle.ALIGN_FIELDS] == out[i].tolist()
