"""Oracle for SURVEY.md §8f F4 (bwa ksw_global2: banded global alignment + CIGAR): analytic known answers, an
independent full-matrix Gotoh DP, and CIGAR re-scoring.  The reference holds no vectors for this path either
(it is not in the RTL at all), so these tests are what pins oracle/ksw_global_ref.c."""
import os
import sys

import numpy as np
import pytest

import _gen

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "py"))
import full_dp  # noqa: E402


def rescore(cigar, q, t, mat, o_del, e_del, o_ins, e_ins):
    """Score of the path a CIGAR describes, and the query/target bases it consumes."""
    mat = np.asarray(mat).reshape(5, 5)
    i = j = sc = 0
    for op, ln in cigar:
        if op == 0:
            for _ in range(ln):
                sc += int(mat[t[i]][q[j]]); i += 1; j += 1
        elif op == 2:
            sc -= o_del + e_del * ln; i += ln
        else:
            sc -= o_ins + e_ins * ln; j += ln
    return sc, j, i


def test_known_answers(host, oracle):
    m = host.bwa_matrix()
    s = np.array([0, 1, 2, 3, 0, 1, 2, 3, 2, 2, 1, 0], np.uint8)
    r = oracle.global2(s, s, m, 6, 1, 6, 1, 100)
    assert r["score"] == len(s) and r["cigar"] == [(0, len(s))]
    # one base missing from the query: a 1-base deletion; ties go to the diagonal first (m >= e), so the gap sits as far
    # left as the backtrack reaches it — score is what matters here
    q = np.delete(s, 5)
    r = oracle.global2(q, s, m, 6, 1, 6, 1, 100)
    assert r["score"] == len(q) - 7 and sum(ln for op, ln in r["cigar"] if op == 2) == 1
    r = oracle.global2(s, q, m, 6, 1, 6, 1, 100)
    assert r["score"] == len(q) - 7 and sum(ln for op, ln in r["cigar"] if op == 1) == 1
    # all mismatches, gaps too expensive: L mismatches
    r = oracle.global2(np.zeros(8, np.uint8), np.ones(8, np.uint8), m, 20, 1, 20, 1, 100)
    assert r["score"] == -32 and r["cigar"] == [(0, 8)]
    # ... with cheap gaps "7I 1M 7D" beats 8 mismatches: -(6+7) - 4 - (6+7) (a gap opens from a match state only)
    r = oracle.global2(np.zeros(8, np.uint8), np.ones(8, np.uint8), m, 6, 1, 6, 1, 100)
    assert r["score"] == -30
    # empty target: the whole query is an insertion; empty query: a deletion
    r = oracle.global2(s, np.zeros(0, np.uint8), m, 6, 1, 6, 1, 100)
    assert r["cigar"] == [(1, len(s))]
    r = oracle.global2(np.zeros(0, np.uint8), s, m, 6, 1, 6, 1, 100)
    assert r["cigar"] == [(2, len(s))]
    # N scores -1 against anything
    r = oracle.global2(np.array([4, 0, 1], np.uint8), np.array([2, 0, 1], np.uint8), m, 6, 1, 6, 1, 10)
    assert r["score"] == 1


@pytest.mark.parametrize("pen", [(6, 1, 6, 1), (5, 2, 7, 1), (0, 1, 0, 1), (11, 3, 4, 2)])
def test_agrees_with_full_matrix_gotoh_when_the_band_cannot_bind(host, oracle, pen):
    rng = np.random.default_rng(sum(pen))
    m = host.bwa_matrix()
    for it in range(60):
        tl = int(rng.integers(0, 70))
        t = rng.integers(0, 4, tl).astype(np.uint8)
        q = _gen.mutate(rng, t, int(rng.integers(0, 70)), 0.08, 0.06) if rng.random() < 0.8 else rng.integers(0, 5, int(rng.integers(0, 70))).astype(np.uint8)
        r = oracle.global2(q, t, m, *pen, 200)
        assert r["score"] == full_dp.global_dp(q, t, m, *pen), (it, pen)
        sc, cq, ct = rescore(r["cigar"], q, t, m, *pen)
        assert (cq, ct) == (len(q), len(t))
        assert sc == r["score"], (it, pen, r["cigar"])


def test_banded_cigars_are_consistent(host, oracle):
    rng = np.random.default_rng(5)
    m = host.bwa_matrix()
    for it in range(200):
        tl = int(rng.integers(1, 200))
        t = rng.integers(0, 4, tl).astype(np.uint8)
        q = _gen.mutate(rng, t, max(1, tl + int(rng.integers(-6, 7))), 0.05, 0.03)
        w = int(rng.integers(abs(len(q) - tl) + 1, 40))
        r = oracle.global2(q, t, m, 6, 1, 6, 1, w)
        sc, cq, ct = rescore(r["cigar"], q, t, m, 6, 1, 6, 1)
        assert (cq, ct) == (len(q), tl) and sc == r["score"]
        wide = oracle.global2(q, t, m, 6, 1, 6, 1, 500)
        assert wide["score"] >= r["score"]
        assert oracle.global2(q, t, m, 6, 1, 6, 1, w, want_cigar=False)["score"] == r["score"]
        assert r["cells"] <= (2 * w + 1) * tl
