"""Differential + property tests of the CPU oracle: scalar restatement vs the row-synchronous
model (the algebra the wave kernel uses) vs an independent full-matrix numpy DP."""
import os
import sys

import numpy as np
import pytest

from test_oracle_kat import mat

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "py"))
from full_dp import full_dp  # noqa: E402


def _case(rng, it):
    ql = int(rng.integers(1, 70))
    tl = int(rng.integers(0, 120))
    q = rng.integers(0, 5 if it % 7 == 0 else 4, ql)
    if it % 3 == 0:
        t = rng.integers(0, 4, tl)
    else:
        t = []
        for b in q:
            x = rng.random()
            if x < 0.04:
                continue
            if x < 0.08:
                t.append(int(rng.integers(0, 4)))
            t.append(int(b) if rng.random() > 0.08 else int(rng.integers(0, 4)))
        t = np.array((t + list(rng.integers(0, 4, tl)))[:tl], dtype=np.uint8)
    pen = [int(rng.integers(0, 8)), int(rng.integers(1, 4)), int(rng.integers(0, 8)), int(rng.integers(1, 4))]
    w = int(rng.choice([1, 2, 5, 10, 30, 100]))
    zd = int(rng.choice([0, 5, 20, 100]))
    return q, t, pen, w, zd, int(rng.integers(1, 60)), int(rng.integers(0, 10))


def test_model_equals_oracle_random(oracle):
    rng = np.random.default_rng(2024)
    m = mat()
    for it in range(6000):
        q, t, pen, w, zd, h0, eb = _case(rng, it)
        for v in (0, 1):
            a = oracle.extend2(q, t, m, *pen, w, eb, zd, h0, variant=v)
            b = oracle.extend2(q, t, m, *pen, w, eb, zd, h0, variant=v, model=True)
            assert a == b, (it, v, a, b, list(q), list(t), pen, w, zd, h0, eb)


def test_model_equals_oracle_random_matrix(oracle):
    rng = np.random.default_rng(77)
    for it in range(1500):
        q, t, pen, w, zd, h0, eb = _case(rng, it)
        m = rng.integers(-6, 4, 25).astype(np.int8)
        for v in (0, 1):
            a = oracle.extend2(q, t, m, *pen, w, eb, zd, h0, variant=v)
            b = oracle.extend2(q, t, m, *pen, w, eb, zd, h0, variant=v, model=True)
            assert a == b, (it, v, a, b)


@pytest.mark.parametrize("variant", [0, 1])
def test_full_matrix_dp_agrees_when_nothing_binds(oracle, variant):
    # h0 large -> every H > 0 -> no trimming, no m==0 break; w and end_bonus large -> no band; zdrop 0
    rng = np.random.default_rng(31 + variant)
    m = mat()
    for it in range(150):
        ql, tl = int(rng.integers(1, 26)), int(rng.integers(1, 26))
        q = rng.integers(0, 5 if it % 5 == 0 else 4, ql)
        t = np.resize(q, tl) if it % 2 == 0 else rng.integers(0, 4, tl)
        h0 = int(rng.integers(150, 250))
        a = oracle.extend2(q, t, m, 6, 1, 5, 2, 1000, 100000, 0, h0, variant=variant)
        b = full_dp(q, t, m, 6, 1, 5, 2, h0, variant=variant)
        assert b["minH"] > 0
        for k in ("score", "qle", "tle", "gtle", "gscore", "max_off"):
            assert a[k] == b[k], (it, k, a, b)
        assert a["cells"] == ql * tl


def test_properties(oracle):
    rng = np.random.default_rng(5)
    m = mat()
    for it in range(1500):
        q, t, pen, w, zd, h0, eb = _case(rng, it)
        pen = [6, 1, 6, 1] if it % 2 else pen
        for v in (0, 1):
            r = oracle.extend2(q, t, m, *pen, w, eb, zd, h0, variant=v)
            assert r["score"] >= h0
            assert r["score"] <= h0 + len(q) * 1
            assert 0 <= r["qle"] <= len(q) and 0 <= r["tle"] <= len(t) and 0 <= r["gtle"] <= len(t)
            assert r["gscore"] <= r["score"] or r["gscore"] <= h0 + len(q)
            assert r["cells"] <= max(len(q), 1) * len(t)
            if r["qle"] == 0:
                assert r["score"] == h0 and r["tle"] == 0
        # score is monotone non-decreasing in w when zdrop = 0 (wider band only adds paths)
        prev = None
        for ww in (1, 3, 10, 40, 200):
            r = oracle.extend2(q, t, m, 6, 1, 6, 1, ww, 100000, 0, h0, variant=1)
            if prev is not None:
                assert r["score"] >= prev
            prev = r["score"]


def test_pair_driver_matches_manual_composition(oracle, host):
    """bsw_pair_ref == mem_chain2aln logic composed by hand from ksw_extend2_ref calls (P1-P3)."""
    import _gen
    rng = np.random.default_rng(99)
    seeds = _gen.random_seeds(rng, 300, qmax=90, indel=0.03)
    tasks, arena = host.make_tasks(seeds)
    for variant in (0, 1):
        p = host.default_params(variant=variant)
        res = oracle.pair_batch(p, tasks)
        m, w = p["mat"][0], int(p["w"][0])
        for k, s in enumerate(seeds):
            score, aw = int(tasks[k]["init_score"]), [w, w]
            out = {}
            if "lq" in s:
                for tr in range(2):
                    prev, aw[0] = score, w << tr
                    r = oracle.extend2(s["lq"], s["lt"], m, 6, 1, 6, 1, aw[0], 5, 100, s["h0"], variant=variant)
                    score = r["score"]
                    if score == prev or r["max_off"] < (aw[0] >> 1) + (aw[0] >> 2):
                        break
                if r["gscore"] <= 0 or r["gscore"] <= score - 5:
                    out.update(qb=len(s["lq"]) - r["qle"], rb=-r["tle"], truesc=score)
                else:
                    out.update(qb=0, rb=-r["gtle"], truesc=r["gscore"])
            else:
                score = s["h0"]
                out.update(qb=0, rb=0, truesc=score)
            sc0 = score
            if "rq" in s:
                for tr in range(2):
                    prev, aw[1] = score, w << tr
                    r = oracle.extend2(s["rq"], s["rt"], m, 6, 1, 6, 1, aw[1], 5, 100, sc0, variant=variant)
                    score = r["score"]
                    if score == prev or r["max_off"] < (aw[1] >> 1) + (aw[1] >> 2):
                        break
                if r["gscore"] <= 0 or r["gscore"] <= score - 5:
                    out.update(qe=r["qle"], re=r["tle"], truesc=out["truesc"] + score - sc0)
                else:
                    out.update(qe=len(s["rq"]), re=r["gtle"], truesc=out["truesc"] + r["gscore"] - sc0)
            else:
                out.update(qe=0, re=0)
            got = res[k]
            assert (got["qb"], got["qe"], got["rb"], got["re"], got["score"], got["truesc"], got["w"]) == \
                (out["qb"], out["qe"], out["rb"], out["re"], score, out["truesc"], max(aw)), (k, variant)
