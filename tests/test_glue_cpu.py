"""mem_chain2aln caller glue on the CPU: window, cal_max_gap, left-side reversal,
result -> alignment region — checked end to end on a synthetic genome with the oracle as the extender."""
import numpy as np
import pytest


def test_cal_max_gap(host):
    L = host.lib()
    for over in (dict(), dict(w=10), dict(o_del=0, e_del=2, o_ins=11, e_ins=3)):
        p = host.default_params(**over)
        a, od, ed, oi, ei, w = 1, int(p["o_del"][0]), int(p["e_del"][0]), int(p["o_ins"][0]), int(p["e_ins"][0]), int(p["w"][0])
        for q in (0, 1, 5, 6, 7, 19, 131, 400):
            l = max(int((q * a - od) / ed + 1.0), int((q * a - oi) / ei + 1.0), 1)
            assert L.bsw_cal_max_gap(p.ctypes.data, q) == min(l, 2 * w)


def test_pac_get_seq_both_strands(host):
    rng = np.random.default_rng(1)
    g = rng.integers(0, 4, 1001).astype(np.uint8)
    pac, lp = host.pack_pac(g), len(g)
    for beg, end in ((0, 50), (3, 7), (990, 1001), (17, 17)):
        assert (host.pac_get_seq(pac, lp, beg, end) == g[beg:end]).all()
    # reverse strand: position x >= l_pac is the complement of base 2*l_pac-1-x
    rc = (3 - g[::-1])
    for beg, end in ((lp, lp + 40), (lp + 100, lp + 333), (2 * lp - 5, 2 * lp)):
        assert (host.pac_get_seq(pac, lp, beg, end) == rc[beg - lp:end - lp]).all()
    assert len(host.pac_get_seq(pac, lp, lp - 10, lp + 10)) == 0        # bridging -> nothing
    assert (host.pac_get_seq(pac, lp, 50, 0) == g[0:50]).all()           # swapped bounds
    assert (host.pac_get_seq(pac, lp, -20, 10) == g[0:10]).all()         # clamped


@pytest.mark.parametrize("variant", [0, 1])
def test_chain2aln_glue_recovers_read_positions(host, oracle, variant):
    rng = np.random.default_rng(5 + variant)
    lp = 20000
    g = rng.integers(0, 4, lp).astype(np.uint8)
    pac = host.pack_pac(g)
    both = np.concatenate([g, 3 - g[::-1]])                               # bwa's 2*l_pac coordinate space
    p = host.default_params(variant=variant)
    reads, seeds, truth = [], np.zeros(300, dtype=host.SEED), []
    for i in range(300):
        rl = 150
        strand = i % 2
        pos = int(rng.integers(300, lp - 300)) + strand * lp
        read = both[pos:pos + rl].copy()
        sl = int(rng.integers(19, 40))
        qb = int(rng.integers(0, rl - sl + 1))
        if i % 3 == 0:                                                    # two substitutions outside the seed
            for x in rng.integers(0, rl, 2):
                if not (qb <= x < qb + sl):
                    read[x] = (read[x] + 1) % 4
        reads.append(read)
        seeds[i] = (pos + qb, qb, sl)
        truth.append(pos)
    tasks, keep = host.seeds_to_tasks(p, pac, lp, reads, seeds)
    # the left sequences are reversed, the right ones are views (F2 layout)
    for i in (0, 1, 2, 7):
        qb, sl = int(seeds[i]["qbeg"]), int(seeds[i]["len"])
        assert (host.task_seq(tasks, i, "lquery", "lqlen") == reads[i][:qb][::-1]).all()
        assert (host.task_seq(tasks, i, "rquery", "rqlen") == reads[i][qb + sl:]).all()
        assert int(tasks[i]["h0"]) == sl and int(tasks[i]["init_score"]) == -1 and int(tasks[i]["qbeg"]) == qb
        lt = host.task_seq(tasks, i, "ltarget", "ltlen")
        if len(lt):
            assert (lt == both[int(seeds[i]["rbeg"]) - len(lt):int(seeds[i]["rbeg"])][::-1]).all()
        rt = host.task_seq(tasks, i, "rtarget", "rtlen")
        assert (rt == both[int(seeds[i]["rbeg"]) + sl:int(seeds[i]["rbeg"]) + sl + len(rt)]).all()
    res = oracle.pair_batch(p, tasks)
    aln = host.results_to_alnregs(seeds, res)
    for i in range(300):
        if i % 3:                                                         # exact reads: end-to-end, full score
            assert (int(aln[i]["qb"]), int(aln[i]["qe"])) == (0, 150)
            assert (int(aln[i]["rb"]), int(aln[i]["re"])) == (truth[i], truth[i] + 150)
            assert int(aln[i]["score"]) == 150 and int(aln[i]["truesc"]) == 150
        else:                                                             # substitutions: still anchored at the truth
            assert int(aln[i]["rb"]) - int(aln[i]["qb"]) == truth[i]
            assert int(aln[i]["re"]) - int(aln[i]["qe"]) == truth[i]
            assert int(aln[i]["score"]) >= 150 - 10 - 10


def test_chain_window_never_bridges_strands(host):
    p = host.default_params()
    lp = 1000
    s = np.zeros(1, dtype=host.SEED)
    r = np.zeros(2, dtype=np.int64)
    s[0] = (lp - 30, 50, 25)                                              # forward seed near the boundary
    assert host.lib().bsw_chain_window(p.ctypes.data, s.ctypes.data, 1, 150, lp, r.ctypes.data) == 0
    assert r[1] == lp and 0 <= r[0] < lp
    s[0] = (lp + 10, 50, 25)                                              # reverse-strand seed
    assert host.lib().bsw_chain_window(p.ctypes.data, s.ctypes.data, 1, 150, lp, r.ctypes.data) == 0
    assert r[0] == lp and r[1] <= 2 * lp
