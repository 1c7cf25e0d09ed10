"""bsw_synth_ref_generate (host C, no GPU): the synthetic genome + reads the bench's device-resident-reference leg runs on."""
import numpy as np


def test_reads_come_from_the_genome_and_windows_are_bwas(host, oracle):
    p = host.default_params()
    lp, n, L = 100_000, 3000, 150
    pac, rt, arena = host.synth_ref_tasks(n, lp, p, seed=5, read_len=L, seed_len_min=19, seed_len_max=50, seed_at_start=0,
                                          sub_rate=0.01, indel_rate=0.002, junk_frac=0.0)
    pos = np.arange(lp)
    genome = ((pac[pos >> 2] >> ((~pos & 3) << 1)) & 3).astype(np.uint8)             # .pac layout: first base in the top two bits
    rmax = np.zeros(2, dtype=np.int64)
    assert (rt["l_query"] == L).all() and (rt["init_score"] == -1).all() and (rt["tag"] == np.arange(n)).all()
    for i in range(0, n, 7):
        sd = rt[i]["seed"]
        q = arena[i * L:(i + 1) * L]
        assert rt[i]["query"] == arena.ctypes.data + i * L
        assert 0 <= sd["rbeg"] and sd["rbeg"] + sd["len"] <= lp and sd["qbeg"] + sd["len"] <= L
        assert (q[sd["qbeg"]:sd["qbeg"] + sd["len"]] == genome[sd["rbeg"]:sd["rbeg"] + sd["len"]]).all()      # the seed is exact
        host.lib().bsw_chain_window(p.ctypes.data, rt[i:i + 1]["seed"].copy().ctypes.data, 1, L, lp, rmax.ctypes.data)
        assert (rt[i]["rmax0"], rt[i]["rmax1"]) == (rmax[0], rmax[1])
    reads = [arena[i * L:(i + 1) * L] for i in range(n)]
    tasks, keep = host.seeds_to_tasks(p, pac, lp, reads, rt["seed"].copy())
    res = oracle.pair_batch(p, tasks, nthreads=4)
    assert np.median(res["score"]) >= 120          # the flanks really derive from the genome (1% subs, 0.2% indels)
    # determinism, and a loud error for a short arena
    pac2, rt2, arena2 = host.synth_ref_tasks(n, lp, p, seed=5, read_len=L, seed_len_min=19, seed_len_max=50, seed_at_start=0,
                                             sub_rate=0.01, indel_rate=0.002, junk_frac=0.0)
    assert (pac2 == pac).all() and (arena2[:n * L] == arena[:n * L]).all() and (rt2["seed"] == rt["seed"]).all()
    try:
        host.synth_ref_tasks(n, lp, p, arena=np.zeros(100, np.uint8))
        assert False
    except ValueError:
        pass
