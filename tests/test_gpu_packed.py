"""bsw_submit_packed / bsw_upload_packed: sequences handed over 4-bit packed (16 bases per uint64, the device layout and
the encoding of the reference's link, sw_pe_array_proc_element.v:1638,1677-1683) must give the same result batch as the
byte-per-base path and as the oracle — through the direct DMA (registered arena) and through the gather path."""
import numpy as np
import pytest

import _gen
from test_gpu_parity import assert_same

pytestmark = pytest.mark.gpu
MIXED = dict(seed_len_min=19, seed_len_max=60, seed_at_start=0, junk_frac=0.1, n_rate=0.002, indel_rate=0.01)


@pytest.mark.parametrize("kernel", [0, 1, 2])
@pytest.mark.parametrize("registered", [False, True])
def test_packed_submit_equals_bytes_and_oracle(host, oracle, kernel, registered):
    tasks, arena = host.synth_tasks(30000, seed=61, **MIXED)
    p = host.default_params()
    want = oracle.pair_batch(p, tasks, nthreads=8)
    ha = None
    if registered:
        need = int(host.lib().bsw_pack_tasks_bound(tasks.ctypes.data, len(tasks)))
        ha = host.HostArena(need + 64)
        pt, words = host.pack_tasks(tasks, ha.view(np.uint64, need // 8 + 1))
    else:
        pt, words = host.pack_tasks(tasks)
    with host.BswContext(device=0, kernel=kernel, chunk_tasks=8192, streams=3) as c:
        got = c.extend_pairs_packed(p, pt)
        assert_same(got, want, tasks)
        assert got.tobytes() == c.extend_pairs(p, tasks).tobytes()
        b = c.upload_packed(p, pt)                      # resident batch from packed input
        c.run(b); c.sync()
        assert c.download(b).tobytes() == want.tobytes()
        b.free()
    if ha is not None:
        ha.free()


def test_packed_edge_shapes(host, oracle, ctx):
    """empty sides, empty targets, 1-base sequences, lengths around the 16-base word, Ns (codes 4..7)"""
    rng = np.random.default_rng(8)
    seeds = []
    for ql in (1, 2, 15, 16, 17, 31, 32, 33, 130, 231):
        for tl in (0, 1, 16, ql, 2 * ql + 3):
            q = rng.integers(0, 4, ql).astype(np.uint8)
            t = np.concatenate([q, rng.integers(0, 4, 300).astype(np.uint8)])[:tl]
            q[rng.random(ql) < 0.05] = 4
            seeds.append(dict(rq=q, rt=t, h0=int(rng.integers(1, 20))))
            seeds.append(dict(lq=q.copy(), lt=t.copy(), h0=int(rng.integers(1, 20))))
            seeds.append(dict(lq=q.copy(), lt=t.copy(), rq=q[::-1].copy(), rt=t[::-1].copy(), h0=3))
    tasks, arena = host.make_tasks(seeds)
    p = host.default_params()
    pt, words = host.pack_tasks(tasks)
    assert_same(ctx.extend_pairs_packed(p, pt), oracle.pair_batch(p, tasks), tasks)


def test_packed_rejects_misaligned_pointers(host, ctx):
    tasks, arena = host.synth_tasks(10, seed=2)
    pt, words = host.pack_tasks(tasks)
    pt["rquery"][3] += 4
    with pytest.raises(host.BswError):
        ctx.extend_pairs_packed(host.default_params(), pt)
    assert len(ctx.extend_pairs(host.default_params(), tasks)) == 10      # the context survives the refusal


def test_packed_over_several_devices_and_empty_batches(host, oracle):
    """chunk k -> devices[k mod n] with packed input (the same ordinal twice on this one-GPU box); n = 0; a ragged tail"""
    n = 40000
    tasks, arena = host.synth_tasks(n, seed=62, **MIXED)
    p = host.default_params(variant=1, o_ins=4, e_ins=2)
    want = oracle.pair_batch(p, tasks, nthreads=8)
    need = int(host.lib().bsw_pack_tasks_bound(tasks.ctypes.data, len(tasks)))
    ha = host.HostArena(need + 64)
    pt, words = host.pack_tasks(tasks, ha.view(np.uint64, need // 8 + 1))
    with host.BswContext(devices=[0, 0], kernel=host.KERNEL_LANE, streams=2, chunk_tasks=3000) as c:
        assert_same(c.extend_pairs_packed(p, pt), want, tasks)
        assert_same(c.extend_pairs_packed(p, pt[:2999]), want[:2999])
        assert len(c.extend_pairs_packed(p, pt[:0])) == 0
        b = c.upload_packed(p, pt[:0])
        c.run(b); c.sync()
        assert len(c.download(b)) == 0
        b.free()
    ha.free()


@pytest.mark.parametrize("kernel", [0, 1, 2])
def test_padding_nibbles_of_the_last_word_are_ignored(host, oracle, kernel):
    """A caller's packed buffer need not be zero behind a sequence's last base: whatever sits in the unused nibbles of its
    last word (the next read's bases in a tightly packed stream, say) changes nothing — not the DP, not the bins' N key."""
    tasks, arena = host.synth_tasks(40000, seed=77, **MIXED)
    p = host.default_params()
    want = oracle.pair_batch(p, tasks, nthreads=8)
    pt, words = host.pack_tasks(tasks)
    rng = np.random.default_rng(5)
    base = words.ctypes.data
    w = words.view(np.uint64)
    for qf, lf in (("lquery", "lqlen"), ("ltarget", "ltlen"), ("rquery", "rqlen"), ("rtarget", "rtlen")):
        ln = pt[lf].astype(np.int64)
        sel = np.nonzero((ln > 0) & (ln % 16 != 0))[0]
        idx = ((pt[qf][sel].astype(np.int64) - base) >> 3) + (ln[sel] - 1) // 16          # the sequence's last word
        keep = (np.uint64(1) << (np.uint64(4) * (ln[sel] % 16).astype(np.uint64))) - np.uint64(1)
        junk = rng.integers(0, 2 ** 63, len(sel), dtype=np.uint64) * np.uint64(2) + np.uint64(1)
        w[idx] = (w[idx] & keep) | (junk & ~keep)
    with host.BswContext(device=0, kernel=kernel, chunk_tasks=16384) as c:
        assert_same(c.extend_pairs_packed(p, pt), want, tasks)


PAIR_FIELDS = ("tag", "qb", "qe", "rb", "re", "score", "truesc", "w")


@pytest.mark.parametrize("kernel", [0, 1, 2])
def test_pair_record_format(host, oracle, kernel):
    """bsw_config.result_format = BSW_RESULT_PAIR: the submit calls hand back the RTL's 5-word record alone (32 of the 96
    bytes, sw_pe_array_proc_element.v:1662-1665) — the same eight fields, bit for bit, as the full-record run; bytes,
    packed and device-reference input, registered and pageable result arrays, lane bins with band retries (redo list)
    and general-kernel classes."""
    tasks, arena = host.synth_tasks(40000, seed=71, indel_rate=0.02, **{k: v for k, v in MIXED.items() if k != "indel_rate"})
    long_seeds = _gen.random_seeds(np.random.default_rng(3), 300, qmin=140, qmax=250, h0max=400)       # general-kernel classes
    lt, la = host.make_tasks(long_seeds)
    for p in (host.default_params(), host.default_params(w=8, zdrop=0)):                              # w = 8: many band retries
        want = oracle.pair_batch(p, tasks, nthreads=8)
        pt, words = host.pack_tasks(tasks)
        hout = host.HostArena(len(tasks) * host.PAIR.itemsize)
        with host.BswContext(device=0, kernel=kernel, chunk_tasks=8192, streams=3, result_format=host.RESULT_PAIR) as c:
            got = c.extend_pairs(p, tasks)
            assert got.dtype == host.PAIR
            for f in PAIR_FIELDS:
                assert (got[f] == want[f]).all(), f
            gp = c.extend_pairs_packed(p, pt, out=hout.view(host.PAIR, len(tasks)))                   # registered result array: direct DMA
            assert gp.tobytes() == got.tobytes()
            gl = c.extend_pairs(p, lt)
            wl = oracle.pair_batch(p, lt, nthreads=8)
            for f in PAIR_FIELDS:
                assert (gl[f] == wl[f]).all(), f
        hout.free()


@pytest.mark.parametrize("n", [200, 3000])
def test_extend_batch_on_a_pair_record_context(host, oracle, n):
    """bsw_extend_batch hands back bsw_ext records whatever the context's result_format is: BSW_RESULT_PAIR applies to the
    bsw_submit* calls only (round 4's advisor: on a PAIR context the synchronous chunk copied 32-byte pair records into the
    96-byte array bsw_extend_batch reads its sides from).  200 tasks: the single-DMA small-batch path; 3 000: a staged chunk."""
    rng = np.random.default_rng(11)
    et = np.zeros(n, dtype=host.EXT_TASK)
    keep = []
    for i in range(n):
        ql, tl = int(rng.integers(1, 180)), int(rng.integers(0, 260))
        t = rng.integers(0, 4, tl).astype(np.uint8)
        q = _gen.mutate(rng, t, ql, 0.04, 0.02)
        keep.append((q, t))
        et[i]["query"], et[i]["target"] = q.ctypes.data, t.ctypes.data if tl else 0
        et[i]["qlen"], et[i]["tlen"] = ql, tl
        et[i]["w"], et[i]["end_bonus"], et[i]["h0"] = int(rng.choice([3, 20, 100])), int(rng.choice([0, 5])), int(rng.integers(1, 70))
    p = host.default_params()
    want = oracle.ext_batch(p, et, nthreads=4)
    with host.BswContext(device=0, result_format=host.RESULT_PAIR) as c:
        got = c.extend_batch(p, et)
    for f in ("score", "qle", "tle", "gtle", "gscore", "max_off", "cells"):
        assert (got[f] == want[f]).all(), f
