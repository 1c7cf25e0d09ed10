"""ABI 6: several submits in flight per context (the reference's manager keeps four task batches going:
batch_manager.v:343-348 request bits, :434-435 busy bitmap), a ticket per submit, the non-blocking status poll
(batch_manager.v:844-854), per-ticket failures, and bsw_host_stats.  Parity is against the CPU oracle."""
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

MIXED = dict(seed_len_min=19, seed_len_max=60, seed_at_start=0, sub_rate=0.02, indel_rate=0.01, junk_frac=0.1, n_rate=0.001)


def test_four_overlapping_submits_bit_exact_in_task_order(host, oracle):
    """MAX_INFLIGHT submits of different sizes and formats through ONE context; each result batch must equal the oracle's
    in task order, whichever order they are collected in."""
    p = host.default_params()
    sizes = [150_000, 40_000, 90_000, 3_000]
    sets = [host.synth_tasks(n, seed=300 + k, **MIXED) for k, n in enumerate(sizes)]
    want = [oracle.pair_batch_avx2(p, t, nthreads=8) for t, _ in sets]
    ptasks, parena = host.pack_tasks(sets[1][0])                       # the second one goes in 4-bit packed
    with host.BswContext(device=0, chunk_tasks=16384) as c:
        outs, tickets = [], []
        for k, (t, _) in enumerate(sets):
            outs.append(c.submit_packed(p, ptasks) if k == 1 else c.submit(p, t))
            tickets.append(c.last_ticket)
        assert len(set(tickets)) == 4 and all(tickets) and c.inflight() == 4
        with pytest.raises(host.BswError) as ei:                       # a fifth is refused, nothing else changes
            c.submit(p, sets[3][0])
        assert ei.value.code == -6
        for k in (2, 0):                                               # collected out of order
            c.wait_ticket(tickets[k])
            assert outs[k].tobytes() == want[k].tobytes(), k
        assert c.inflight() == 2
        with pytest.raises(host.BswError):                             # a collected ticket is gone
            c.wait_ticket(tickets[0])
        c.wait()                                                       # the rest
        for k in (1, 3):
            assert outs[k].tobytes() == want[k].tobytes(), k
        st = c.host_stats()
        assert st["seeds"] == sum(sizes) and st["submits"] == 4 and st["slot_threads"] == 4
        assert st["slot_cpu_ns"] > 0 and st["d2h_bytes"] == sum(sizes) * host.RESULT.itemsize


def test_status_poll_never_blocks(host, oracle):
    p = host.default_params()
    tasks, arena = host.synth_tasks(400_000, seed=77, **MIXED)
    want = oracle.pair_batch_avx2(p, tasks, nthreads=8)
    with host.BswContext(device=0) as c:
        out = c.submit(p, tasks)
        t = c.last_ticket
        polls, t0 = 0, time.perf_counter()
        first = c.test(t)
        dt_first = time.perf_counter() - t0
        for _ in range(2):                                              # (the quickest of three: a loaded box may deschedule one call)
            t1 = time.perf_counter()
            c.test(t)
            dt_first = min(dt_first, time.perf_counter() - t1)
        while not c.test(t):
            polls += 1
            assert time.perf_counter() - t0 < 60
            time.sleep(0.0005)
        assert dt_first < 0.005                                         # the poll itself does not wait for the GPU
        assert not first and polls > 0                                  # 400 k seeds are not done in the time of one call
        assert c.test(t) and c.inflight() == 1                          # polling does not collect
        c.wait_ticket(t)
        assert out.tobytes() == want.tobytes()
        with pytest.raises(host.BswError):
            c.test(t)


def test_a_failing_submit_leaves_the_others_alone(host, oracle):
    """One submit whose middle chunk holds an invalid seed (h0 = 0) between two good ones: the bad ticket reports
    BSW_E_INVAL naming the task, the others are bit-exact, the context keeps working."""
    p = host.default_params()
    good_a, _a = host.synth_tasks(60_000, seed=11, **MIXED)
    bad, _b = host.synth_tasks(50_000, seed=12, **MIXED)
    good_c, _c = host.synth_tasks(70_000, seed=13, **MIXED)
    bad["h0"][25_017] = 0
    wa, wc = oracle.pair_batch_avx2(p, good_a, nthreads=8), oracle.pair_batch_avx2(p, good_c, nthreads=8)
    with host.BswContext(device=0, chunk_tasks=8192) as c:
        oa = c.submit(p, good_a); ta = c.last_ticket
        c.submit(p, bad); tb = c.last_ticket
        oc = c.submit(p, good_c); tc = c.last_ticket
        with pytest.raises(host.BswError) as ei:
            c.wait_ticket(tb)
        assert ei.value.code == -2 and "25017" in str(ei.value)
        c.wait_ticket(tc)
        c.wait_ticket(ta)
        assert oa.tobytes() == wa.tobytes() and oc.tobytes() == wc.tobytes()
        again = c.extend_pairs(p, good_a)                               # the slots are in order after the failure
        assert again.tobytes() == wa.tobytes()
        # bsw_wait reports the first failure in submit order and still waits for everything
        o1 = c.submit(p, good_c)
        c.submit(p, bad)
        with pytest.raises(host.BswError) as ei:
            c.wait()
        assert ei.value.code == -2 and c.inflight() == 0
        assert o1.tobytes() == wc.tobytes()


def test_stream_of_submits_through_one_context(host, oracle):
    """The usage the queue is for: an aligner that keeps producing seed batches keeps two submits in flight in ONE
    context (bench.py's stream legs): 12 batches, alternating buffers, every result batch checked."""
    p = host.default_params()
    batches = [host.synth_tasks(30_000 + 1000 * k, seed=500 + k, **MIXED) for k in range(4)]
    want = [oracle.pair_batch_avx2(p, t, nthreads=8) for t, _ in batches]
    with host.BswContext(device=0, streams=4, chunk_tasks=8192) as c:
        pending = []
        for k in range(12):
            if len(pending) == 2:
                j, out, tk = pending.pop(0)
                c.wait_ticket(tk)
                assert out.tobytes() == want[j % 4].tobytes(), j
            out = c.submit(p, batches[k % 4][0])
            pending.append((k, out, c.last_ticket))
        for j, out, tk in pending:
            c.wait_ticket(tk)
            assert out.tobytes() == want[j % 4].tobytes(), j
        assert c.host_stats()["submits"] == 12


def test_effective_timeout_from_the_environment(host, monkeypatch):
    """bsw_default_config leaves timeout_ms = 0 = `the library default`: BSW_TIMEOUT_MS when set, else 120 s; an explicit
    value wins (the default used to write 120000, which made the environment variable dead)."""
    cfg = np.zeros(1, dtype=host.CONFIG)
    host.lib().bsw_default_config(cfg.ctypes.data)
    assert int(cfg["timeout_ms"][0]) == 0
    monkeypatch.delenv("BSW_TIMEOUT_MS", raising=False)
    assert host.lib().bsw_effective_timeout_ms(cfg.ctypes.data) == 120000
    monkeypatch.setenv("BSW_TIMEOUT_MS", "5000")
    assert host.lib().bsw_effective_timeout_ms(cfg.ctypes.data) == 5000
    cfg["timeout_ms"] = 250
    assert host.lib().bsw_effective_timeout_ms(cfg.ctypes.data) == 250
    # ... and a context created from the defaults runs with it: a 1 ms watchdog from the environment trips on a large batch
    monkeypatch.setenv("BSW_TIMEOUT_MS", "1")
    n = 600000
    ha = host.HostArena(host.synth_arena_bound(n) + 4096)
    ho = host.HostArena(n * host.RESULT.itemsize)
    tasks, _ = host.synth_tasks(n, arena=ha.u8, seed=55)
    c = host.BswContext(device=0, streams=1, chunk_tasks=n)
    with pytest.raises(host.BswError) as ei:
        c.extend_pairs(host.default_params(), tasks, out=ho.view(host.RESULT, n))
    assert ei.value.code == -4 and "timeout" in str(ei.value)
    c.close()
    time.sleep(0.05)
    ha.free()
    ho.free()
