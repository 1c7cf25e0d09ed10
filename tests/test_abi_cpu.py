"""C-ABI library on a CPU-only box: it loads, exports every symbol of include/bwa_sw_mi355.h,
struct layouts match, host-only entry points work, GPU entry points fail loudly (no fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    txt = open(os.path.join(ROOT, "include", "bwa_sw_mi355.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b((?:bsw|ksw)_[a-z0-9_]+)\s*\(", txt)))


def test_every_declared_symbol_is_exported(host):
    L = C.CDLL(host.lib_path())
    names = header_functions()
    assert len(names) >= 25
    for n in names:
        assert hasattr(L, n), "libbwasw_mi355.so does not export %s" % n
    assert set(names) == set(host.EXPORTS)


def test_struct_sizes_match_header(host, tmp_path):
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include "bwa_sw_mi355.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu %zu\\n",'
                   'sizeof(bsw_params),sizeof(bsw_task),sizeof(bsw_ext),sizeof(bsw_result),sizeof(bsw_ext_task),'
                   'sizeof(bsw_synth_spec),sizeof(bsw_config));return 0;}\n')
    exe = tmp_path / "sz"
    import subprocess
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    sizes = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    assert sizes == [host.PARAMS.itemsize, host.TASK.itemsize, host.EXT.itemsize, host.RESULT.itemsize,
                     host.EXT_TASK.itemsize, host.SYNTH.itemsize, host.CONFIG.itemsize]


def test_default_params(host):
    p = host.default_params()
    assert (p["mat"][0] == host.bwa_matrix()).all()
    assert (int(p["o_del"][0]), int(p["e_del"][0]), int(p["o_ins"][0]), int(p["e_ins"][0]), int(p["w"][0])) == (6, 1, 6, 1, 100)
    assert (int(p["pen_clip5"][0]), int(p["pen_clip3"][0]), int(p["zdrop"][0]), int(p["max_band_try"][0]), int(p["variant"][0])) == (5, 5, 100, 2, 0)


def test_synth_is_deterministic_and_shaped(host):
    a, aa = host.synth_tasks(500, seed=3)
    b, ab = host.synth_tasks(500, seed=3)
    c, ac = host.synth_tasks(500, seed=4)
    assert (a["rqlen"] == 131).all() and (a["rtlen"] == 257).all() and (a["lqlen"] == 0).all() and (a["h0"] == 19).all()
    sa = [host.task_seq(a, i, "rquery", "rqlen").tobytes() + host.task_seq(a, i, "rtarget", "rtlen").tobytes() for i in range(500)]
    sb = [host.task_seq(b, i, "rquery", "rqlen").tobytes() + host.task_seq(b, i, "rtarget", "rtlen").tobytes() for i in range(500)]
    sc = [host.task_seq(c, i, "rquery", "rqlen").tobytes() + host.task_seq(c, i, "rtarget", "rtlen").tobytes() for i in range(500)]
    assert sa == sb and sa != sc
    m, am = host.synth_tasks(300, seed=9, seed_at_start=0, seed_len_min=19, seed_len_max=60, read_len=250, w=500)
    assert ((m["lqlen"] + m["rqlen"] + m["h0"]) == 250).all()
    assert (m["ltlen"] >= m["lqlen"]).all() and (m["rtlen"] >= m["rqlen"]).all()
    assert (m["qbeg"] == m["lqlen"]).all()


def test_no_gpu_means_loud_failure_not_fallback(host):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    assert host.lib().bsw_device_count() == 0
    with pytest.raises(host.BswError) as ei:
        host.BswContext(device=0)
    assert ei.value.code == -1          # BSW_E_NODEVICE
    # the drop-in scalar entry point must not silently compute on the CPU either
    q = np.zeros(4, np.uint8)
    m = host.bwa_matrix()
    r = host.lib().ksw_extend2(4, q.ctypes.data, 4, q.ctypes.data, 5, m.ctypes.data, 6, 1, 6, 1, 100, 5, 100, 10,
                               None, None, None, None, None)
    assert r == -1


def test_refbatch_roundtrip(host, oracle):
    tasks, arena = host.synth_tasks(900, seed=21, seed_at_start=0, seed_len_min=19, seed_len_max=60, indel_rate=0.01, n_rate=0.01)
    p = host.default_params()
    words, n = host.refbatch_encode(p, tasks)
    assert 0 < n <= host.REFBATCH_MAX_TASKS and words[2] == n
    p2, t2, seqbuf = host.refbatch_decode(words)
    for f in ("o_del", "e_del", "o_ins", "e_ins", "w", "pen_clip5", "pen_clip3"):
        assert int(p2[f][0]) == int(p[f][0])
    assert int(p2["zdrop"][0]) == 0 and int(p2["max_band_try"][0]) == 2
    for f in ("lqlen", "ltlen", "rqlen", "rtlen", "h0", "init_score", "qbeg", "tag"):
        assert (t2[f] == tasks[f][:n]).all(), f
    for i in range(0, n, 37):
        for pf, lf in (("lquery", "lqlen"), ("ltarget", "ltlen"), ("rquery", "rqlen"), ("rtarget", "rtlen")):
            assert (host.task_seq(t2, i, pf, lf) == host.task_seq(tasks, i, pf, lf)).all()
    # MSB-first nibble order of the first data word (proc_element.v:1638,1677)
    pos = int(words[8 + 2])
    first = np.concatenate([host.task_seq(tasks, 0, pf, lf) for pf, lf in (("lquery", "lqlen"), ("rquery", "rqlen"), ("ltarget", "ltlen"), ("rtarget", "rtlen"))])[:8]
    exp = 0
    for k, b in enumerate(first):
        exp |= int(b) << (28 - 4 * k)
    assert int(words[pos]) == exp
    # results
    p2["zdrop"] = 0
    res = oracle.pair_batch(p2, t2)
    rw = host.refbatch_encode_results(res)
    back = host.refbatch_decode_results(rw, n)
    for f in ("tag", "qb", "qe", "rb", "re", "score", "truesc", "w"):
        assert (back[f] == res[f]).all(), f


def test_refbatch_limits(host):
    p = host.default_params()
    seeds = [dict(rq=np.zeros(300, np.uint8), rt=np.zeros(300, np.uint8), h0=10)]      # qlen > 255: does not fit (Q1)
    t, a = host.make_tasks(seeds)
    words, n = host.refbatch_encode(p, t)
    assert n == 0
    seeds = [dict(rq=np.zeros(30, np.uint8), rt=np.zeros(30, np.uint8), h0=200)]       # h0 > 127: int8 datapath
    t, a = host.make_tasks(seeds)
    assert host.refbatch_encode(p, t)[1] == 0
    big = host.default_params(w=300)
    with pytest.raises(host.BswError):
        host.refbatch_encode(big, t)
    many, am = host.synth_tasks(2000, seed=1)
    words, n = host.refbatch_encode(p, many)
    assert n == host.REFBATCH_MAX_TASKS                                                 # 819 five-word records (rbb.v)


def test_pack_bases_matches_reference_packer(host):
    """16 bases per uint64, base k in bits [4k,4k+3], codes > 4 stored as N (4) — SWAR fast path and tails."""
    rng = np.random.default_rng(12)
    for trial in range(300):
        n = int(rng.integers(0, 200))
        hi = [4, 5, 8, 256][trial % 4]
        b = rng.integers(0, hi, n).astype(np.uint8)
        if trial % 7 == 0 and n:
            b[rng.integers(0, n)] = 255
        words, has_n = host.pack_bases(b)
        ref = np.zeros((n + 15) // 16, dtype=np.uint64)
        for k, v in enumerate(b):
            ref[k >> 4] |= np.uint64(min(int(v), 4)) << np.uint64(4 * (k & 15))
        assert (words == ref).all(), (trial, b)
        assert has_n == bool((b >= 4).any())


def test_batch_plan_bins(host):
    """Host batch manager (no GPU): every seed lands in exactly one kernel bin; lane bins hold each side once,
    sorted by (query holds an N, query length descending) inside a class; long or wide-score seeds go to the wave classes."""
    import _gen
    rng = np.random.default_rng(3)
    seeds = _gen.random_seeds(rng, 8000, qmax=300, nrate=0.002, h0max=80)
    seeds += [dict(rq=np.zeros(100, np.uint8), rt=np.zeros(120, np.uint8), h0=70000)]       # beyond 16-bit bins
    tasks, arena = host.make_tasks(seeds)
    n = len(tasks)
    p = host.default_params()
    lane_cols = [(8, 72), (8, 136), (8, 232), (16, 136)]      # narrowest first: a side takes the first class of its width that holds it
    for kernel in (host.KERNEL_AUTO, host.KERNEL_LANE, host.KERNEL_WAVE):
        order, seg, words = host.plan_batch(p, tasks, kernel=kernel, pack_threads=3)
        exp_words = int((((tasks["lqlen"] + 15) // 16 + (tasks["ltlen"] + 15) // 16) * (tasks["lqlen"] > 0)
                         + ((tasks["rqlen"] + 15) // 16 + (tasks["rtlen"] + 15) // 16) * (tasks["rqlen"] > 0)).sum())
        assert words == exp_words
        wave = order[seg[0]:seg[8]]
        lane = order[seg[8]:seg[9]]
        assert len(wave) + len(lane) == n and len(np.unique(np.concatenate([wave, lane]))) == n
        if kernel == host.KERNEL_WAVE or kernel == host.KERNEL_AUTO:     # AUTO: too few eligible seeds for a lane launch
            assert len(lane) == 0
        else:
            assert len(lane) > 2048
        # wave classes by columns per lane
        cols = [64, 128, 192, 256, 512, 1024]
        for c in range(6):
            t = tasks[order[seg[c]:seg[c + 1]]]
            qm = np.maximum(t["lqlen"], t["rqlen"])
            assert (qm + 1 <= cols[c]).all() and (c == 0 or (qm + 1 > cols[c - 1]).all())
        top = tasks["h0"].astype(np.int64) + tasks["lqlen"] + tasks["rqlen"]
        qmax = np.maximum(tasks["lqlen"], tasks["rqlen"])
        assert (top[lane] < 65000).all() and (qmax[lane] + 1 <= 232).all()
        for side, base, qf in ((0, 9, "lqlen"), (1, 17, "rqlen")):
            allside = order[seg[base]:seg[base + 8]]
            assert sorted(allside) == sorted(lane[tasks[qf][lane] > 0])
            for c, (bits, ncol) in enumerate(lane_cols):
                t = tasks[order[seg[base + c]:seg[base + c + 1]]]
                if len(t) == 0:
                    continue
                q = t[qf]
                hn = _gen.query_has_n(tasks, arena, side)[order[seg[base + c]:seg[base + c + 1]]]
                key = (~hn).astype(np.int64) * 1000 - q.astype(np.int64)
                assert (np.diff(key) >= 0).all() and (q + 1 <= ncol).all()        # queries with an N first, each part longest first
                if side == 0:                                                     # ... and inside a length the left sides by h0 bucket
                    lane_h0 = tasks["h0"][order[seg[9]:seg[13]]].astype(np.int64)
                    lo, hi = int(lane_h0.min()), int(lane_h0.max())
                    mul = (8 << 16) // (hi - lo + 1) if hi > lo else 0
                    hb = np.minimum(((t["h0"].astype(np.int64) - lo) * mul) >> 16, 7)
                    assert (np.diff(key * 8 + hb) >= 0).all()
                folded = c == 1 and seg[base + 1] == seg[base]                    # the 72-column class folded into the 136-column one
                if c > 0 and lane_cols[c - 1][0] == bits and not folded:
                    assert (q + 1 > lane_cols[c - 1][1]).all()                    # ... and not in a narrower class of the same width
                tt = t["h0"].astype(np.int64) + t["lqlen"] + t["rqlen"]
                # 8-bit classes need h0 + qlen*a + b <= 255 (b = 4: the packed kernel forms H + a + b in 8 bits)
                assert ((tt + 4 <= 255) if bits == 8 else (tt + 4 > 255) | (np.maximum(t["lqlen"], t["rqlen"]) + 1 > 232)).all()
        assert seg[26] - seg[25] == len(lane)                                 # redo list space
    # AUTO: small batches stay on the wave kernel, big eligible ones go to the lane bins
    order, seg, words = host.plan_batch(p, tasks[:500], kernel=host.KERNEL_AUTO)
    assert seg[9] - seg[8] == 0
    big, abig = host.synth_tasks(host.LANE_AUTO_MIN + 5000, seed=2)
    order, seg, words = host.plan_batch(p, big, kernel=host.KERNEL_AUTO, pack_threads=4)
    assert seg[9] - seg[8] == len(big) and seg[8] == 0


def test_bsw_bench_cli_without_a_gpu(host, tmp_path):
    """tools/bsw-bench (the reference host's CLI for this path, reference README.md:29-36): built by the Makefile,
    refuses --target=cpu, fails loudly without a GPU, and its task-batch dump round-trips through --load."""
    import subprocess
    exe = os.path.join(ROOT, "tools", "bsw-bench")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "bwa-mem-sw_amd", "csrc"), "bsw-bench"])
    r = subprocess.run([exe, "--target=cpu"], capture_output=True, text=True)
    assert r.returncode == 2 and "no CPU path" in r.stderr
    r = subprocess.run([exe, "--frobnicate"], capture_output=True, text=True)
    assert r.returncode == 2
    f1, f2 = str(tmp_path / "a.swb"), str(tmp_path / "b.swb")
    r = subprocess.run([exe, "-n", "700", "-l", "150", "-w", "50", "--pageable", "--dump", f1], capture_output=True, text=True)
    assert os.path.getsize(f1) > 700 * 40
    if host.lib().bsw_device_count() == 0:
        assert r.returncode == 1 and "no CPU path" in r.stderr          # no silent fallback
    r = subprocess.run([exe, "--load", f1, "--pageable", "--dump", f2], capture_output=True, text=True)
    assert open(f1, "rb").read() == open(f2, "rb").read()
    open(f2, "wb").write(b"garbage")
    r = subprocess.run([exe, "--load", f2, "--pageable"], capture_output=True, text=True)
    assert r.returncode == 1 and "cannot load" in r.stderr


def test_narrow_lane_class_is_a_per_chunk_decision(host):
    """The 72-column class (three waves per SIMD) is used when the short sides are most of a chunk's lane work and folded
    into the 136-column class otherwise (one launch per side packs better than two): bsw_plan_batch shows the decision."""
    import _gen
    rng = np.random.default_rng(8)
    p = host.default_params()

    def plan(qmax_list):
        seeds = []
        for qm in qmax_list:
            seeds += _gen.random_seeds(rng, 1500, qmax=qm, h0max=40)
        tasks, arena = host.make_tasks(seeds)
        order, seg, words = host.plan_batch(p, tasks, kernel=host.KERNEL_LANE)
        return tasks, order, seg

    def eight_bit(tasks):                                                 # (the 16-bit class takes the wide scores)
        return tasks["h0"].astype(np.int64) + tasks["lqlen"] + tasks["rqlen"] + 4 <= 255

    tasks, order, seg = plan([60])                                        # every side fits 72 columns
    for base, qf in ((9, "lqlen"), (17, "rqlen")):
        assert seg[base + 1] - seg[base] == int(((tasks[qf] > 0) & eight_bit(tasks)).sum()) and seg[base + 2] == seg[base + 1]
    tasks, order, seg = plan([60, 130, 130])                              # mostly long sides: folded, the narrow lists are empty
    for base, qf in ((9, "lqlen"), (17, "rqlen")):
        assert seg[base + 1] == seg[base]
        t = tasks[order[seg[base + 1]:seg[base + 2]]]
        assert len(t) == int(((tasks[qf] > 0) & eight_bit(tasks)).sum()) and (t[qf] < 72).any() and (t[qf] >= 72).any()


def test_abi_version_and_sized_create(host):
    """ABI 5: bsw_config has no size field, so a caller built against an older, shorter struct passes ITS sizeof to
    bsw_create_sized (fields it does not know keep their defaults); sizes that cannot be a bsw_config of any ABI are refused
    before anything touches the GPU; bsw_default_config turns the slot-thread pinning on; the Python mirror agrees with the
    header about the version."""
    import ctypes as C
    import re
    L = host.lib()
    hdr = open(os.path.join(ROOT, "include", "bwa_sw_mi355.h")).read()
    assert L.bsw_abi_version() == int(re.search(r"#define BSW_ABI_VERSION (\d+)", hdr).group(1)) == 6
    cfg = np.zeros(1, dtype=host.CONFIG)
    L.bsw_default_config(cfg.ctypes.data)
    assert cfg["pin_threads"][0] == 1 and cfg["timeout_ms"][0] == 0 and cfg["result_format"][0] == host.RESULT_FULL
    h = C.c_void_p()
    for bad in (0, 8, host.CONFIG.itemsize + 8, 4096):
        assert L.bsw_create_sized(cfg.ctypes.data, bad, C.byref(h)) == -2 and not h.value          # BSW_E_INVAL
    import torch
    if not torch.cuda.is_available():       # the accepted sizes get as far as the device check (ABI 3: 96 bytes, ABI 4/5: 104)
        for ok in (96, host.CONFIG.itemsize):
            assert L.bsw_create_sized(cfg.ctypes.data, ok, C.byref(h)) == -1 and not h.value        # BSW_E_NODEVICE


def test_default_config_leaves_the_watchdog_to_the_environment(host, monkeypatch):
    """bsw_default_config wrote timeout_ms = 120000, so BSW_TIMEOUT_MS was dead for every host that starts from
    the defaults.  Now 0 = the library default, resolved by bsw_effective_timeout_ms (what bsw_create_sized stores)."""
    cfg = np.zeros(1, dtype=host.CONFIG)
    host.lib().bsw_default_config(cfg.ctypes.data)
    assert int(cfg["timeout_ms"][0]) == 0
    monkeypatch.delenv("BSW_TIMEOUT_MS", raising=False)
    assert host.lib().bsw_effective_timeout_ms(cfg.ctypes.data) == 120000
    assert host.lib().bsw_effective_timeout_ms(None) == 120000
    monkeypatch.setenv("BSW_TIMEOUT_MS", "5000")
    assert host.lib().bsw_effective_timeout_ms(cfg.ctypes.data) == 5000
    monkeypatch.setenv("BSW_TIMEOUT_MS", "junk")
    assert host.lib().bsw_effective_timeout_ms(cfg.ctypes.data) == 120000
    cfg["timeout_ms"] = 250
    monkeypatch.setenv("BSW_TIMEOUT_MS", "5000")
    assert host.lib().bsw_effective_timeout_ms(cfg.ctypes.data) == 250          # explicit wins
    assert host.lib().bsw_abi_version() == 6


def test_auto_policy_by_work(host):
    """BSW_KERNEL_AUTO decides by the WORK per launched side (sum of query lengths): below 1.5 M bases the general kernels,
    up to 5 M bsw_lane2g_kernel (lane lists without the 16-bit seeds), above that the lane kernels (plan only: no GPU)."""
    p = host.default_params()
    t, a = host.synth_tasks(60000, seed=5)                               # 131-base right sides
    lane = lambda n: (lambda seg: int(seg[9] - seg[8]))(host.plan_batch(p, t[:n], kernel=host.KERNEL_AUTO)[1])
    assert lane(9000) == 0                                               # 1.18 M bases: general kernels
    assert lane(13000) == 13000                                          # 1.70 M: group kernel
    assert lane(45000) == 45000                                          # 5.9 M: lane kernels
    t2 = t[:20000].copy()
    t2["h0"][::10] = 300                                                 # 16-bit seeds in a group-sized chunk go to the wave classes
    seg = host.plan_batch(p, t2, kernel=host.KERNEL_AUTO)[1]
    assert int(seg[9] - seg[8]) == 18000 and int(seg[8] - seg[0]) == 2000
    big = t.copy(); big["h0"][::10] = 300
    seg = host.plan_batch(p, big, kernel=host.KERNEL_AUTO)[1]
    assert int(seg[9] - seg[8]) == 60000                                 # lane-kernel-sized: the 16-bit class keeps them


def test_auto_policy_fuses_the_two_sides_of_mid_sized_chunks(host):
    """A group-kernel chunk of two-sided seeds up to GROUP_FUSE_MAX (49 152) runs both sides of a seed in one launch: every 8-bit
    lane seed on the LEFT lists (the ones without a left side last in their class, at query length 0), no right list; the
    group kernel then starts at 1.5 M query bases over BOTH sides.  Up to LANE_FUSE_MAX (262 144) the lane kernels do the same
    (bsw_lane2_kernel's fused instantiation); larger chunks keep one list per side (plan only: no GPU)."""
    p = host.default_params()
    spec = dict(seed_len_min=19, seed_len_max=60, seed_at_start=0, junk_frac=0.05, n_rate=0.002)
    t, a = host.synth_tasks(70000, seed=6, **spec)
    t["lqlen"][7::50] = 0                                                # some seeds without a left side
    both = lambda x: int((x["lqlen"].astype(np.int64) + x["rqlen"]).sum())
    lists = lambda seg: (int(seg[9] - seg[8]), int(seg[17] - seg[9]), int(seg[25] - seg[17]))       # lane seeds, left entries, right entries
    n_small = 9000
    assert both(t[:n_small]) < 1_500_000
    assert lists(host.plan_batch(p, t[:n_small], kernel=host.KERNEL_AUTO)[1]) == (0, 0, 0)           # general kernels
    n_mid = 20000
    assert both(t[:n_mid]) >= 1_500_000 and max(int(t[:n_mid]["lqlen"].sum()), int(t[:n_mid]["rqlen"].sum())) < 1_500_000
    order, seg, _ = host.plan_batch(p, t[:n_mid], kernel=host.KERNEL_AUTO)
    has_side = int(((t[:n_mid]["lqlen"] > 0) | (t[:n_mid]["rqlen"] > 0)).sum())
    assert lists(seg) == (has_side, has_side, 0)                         # fused: one list, every seed once
    tm = t[:n_mid]

    def has_n(ptr, ln):
        return bool(ln) and bool((np.frombuffer(C.string_at(int(ptr), int(ln)), dtype=np.uint8) >= 4).any())
    either_n = np.array([has_n(tm["lquery"][i], tm["lqlen"][i]) or has_n(tm["rquery"][i], tm["rqlen"][i]) for i in range(n_mid)])
    sided = (tm["lqlen"] > 0) | (tm["rqlen"] > 0)
    # ... and a chunk that does not fill the machine keeps the seeds with an N in a query off the lane lists (bsw_binparams.nsplit:
    # a wavefront of them would be the launch's slowest): they sit on a list of their own behind the redo list, for the general
    # kernel, and the lane lists' unused tails hold 0xffffffff
    NONE = 0xffffffff
    left = order[seg[9]:seg[17]]
    kept = left[left != NONE]
    assert 0.1 < either_n.mean() < 0.5
    assert len(kept) == len(np.unique(kept)) == int((sided & ~either_n).sum()) and not either_n[kept].any()
    assert (left[len(kept):] == NONE).all()                              # the tail, nothing in between
    nlist_off = int(seg[25]) + has_side                                  # behind the redo list (one place per lane seed)
    nl = order[nlist_off:nlist_off + int((sided & either_n).sum())]
    assert sorted(nl.tolist()) == np.flatnonzero(sided & either_n).tolist()
    assert int(seg[26]) == nlist_off + has_side                          # order_len covers the N list
    ql = tm["lqlen"][kept].astype(np.int64)
    assert (np.diff(ql) <= 0).all()                                      # longest left side first, length 0 last
    n_big = 60000                                                        # past the group kernel's fused range: the lane kernels' fused launch (up to LANE_FUSE_MAX = 262 144)
    l, le, re_ = lists(host.plan_batch(p, t[:n_big], kernel=host.KERNEL_AUTO)[1])
    assert l > 49152 and le == l and re_ == 0
    l, le, re_ = lists(host.plan_batch(p, t[:n_big], kernel=host.KERNEL_LANE)[1])
    assert le == int((t[:n_big]["lqlen"] > 0).sum()) and re_ == int((t[:n_big]["rqlen"] > 0).sum())      # forced lane bins: a list per side
    t3, a3 = host.synth_tasks(300000, seed=7, **spec)
    l, le, re_ = lists(host.plan_batch(p, t3, kernel=host.KERNEL_AUTO)[1])
    assert l == 300000 - int(((t3["lqlen"] == 0) & (t3["rqlen"] == 0)).sum())
    assert le == int((t3["lqlen"] > 0).sum()) and re_ == int((t3["rqlen"] > 0).sum())                    # throughput-bound: a list per side


def test_n_split_follows_a_sample_of_the_chunk(host):
    """bsw_binparams.nsplit moves the lane seeds with an N in a query to the general kernel — one wavefront per seed — so it must
    not fire on a chunk of low-quality reads: the host pass samples the queries of 512 seeds and splits only while the N list
    would stay below NLIST_WORK_MAX (1.8 M query bases), for chunks of up to NSPLIT_MAX (262 144) lane seeds (plan only: no GPU)."""
    p = host.default_params()
    NONE = 0xffffffff
    spec = dict(seed_len_min=19, seed_len_max=60, seed_at_start=0, junk_frac=0.05)

    def holes(tasks):
        order, seg, _ = host.plan_batch(p, tasks, kernel=host.KERNEL_AUTO)
        return int((order[seg[9]:seg[25]] == NONE).sum())
    few, a1 = host.synth_tasks(200_000, seed=31, n_rate=0.0001, **spec)          # a sequencer's rate: ~2.6 % of the seeds, 5 k of 200 k
    assert 2_000 < holes(few) < 12_000
    many, a2 = host.synth_tasks(200_000, seed=32, n_rate=0.003, **spec)          # ~28 % of the seeds: 55 k would go one per wavefront
    assert holes(many) == 0
    assert holes(many[:40_000]) > 8_000                                          # ... 11 k of 40 k (1.2 M bases) still pay
    big, a3 = host.synth_tasks(270_000, seed=33, n_rate=0.0001, **spec)          # past NSPLIT_MAX: a list per side, no split
    assert holes(big) == 0
