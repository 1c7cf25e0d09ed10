#!/usr/bin/env python3
"""Rates of the paths either side of the headline bench: the reference wire format through
bsw_refbatch_submit/wait (F1) and banded global alignment with CIGAR (F4).  One JSON line each."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft
host = graft.load_package().host
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import _gen

def wire():
    p = host.default_params(zdrop=0)
    nb = 128
    tasks, arena = host.synth_tasks(nb * 819, seed=51, seed_len_min=19, seed_len_max=60, seed_at_start=0, junk_frac=0.05)
    ins, outs, lo = [], [], 0
    while lo < len(tasks) and len(ins) < nb:
        w, n = host.refbatch_encode(p, tasks[lo:lo + 819]); ins.append(w); outs.append(np.zeros(host.REFBATCH_OUT_WORDS, np.uint32)); lo += n
    res = []
    with host.BswContext(device=0) as c:
        for inflight in (1, 4, 32, 128):
            best = 1e9
            for rep in range(3):
                t0 = time.perf_counter()
                for k0 in range(0, len(ins), inflight):
                    for a, b in zip(ins[k0:k0 + inflight], outs[k0:k0 + inflight]):
                        c.refbatch_submit(a, b)
                    c.refbatch_wait(0, 0)
                best = min(best, time.perf_counter() - t0)
            res.append(dict(batches_in_flight=inflight, seeds=lo, seconds=round(best, 5), seeds_per_s=round(lo / best), ms_per_256KiB_batch=round(best / len(ins) * 1e3, 4)))
    print(json.dumps({"path": "F1 wire format, bsw_refbatch_submit/wait", "runs": res}), flush=True)

def glob():
    rng = np.random.default_rng(21)
    n = 400000
    ref = rng.integers(0, 4, 4_000_000).astype(np.uint8)
    reads = [_gen.mutate(rng, ref[s:s + 200], 150, 0.01, 0.004) for s in rng.integers(0, len(ref) - 400, 500)]
    starts = rng.integers(0, len(ref) - 400, n)
    ha = host.HostArena(n * 160 + len(ref) + 4096)
    ha.u8[:len(ref)] = ref
    off = len(ref)
    gt = np.zeros(n, dtype=host.GTASK)
    base = ha.u8.ctypes.data
    for k in range(500):
        ha.u8[off + 160 * k: off + 160 * k + 150] = reads[k]
    gt["query"] = base + off + 160 * (np.arange(n) % 500)
    gt["target"] = base + starts
    gt["qlen"], gt["tlen"] = 150, 150 + (np.arange(n) % 7)
    for w in (10, 25, 100):
        gt["w"] = w
        p = host.default_params()
        with host.BswContext(device=0) as c:
            c.global_batch(p, gt[:1000], max_cigar=32)
            t0 = time.perf_counter(); res, cig = c.global_batch(p, gt, max_cigar=32); dt = time.perf_counter() - t0
            t0 = time.perf_counter(); res2, _ = c.global_batch(p, gt, want_cigar=False); dt2 = time.perf_counter() - t0
        ncol = np.minimum(gt["qlen"], 2 * w + 1).astype(np.int64)
        cells = int((ncol * gt["tlen"]).sum())
        print(json.dumps({"path": "F4 ksw_global2 batch (150 bp reads)", "alignments": n, "band_w": w, "seconds_with_cigar": round(dt, 4),
                          "alignments_per_s": round(n / dt), "banded_cells_per_s_G": round(cells / dt / 1e9, 2),
                          "z_bytes_GB_per_s_end_to_end": round(cells / dt / 1e9, 2), "seconds_scores_only": round(dt2, 4),
                          "mean_cigar_ops": float(np.abs(res["n_cigar"]).mean()), "note": "wall time of the whole call: host layout + H2D + kernels + D2H"}), flush=True)

def align():
    """Mate-rescue shaped ksw_align2 calls: 150 bp reads against 400..800 bp windows that hold them (2 % errors), bwa's
    flags (KSW_XSUBO | KSW_XSTART | min_seed_len, 8-bit mode) and the same in 16-bit mode."""
    rng = np.random.default_rng(31)
    n = 200000
    ref = rng.integers(0, 4, 4_000_000).astype(np.uint8)
    origin = rng.integers(1000, len(ref) - 2000, 2000)
    reads = [_gen.mutate(rng, ref[s:s + 200], 150, 0.015, 0.004) for s in origin]
    ha = host.HostArena(2000 * 160 + len(ref) + 4096)
    ha.u8[:len(ref)] = ref
    off = len(ref)
    for k in range(2000):
        ha.u8[off + 160 * k: off + 160 * k + len(reads[k])] = reads[k]
    at = np.zeros(n, dtype=host.ATASK)
    base = ha.u8.ctypes.data
    rid = np.arange(n) % 2000
    at["query"] = base + off + 160 * rid
    at["qlen"] = np.array([len(r) for r in reads])[rid]
    at["tlen"] = rng.integers(400, 800, n)
    at["target"] = base + origin[rid] - rng.integers(0, 200, n)          # the window holds the read's origin (9 in 10: the rest rescue nothing)
    miss = rng.random(n) < 0.1
    at["target"][miss] = base + rng.integers(0, len(ref) - 1000, int(miss.sum()))
    p = host.default_params()
    for mode, flag in (("8-bit (KSW_XBYTE)", host.KSW_XBYTE), ("16-bit", 0)):
        at["xtra"] = flag | host.KSW_XSUBO | host.KSW_XSTART | 19
        with host.BswContext(device=0) as c:
            c.align_batch(p, at[:2000])
            t0 = time.perf_counter(); res = c.align_batch(p, at); dt = time.perf_counter() - t0
        cells = int((at["qlen"].astype(np.int64) * at["tlen"]).sum())
        print(json.dumps({"path": "F4 ksw_align2 batch (mate rescue shapes)", "mode": mode, "alignments": n, "seconds": round(dt, 4),
                          "alignments_per_s": round(n / dt), "first_pass_cells_per_s_G": round(cells / dt / 1e9, 1),
                          "with_start_pass": int((res["tb"] >= 0).sum()), "mean_score": float(res["score"].mean()),
                          "note": "wall time of the whole call: host layout + H2D + pack + kernels + D2H; cells = qlen x tlen of the first pass only"}), flush=True)


if __name__ == "__main__":
    which = sys.argv[1:] or ["wire", "glob", "align"]
    if "align" in which: align()
    if "wire" in which: wire()
    if "glob" in which: glob()
