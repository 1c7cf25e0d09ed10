#!/usr/bin/env python3
"""Slot occupancy over time of one lane2 launch, from a -DBSW_L2_STAMP build (libbwasw_stamp.so): every wave reports its start /
end (s_memrealtime, 10 ns ticks) and where it ran (HW_ID, XCC_ID).  Prints how many waves were resident in each 2 % slice
of the launch and per-wave durations — where a launch's tail is, and what it costs."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft
pkg = graft.load_package()
host = pkg.host
assert host.lib_path().endswith("libbwasw_stamp.so"), "run with BSW_LIB_PATH=.../libbwasw_stamp.so (make stamp STAMP_MODE=1)"
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
mixed = len(sys.argv) > 2 and sys.argv[2] == "mixed"
p = host.default_params(max_band_try=1)
if mixed:
    smin, smax = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (19, 60)
    tasks, arena = host.synth_tasks(n, seed=2000, seed_len_min=smin, seed_len_max=smax, seed_at_start=0, junk_frac=0.05, n_rate=0.0005)
else:
    tasks, arena = host.synth_tasks(n, seed=1000)
with host.BswContext(device=0, kernel=2) as c:
    b = c.upload(p, tasks); c.run(b); c.sync(); c.run_history(); c.run(b); c.sync(); ms = c.run_history()
    res = c.download(b)
# (mixed: the LEFT launch — in this build the left kernel's records hold stamps, not scores, so the right launch starts from garbage h0)
side, qf = ("left", "lqlen") if mixed else ("right", "rqlen")
r = res[side][tasks[qf] > 0]
t0, t1, hw = r["max_off"].astype(np.int64), r["aw"].astype(np.int64), r["cells"].astype(np.int64)
pro, ql = r["gscore"].astype(np.int64), tasks[qf][tasks[qf] > 0]
key = (hw << 32) | t0
_, idx = np.unique(key, return_index=True)               # one record per wave (all 128 seeds of a wave carry the same stamps)
t0, t1, hw, pro, ql = t0[idx], t1[idx], hw[idx], pro[idx], ql[idx]
base = t0.min()
t0, t1 = (t0 - base) * 10e-9, (t1 - base) * 10e-9        # seconds
T = t1.max()
slices = 50
dur = t1 - t0
occ = []
for k in range(slices):
    a, b_ = T * k / slices, T * (k + 1) / slices
    occ.append(float((np.clip(np.minimum(t1, b_) - np.maximum(t0, a), 0, None)).sum() / (b_ - a)))
simd = hw & 0xfffffff0                                      # drop the wave-slot bits: (xcc, se, sh, cu, pipe, simd)
# duration by start time (10 buckets) and by how much of the wave's life another wave shared its SIMD
order = np.argsort(t0)
bk = []
for part in np.array_split(order, 10):
    bk.append([round(float(t0[part].min()) * 1e3, 3), round(float(np.median(dur[part])) * 1e3, 4), round(float(dur[part].max()) * 1e3, 4)])
share = np.zeros(len(t0))
for sid in np.unique(simd):
    m = np.nonzero(simd == sid)[0]
    for a in m:
        ov = np.clip(np.minimum(t1[m], t1[a]) - np.maximum(t0[m], t0[a]), 0, None).sum() - dur[a]
        share[a] = ov / dur[a]
cls = {}
for lo, hi in ((0, 0.1), (0.1, 0.5), (0.5, 0.9), (0.9, 1.01), (1.01, 9)):
    m = (share >= lo) & (share < hi)
    if m.any():
        cls["shared %.1f-%.1f" % (lo, hi)] = [int(m.sum()), round(float(np.median(dur[m])) * 1e3, 4)]
print(json.dumps({"waves": int(len(t0)), "kernel_ms_events": ms, "span_ms": round(T * 1e3, 4),
                  "mean_resident_waves": round(float(dur.sum() / T), 1), "slots": 2048,
                  "resident_waves_per_2pct_slice": [round(x) for x in occ],
                  "wave_ms": {"min": round(float(dur.min()) * 1e3, 4), "p10": round(float(np.percentile(dur, 10)) * 1e3, 4), "median": round(float(np.median(dur)) * 1e3, 4),
                              "p90": round(float(np.percentile(dur, 90)) * 1e3, 4), "max": round(float(dur.max()) * 1e3, 4)},
                  "distinct_simds": int(len(np.unique(simd))), "prologue_us": {"median": round(float(np.median(pro)) * 0.01, 2), "p90": round(float(np.percentile(pro, 90)) * 0.01, 2), "max": round(float(pro.max()) * 0.01, 2)},
                  "qlen__waves__median_wave_us": [[int(q0), int(((ql >= q0) & (ql < q0 + 16)).sum()), round(float(np.median(dur[(ql >= q0) & (ql < q0 + 16)])) * 1e6, 1)] for q0 in range(0, 144, 16) if ((ql >= q0) & (ql < q0 + 16)).any()], "start_ms__median_ms__max_ms_by_start_decile": bk, "waves__median_ms_by_simd_sharing": cls, "start_ms_of_last_wave": round(float(t0.max()) * 1e3, 4)}))
