#!/usr/bin/env python3
"""Per-wave durations of the looped 232-column kernel on the 250 bp workload's LEFT launch (libbwasw_wavelog.so: make variantl
NAME=wavelog L2_EXTRA=-DBSW_L2L_WAVELOG): waves with / without query Ns, by query length."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft
host = graft.load_package().host
host._LIB = os.path.join(os.path.dirname(host._LIB), "libbwasw_wavelog.so")
n = 1000000
nr = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0005
side = int(sys.argv[2]) if len(sys.argv) > 2 else 0          # 0 left launch, 1 right launch
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
p = host.default_params(max_band_try=1, w=500)
tasks, arena = host.synth_tasks(n, seed=seed, read_len=250, seed_len_min=19, seed_len_max=40, seed_at_start=0, sub_rate=0.04, indel_rate=0.01, junk_frac=0.05, n_rate=nr, w=500)
with host.BswContext(device=0, kernel=2) as c:
    b = c.upload(p, tasks); c.run(b); c.sync(); c.run_history(); c.run(b); c.sync(); ms = c.run_history()
    res = c.download(b)
qn = "rqlen" if side else "lqlen"
m = tasks[qn] >= 136                                       # the 232-column class's sides of that launch
r = res["right" if side else "left"][m]
t0, t1, hw, nb = r["max_off"].astype(np.int64), r["aw"].astype(np.int64), r["cells"].astype(np.int64), r["gscore"]
ql = tasks[qn][m]
_, idx = np.unique((hw << 32) | t0, return_index=True)
t0, t1, nb, ql = t0[idx], t1[idx], nb[idx], ql[idx]
dur = (t1 - t0) * 10e-9 * 1e3                               # ms
start = (t0 - t0.min()) * 10e-9 * 1e3
out = {"side": side, "seed": seed, "n_rate": nr, "kernel_ms_events": ms, "waves": int(len(dur)), "span_ms": round(float((t1.max() - t0.min()) * 10e-6), 3),
       "n_waves": int((nb > 0).sum()), "mean_n_blocks_in_n_waves": round(float(nb[nb > 0].mean()), 1) if (nb > 0).any() else 0}
for name, sel in (("plain", nb == 0), ("with_N", nb > 0)):
    rows = []
    for q0 in range(136, 232, 24):
        s = sel & (ql >= q0) & (ql < q0 + 24)
        if s.any():
            rows.append([q0, int(s.sum()), round(float(np.median(dur[s])), 3), round(float(np.median(start[s])), 3)])
    out[name + "__qlen__waves__median_ms__median_start_ms"] = rows
end = start + dur
out["waves_running_at_span_minus_ms"] = {str(x): int(((start <= end.max() - x) & (end > end.max() - x)).sum()) for x in (0.05, 0.2, 0.4, 0.6, 0.8, 1.0, 1.5, 2.0)}
last = np.argsort(-end)[:8]
out["latest_ending_waves__start_ms__dur_ms__qlen__n_blocks"] = [[round(float(start[i]), 3), round(float(dur[i]), 3), int(ql[i]), int(nb[i])] for i in last]
out["wave_ms_sum_over_1024_slots"] = round(float(dur.sum() / 1024), 3)
print(json.dumps(out))
