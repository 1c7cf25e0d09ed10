#!/usr/bin/env python3
"""How much of the looped 232-column kernel's slot time is lost INSIDE its four-wave workgroups (a workgroup keeps its CU until
its slowest wave is done): libbwasw_wavelog.so stamps, 250 bp workload, one side's launch."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft
host = graft.load_package().host
host._LIB = os.path.join(os.path.dirname(host._LIB), "libbwasw_wavelog.so")
side = int(sys.argv[1]) if len(sys.argv) > 1 else 0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
n = 1000000
p = host.default_params(max_band_try=1, w=500)
tasks, arena = host.synth_tasks(n, seed=seed, read_len=250, seed_len_min=19, seed_len_max=40, seed_at_start=0, sub_rate=0.04, indel_rate=0.01, junk_frac=0.05, n_rate=0.0005, w=500)
with host.BswContext(device=0, kernel=2) as c:
    b = c.upload(p, tasks); c.run(b); c.sync(); c.run_history(); c.run(b); c.sync(); ms = c.run_history()
    res = c.download(b)
qn = "rqlen" if side else "lqlen"
m = tasks[qn] >= 136
r = res["right" if side else "left"][m]
t0, t1, hw = r["max_off"].astype(np.int64), r["aw"].astype(np.int64), r["cells"].astype(np.int64)
_, idx = np.unique((hw << 32) | t0, return_index=True)
t0, t1, hw = t0[idx], t1[idx], hw[idx]
cu = hw >> 8                                   # everything above wave_id / simd_id / pipe_id: cu, sh, se, xcc
start = (t0 - t0.min()) * 10e-6                # ms
dur = (t1 - t0) * 10e-6
# a workgroup = the (up to four) waves of one CU whose starts lie within 20 us of each other
order = np.lexsort((start, cu))
cu_s, st_s, du_s = cu[order], start[order], dur[order]
new = np.ones(len(order), bool)
new[1:] = (cu_s[1:] != cu_s[:-1]) | (st_s[1:] - st_s[:-1] > 0.02)
gid = np.cumsum(new) - 1
ng = gid.max() + 1
gmax = np.zeros(ng); np.maximum.at(gmax, gid, st_s + du_s)
gmin = np.full(ng, 1e9); np.minimum.at(gmin, gid, st_s)
gcnt = np.bincount(gid)
wave_ms = du_s.sum()
wg_ms = ((gmax - gmin) * gcnt).sum()
span = float((st_s + du_s).max())
print(json.dumps({"side": side, "seed": seed, "waves": int(len(du_s)), "workgroups": int(ng), "waves_per_workgroup": [int(x) for x in np.bincount(gcnt)],
                  "span_ms": round(span, 3), "wave_ms_over_1024": round(float(wave_ms / 1024), 3), "workgroup_ms_x4_over_1024": round(float(wg_ms / 1024), 3),
                  "lost_inside_workgroups": round(float(1 - wave_ms / wg_ms), 4), "kernel_ms_events": ms}))
