#!/usr/bin/env python3
"""Section timing of the lane2 row loop from a `make stamp STAMP_MODE=2` build (libbwasw_stamp.so): cycles per wave spent in
loop top, wave reductions, match words, cell blocks, row tails — per side, for any bench.py workload.
Usage: l2_stamps.py [seeds] [workload] [min side length]"""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["BSW_LIB_PATH"] = os.path.join(ROOT, "bwa-mem-sw_amd", "libbwasw_stamp.so")     # read by host.py when it is imported
import __graft_entry__ as graft
import bench
pkg = graft.load_package()
host = pkg.host
assert host.lib_path().endswith("libbwasw_stamp.so")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
wl = sys.argv[2] if len(sys.argv) > 2 else "150bp_w100_single_bin"
qmin = int(sys.argv[3]) if len(sys.argv) > 3 else 0          # only sides of at least this many bases (250 bp: 136 = the looped kernel's class, `make stampl`)
spec = dict(bench.WORKLOADS[wl])
p = host.default_params(w=spec["w"], max_band_try=1)      # (no second band try: the stamped records would send every seed to the redo list)
tasks, arena = host.synth_tasks(n, seed=1000, **spec)
with host.BswContext(device=0, kernel=2) as c:
    b = c.upload(p, tasks)
    for _ in range(3):
        c.run(b)
    c.sync(); c.run_history(); c.run(b); c.sync(); ms = c.run_history()
    res = c.download(b)
names = ["loop top + row_begin", "reductions", "match words + consts", "cell blocks", "row tails"]
out = {"workload": wl, "seeds": n, "kernel_ms": ms}
for side, qf in (("left", "lqlen"), ("right", "rqlen")):
    sel = tasks[qf] > qmin
    if not sel.any():
        continue
    r = res[side][sel]
    vals = [float((r[f].astype(np.float64) * 16).mean()) for f in ("score", "qle", "tle", "gtle", "gscore")]
    tot = sum(vals)
    out[side] = {"cycles_per_wave_total": round(tot), "sections": {k: [round(v), round(v / tot, 3)] for k, v in zip(names, vals)}}
print(json.dumps(out))
