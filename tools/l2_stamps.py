#!/usr/bin/env python3
"""Section timing of the lane2 row loop from a -DBSW_L2_STAMP build (libbwasw_stamp.so): cycles per wave spent in
loop top, wave reductions, match words, cell blocks, row tails."""
import ctypes as C, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft
pkg = graft.load_package()
host = pkg.host
host._LIB = os.path.join(os.path.dirname(host._LIB), "libbwasw_stamp.so")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
p = host.default_params()
tasks, arena = host.synth_tasks(n, seed=1000)
with host.BswContext(device=0, kernel=2) as c:
    b = c.upload(p, tasks); c.run(b); c.sync(); c.run_history(); c.run(b); c.sync(); ms = c.run_history()
    res = c.download(b)
r = res["right"]
names = ["loop top + row_begin", "reductions", "match words + consts", "cell blocks", "row tails"]
tot = 0
vals = []
for f in ("score", "qle", "tle", "gtle", "gscore"):
    v = r[f].astype(np.float64) * 16
    vals.append(v.mean()); tot += v.mean()
print(json.dumps({"kernel_ms": ms, "cycles_per_wave_total": tot, "sections": {k: [round(v), round(v / tot, 3)] for k, v in zip(names, vals)}}))
