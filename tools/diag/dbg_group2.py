import os, sys
import numpy as np
sys.path.insert(0, "/root/repo")
import __graft_entry__ as g
host = g.load_package().host; orc = g.load_oracle()
p = host.default_params(zdrop=0)
rng = np.random.default_rng(1)
q = rng.integers(0, 4, 100).astype(np.uint8)
t = np.concatenate([q, rng.integers(0, 4, 60).astype(np.uint8)])
seeds = [dict(lq=q, lt=t, rq=[], rt=[], h0=30) for _ in range(16)]
tasks, arena = host.make_tasks(seeds)
with host.BswContext(device=0) as c:
    got = c.extend_pairs(p, tasks)
want = orc.pair_batch(p, tasks, nthreads=1)
print("got ", got["left"][0]); print("want", want["left"][0])
