import os, sys
import numpy as np
sys.path.insert(0, "/root/repo")
import __graft_entry__ as g
host = g.load_package().host; orc = g.load_oracle()
p = host.default_params(zdrop=0)
tasks, arena = host.synth_tasks(4096, seed=11, seed_len_min=19, seed_len_max=60, seed_at_start=0, sub_rate=0.02, indel_rate=0.01, junk_frac=0.15, n_rate=0.002)
with host.BswContext(device=0) as c:
    got = c.extend_pairs(p, tasks)
want = orc.pair_batch(p, tasks, nthreads=8)
bad = np.nonzero([got[i].tobytes() != want[i].tobytes() for i in range(len(tasks))])[0]
print(os.environ.get("BSW_LIB_PATH"), "bad", len(bad), "of", len(tasks))
for i in bad[:6]:
    print(i, "lq", tasks["lqlen"][i], "rq", tasks["rqlen"][i], "h0", tasks["h0"][i], "\n   got L", got["left"][i], "R", got["right"][i], "\n  want L", want["left"][i], "R", want["right"][i])
