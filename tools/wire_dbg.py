import json, os, sys, time
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import __graft_entry__ as graft
host = graft.load_package().host
p = host.default_params(zdrop=0)
nb = 128
tasks, arena = host.synth_tasks(nb * 819, seed=51, seed_len_min=19, seed_len_max=60, seed_at_start=0, junk_frac=0.05)
ins, outs, lo = [], [], 0
while lo < len(tasks) and len(ins) < nb:
    w, n = host.refbatch_encode(p, tasks[lo:lo + 819]); ins.append(w); outs.append(np.zeros(host.REFBATCH_OUT_WORDS, np.uint32)); lo += n
with host.BswContext(device=0) as c:
    for rep in range(3):
        t0 = time.perf_counter()
        for a, b in zip(ins, outs):
            c.refbatch_submit(a, b)
        c.refbatch_wait(0, 0)
        print("rep", rep, "ms", (time.perf_counter() - t0) * 1e3, file=sys.stderr)
