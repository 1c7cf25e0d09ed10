#!/usr/bin/env python3
"""Four mid-sized submits in flight in ONE context (tickets), every submit ONE chunk — so each takes the fused launches and the N list
(chunks that do not fill the machine) while three others are on the GPU beside it; bytes and packed input, full and pair records;
every result batch against the oracle.  python tools/diag/mid_submit_stress.py [rounds]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft
host = graft.load_package().host
orc = graft.load_oracle()
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 30
p = host.default_params()
spec = dict(seed_len_min=19, seed_len_max=60, seed_at_start=0, sub_rate=0.02, indel_rate=0.01, junk_frac=0.1, n_rate=0.0005)
sizes = [20_000, 45_000, 80_000, 140_000]
sets = [host.synth_tasks(n, seed=900 + k, **spec) for k, n in enumerate(sizes)]
want = [orc.pair_batch_avx2(p, t, nthreads=16) for t, _ in sets]
packed = [host.pack_tasks(t) for t, _ in sets]
bad = 0
rng = np.random.default_rng(4)
for fmt in (host.RESULT_FULL, host.RESULT_PAIR):
    with host.BswContext(device=0, chunk_tasks=200_000, streams=4, result_format=fmt) as c:
        for r in range(rounds):
            outs, tks = [], []
            perm = rng.permutation(4)
            for k in perm:
                use_packed = bool(rng.integers(0, 2))
                outs.append((k, c.submit_packed(p, packed[k][0]) if use_packed else c.submit(p, sets[k][0])))
                tks.append(c.last_ticket)
            for j in rng.permutation(4):
                c.wait_ticket(tks[j])
                k, o = outs[j]
                if fmt == host.RESULT_FULL:
                    ok = o.tobytes() == want[k].tobytes()
                else:
                    ok = all((o[f] == want[k][f]).all() for f in ("tag", "qb", "qe", "rb", "re", "score", "truesc", "w"))
                if not ok:
                    bad += 1
                    print("MISMATCH round", r, "set", k, "format", fmt, flush=True)
print("mid-sized submit stress: %d rounds x 4 submits x 2 record formats, mismatches: %d" % (rounds, bad))
sys.exit(1 if bad else 0)
