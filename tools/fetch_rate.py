#!/usr/bin/env python3
"""F3 measurement: upload time and PCIe payload with host-shipped targets vs device-fetched targets, 150 bp PE seeds
against a synthetic 64 Mbase reference; the fetch kernel itself is timed with rocprofv3 (profiles/)."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft
host = graft.load_package().host
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
rng = np.random.default_rng(1)
lp = 64_000_000
genome = rng.integers(0, 4, lp, dtype=np.uint8)
pac = host.pack_pac(genome)
p = host.default_params()
rl = 150
pos = rng.integers(500, lp - 500, n)
sl = rng.integers(19, 60, n)
qb = (rng.random(n) * (rl - sl + 1)).astype(np.int64)
reads = np.stack([genome[x:x + rl] for x in pos])           # forward-strand exact reads
rt = np.zeros(n, dtype=host.REF_TASK)
seeds = np.zeros(n, dtype=host.SEED)
seeds["rbeg"], seeds["qbeg"], seeds["len"] = pos + qb, qb, sl
rmax = np.zeros(2, dtype=np.int64)
L = host.lib()
for i in range(n):
    L.bsw_chain_window(p.ctypes.data, seeds[i:i + 1].ctypes.data, 1, rl, lp, rmax.ctypes.data)
    rt[i]["rmax0"], rt[i]["rmax1"] = rmax
rt["query"] = reads.ctypes.data + np.arange(n, dtype=np.uint64) * rl
rt["l_query"], rt["init_score"], rt["seed"], rt["tag"] = rl, -1, seeds, np.arange(n)
with host.BswContext(device=0) as ctx:
    ref = ctx.ref_upload(pac, lp)
    t0 = time.perf_counter(); got = ctx.extend_ref(p, ref, rt); t1 = time.perf_counter() - t0
    t0 = time.perf_counter(); got = ctx.extend_ref(p, ref, rt); t1 = time.perf_counter() - t0
    ctx.ref_free(ref)
ok = bool((got["score"] == 150).all() and (got["qb"] == 0).all() and (got["qe"] == rl - qb - sl).all())
print(json.dumps(dict(seeds=n, extend_ref_seconds=round(t1, 4), seeds_per_s=round(n / t1), exact_reads_end_to_end=ok)))
