#!/usr/bin/env python3
"""Rate of bsw_submit_ref (seeds against a device-resident 2-bit reference; only the reads cross PCIe) on the bench
workload's shape: synthetic genome, 150 bp reads with one 19 bp seed at base 0.  Prints one JSON line per setting."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft
host = graft.load_package().host
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
lp = (int(sys.argv[2]) if len(sys.argv) > 2 else 64) * 1_000_000
p = host.default_params()
ha = host.HostArena(150 * n + 4096)
ho = host.HostArena(n * host.RESULT.itemsize)
pac, rt, _ = host.synth_ref_tasks(n, lp, p, arena=ha.u8, seed=3000, read_len=150, seed_len_min=19, seed_len_max=19, seed_at_start=1)
obuf = ho.view(host.RESULT, n)

def cells_of(res):
    return int(res["left"]["cells"].astype(np.int64).sum() + res["right"]["cells"].astype(np.int64).sum())

for streams, chunk in ((4, 131072), (4, 65536), (3, 131072), (6, 131072), (4, 262144)):
    with host.BswContext(device=0, streams=streams, chunk_tasks=chunk) as ctx:
        ref = ctx.ref_upload(pac, lp)
        ctx.submit_ref(p, ref, rt, out=obuf); ctx.wait()
        runs = []
        for _ in range(5):
            t0 = time.perf_counter()
            ctx.submit_ref(p, ref, rt, out=obuf); ctx.wait()
            runs.append(time.perf_counter() - t0)
        dt = float(np.median(runs))
        ctx.ref_free(ref)
    print(json.dumps(dict(streams=streams, chunk_tasks=chunk, seconds=round(dt, 5), seeds_per_s=round(n / dt), gcups=round(cells_of(obuf) / dt / 1e9, 1))), flush=True)

# steady state: a stream of 1M-seed batches, two in flight through two contexts (each with its own copy of the reference)
ho2 = host.HostArena(n * host.RESULT.itemsize)
obuf2 = ho2.view(host.RESULT, n)
for streams, chunk in ((2, 131072), (2, 262144), (3, 131072), (4, 131072)):
    a = host.BswContext(device=0, streams=streams, chunk_tasks=chunk)
    b = host.BswContext(device=0, streams=streams, chunk_tasks=chunk)
    ra, rb = a.ref_upload(pac, lp), b.ref_upload(pac, lp)
    a.submit_ref(p, ra, rt, out=obuf); a.wait(); b.submit_ref(p, rb, rt, out=obuf2); b.wait()
    reps = 8
    t0 = time.perf_counter()
    a.submit_ref(p, ra, rt, out=obuf); b.submit_ref(p, rb, rt, out=obuf2)
    for _ in range(reps - 1):
        a.wait(); a.submit_ref(p, ra, rt, out=obuf)
        b.wait(); b.submit_ref(p, rb, rt, out=obuf2)
    a.wait(); b.wait()
    dt = (time.perf_counter() - t0) / (2 * reps)
    print(json.dumps(dict(mode="stream of submits, two in flight", streams_per_ctx=streams, slot_threads=2 * streams, chunk_tasks=chunk,
                          seconds_per_batch=round(dt, 5), seeds_per_s=round(n / dt), gcups=round(cells_of(obuf) / dt / 1e9, 1),
                          identical=bool(obuf.tobytes() == obuf2.tobytes()))), flush=True)
    a.ref_free(ra); b.ref_free(rb); a.close(); b.close()
