#!/usr/bin/env python3
"""Soak of the lane-launch chain: resident 250 bp batches (chain of four launches) re-run hundreds of times, interleaved with mixed
PE bins (no chain) and synchronous chunks on the same stream, in contexts of 4, 2 and 1 slot streams (rotation over 4 / 2
streams / chain off); every run's records must equal the first run's, and the first run's the oracle's on a sample."""
import os, sys, time, zlib
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as graft
import bench
host, orc = graft.load_package().host, graft.load_oracle()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
n = 200000
sp250, spmix = dict(bench.WORKLOADS["250bp_w500"]), dict(bench.WORKLOADS["150bp_w100_mixed_bins"])
p250, pmix = host.default_params(zdrop=100, w=sp250["w"]), host.default_params(zdrop=100, w=spmix["w"])
t250, a1 = host.synth_tasks(n, seed=4242, **sp250)
tmix, a2 = host.synth_tasks(n, seed=4243, **spmix)
want = orc.pair_batch(p250, t250[:20000], nthreads=8)
for streams in (4, 2, 1):
    t0 = time.time()
    with host.BswContext(device=0, streams=streams) as c:
        b250, bmix = c.upload(p250, t250), c.upload(pmix, tmix)
        c.run(b250); c.run(bmix); c.sync()
        r250, rmix = c.download(b250), c.download(bmix)
        assert r250[:20000].tobytes() == want.tobytes(), "250 bp batch differs from the oracle"
        k250, kmix = zlib.crc32(r250.tobytes()), zlib.crc32(rmix.tobytes())
        for i in range(reps):
            c.run(b250)
            if i % 3 == 0:
                c.run(bmix)
            if i % 50 == 49:
                got = c.extend_pairs(p250, t250[:60000])            # a synchronous chunk between resident runs
                assert got[:20000].tobytes() == want.tobytes(), ("sync chunk", i)
                assert zlib.crc32(c.download(b250).tobytes()) == k250, ("250", i)
                assert zlib.crc32(c.download(bmix).tobytes()) == kmix, ("mixed", i)
                print("streams %d: %d runs ok (%.1f s)" % (streams, i + 1, time.time() - t0), flush=True)
        b250.free(); bmix.free()
print("chain soak ok")
