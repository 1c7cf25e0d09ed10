#!/usr/bin/env python3
"""The reference's wire format at queue depths 1 .. 256 against the oracle, batch by batch (round 5: result batches written on
the device, the batches DMA'd ahead of the parse, the big groups of a deep queue enqueued side by side on host threads).  Per round: a random
queue depth, random gap penalties / band in the header (changing inside the queue in some rounds), ragged task counts, empty batches,
task batches in registered memory (back to back or with gaps), pageable memory or a mix, both variants; every pair-level field of
every record and the zero tail of every 16 KiB result batch."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft
host = graft.load_package().host
oracle = graft.load_oracle()
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 30
F = ["tag", "qb", "qe", "rb", "re", "score", "truesc", "w"]
W = host.REFBATCH_IN_WORDS
pool, arena = host.synth_tasks(256 * 819, seed=4242, seed_len_min=19, seed_len_max=60, seed_at_start=0, junk_frac=0.08, n_rate=0.001, indel_rate=0.01)
reg = host.HostArena(2 * 256 * W * 4)
rview = reg.view(np.uint32, 2 * 256 * W).reshape(2 * 256, W)
OW = host.REFBATCH_OUT_WORDS
oreg = host.HostArena(3 * 256 * OW * 4)                      # result batches in registered memory: DMA'd where they are
oview = oreg.view(np.uint32, 3 * 256 * OW).reshape(3 * 256, OW)   # slots 0..255 back to back, 256.. every other one
tot = nb_tot = 0
with host.BswContext(device=0) as c:
    for r in range(rounds):
        rng = np.random.default_rng(7000 + r)
        depth = int(rng.choice([1, 2, 15, 16, 17, 40, 95, 96, 97, 128, 129, 160, 200, 255, 256]))
        variant = int(rng.integers(0, 2))
        mem = str(rng.choice(["registered", "registered_gaps", "pageable", "mixed"]))
        def header():
            return host.default_params(zdrop=0, variant=variant, o_del=int(rng.integers(1, 9)), e_del=int(rng.integers(1, 4)), o_ins=int(rng.integers(1, 9)),
                                       e_ins=int(rng.integers(1, 4)), w=int(rng.choice([20, 60, 100])), pen_clip5=int(rng.choice([0, 5])), pen_clip3=int(rng.choice([0, 5])))
        pa, pb = header(), header()
        switch = int(rng.integers(0, depth + 1)) if rng.random() < 0.4 else depth       # header changes at batch `switch`
        ins, outs, meta, lo = [], [], [], int(rng.integers(0, 4000))
        for k in range(depth):
            p = pa if k < switch else pb
            cnt = 0 if rng.random() < 0.04 else int(rng.choice([819, 819, 819, int(rng.integers(1, 820))]))
            if cnt:
                words, n = host.refbatch_encode(p, pool[lo:lo + cnt])
            else:
                words, n = np.zeros(W, np.uint32), 0
                ref = host.refbatch_encode(p, pool[:1])[0]
                words[0], words[1] = ref[0], ref[1]
            where = mem if mem != "mixed" else str(rng.choice(["registered", "pageable"]))
            if where.startswith("registered"):
                slot = 2 * k if where == "registered_gaps" else k
                rview[slot] = words; words = rview[slot]
            omem = str(rng.choice(["pageable", "registered", "registered_gaps"])) if k == 0 or rng.random() < 0.1 else omem
            if omem == "pageable":
                o = np.full(host.REFBATCH_OUT_WORDS, 0xdeadbeef, np.uint32)
            else:
                o = oview[256 + 2 * k if omem == "registered_gaps" else k]; o[:] = 0xdeadbeef
            ins.append(words); outs.append(o); meta.append((lo, n, p)); lo += n
        for a, b in zip(ins, outs):
            c.refbatch_submit(a, b)
        assert c.refbatch_wait(variant, 0) == depth
        wa = oracle.pair_batch(pa, pool[:lo], nthreads=16)
        wb = oracle.pair_batch(pb, pool[:lo], nthreads=16) if switch < depth else wa
        for k, ((l0, n, p), o) in enumerate(zip(meta, outs)):
            want = (wa if k < switch else wb)[l0:l0 + n]
            got = host.refbatch_decode_results(o, n)
            want16 = host.refbatch_decode_results(host.refbatch_encode_results(want), n)          # the record's 16-bit fields
            for f in F:
                if not (got[f] == want16[f]).all():
                    print("MISMATCH round", r, "batch", k, f, depth, mem, variant, flush=True); sys.exit(1)
            if not (o[5 * n:] == 0).all():
                print("TAIL NOT ZERO round", r, "batch", k, flush=True); sys.exit(1)
        tot += lo - meta[0][0]; nb_tot += depth
        print("round", r, "ok: depth", depth, mem, "variant", variant, "header switch at", switch, "-", nb_tot, "batches,", tot, "seeds so far", flush=True)
print("wire-format fuzz ok:", nb_tot, "task batches,", tot, "seeds: every record field and every result batch's zero tail identical to the oracle's")
