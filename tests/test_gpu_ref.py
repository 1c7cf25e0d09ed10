"""Seeds against a device-resident 2-bit reference: the GPU fetches the extension targets itself
(bns_get_seq semantics, both strands, left side reversed).  The result batch must equal, byte for byte, the one
obtained from host-side extraction (F2 glue) — on the GPU and on the CPU oracle."""
import numpy as np
import pytest

from test_gpu_parity import assert_same

pytestmark = pytest.mark.gpu


def _reads_and_seeds(host, rng, genome, n, rl=150):
    lp = len(genome)
    both = np.concatenate([genome, 3 - genome[::-1]])
    reads, seeds = [], np.zeros(n, dtype=host.SEED)
    for i in range(n):
        strand = i % 2
        pos = int(rng.integers(400, lp - 400)) + strand * lp
        read = both[pos:pos + rl].copy()
        sl = int(rng.integers(19, 60))
        qb = int(rng.integers(0, rl - sl + 1)) if i % 11 else 0
        if i % 13 == 0:
            qb = rl - sl                                                    # no right side
        for x in np.nonzero(rng.random(rl) < 0.02)[0]:                      # substitutions / N outside the seed
            if not (qb <= x < qb + sl):
                read[x] = 4 if rng.random() < 0.1 else (read[x] + 1 + rng.integers(0, 3)) % 4
        if i % 7 == 0 and qb + sl + 20 < rl:                                # a deletion in the read (right flank)
            cut = int(rng.integers(qb + sl + 5, rl - 10))
            read = np.concatenate([read[:cut], read[cut + 3:], both[pos + rl:pos + rl + 3]])
        reads.append(read.astype(np.uint8))
        seeds[i] = (pos + qb, qb, sl)
    return reads, seeds


@pytest.mark.parametrize("kernel", [0, 1, 2])
def test_device_fetch_equals_host_extraction(host, oracle, kernel):
    rng = np.random.default_rng(100 + kernel)
    lp = 50000
    genome = rng.integers(0, 4, lp).astype(np.uint8)
    pac = host.pack_pac(genome)
    n = 3000
    reads, seeds = _reads_and_seeds(host, rng, genome, n)
    p = host.default_params()
    tasks, keep = host.seeds_to_tasks(p, pac, lp, reads, seeds)             # host-side extraction (F2)
    want = oracle.pair_batch(p, tasks, nthreads=8)
    rt = np.zeros(n, dtype=host.REF_TASK)
    rmax = np.zeros(2, dtype=np.int64)
    qkeep = []
    for i in range(n):
        q = np.ascontiguousarray(reads[i])
        qkeep.append(q)
        host.lib().bsw_chain_window(p.ctypes.data, seeds[i:i + 1].ctypes.data, 1, len(q), lp, rmax.ctypes.data)
        rt[i]["query"], rt[i]["l_query"], rt[i]["init_score"] = q.ctypes.data, len(q), -1
        rt[i]["seed"] = seeds[i]
        rt[i]["rmax0"], rt[i]["rmax1"], rt[i]["tag"] = rmax[0], rmax[1], i
    with host.BswContext(device=0, kernel=kernel) as ctx:
        ref = ctx.ref_upload(pac, lp)
        got = ctx.extend_ref(p, ref, rt)
        assert_same(got, want, tasks)
        assert_same(ctx.extend_pairs(p, tasks), want, tasks)                # same seeds, targets shipped from the host
        # errors, not fallbacks
        bad = rt[:1].copy()
        bad["rmax0"], bad["rmax1"] = lp - 5, lp + 5                         # bridges the strands
        with pytest.raises(host.BswError):
            ctx.extend_ref(p, ref, bad)
        ctx.ref_free(ref)
    aln = host.results_to_alnregs(seeds, got)
    assert ((aln["re"] - aln["rb"]) > 0).all() and (aln["qe"] > aln["qb"]).all()


@pytest.mark.parametrize("mode", ["pageable", "registered", "two devices"])
def test_streaming_submit_against_the_resident_reference(host, oracle, mode):
    """bsw_submit_ref: the streaming form of F3.  Reads in pageable memory (left flanks mirrored by the gather),
    in registered memory (DMA'd as they are, mirrored by the pack kernel), and over two slot sets of one context."""
    rng = np.random.default_rng(7)
    lp = 80000
    genome = rng.integers(0, 4, lp).astype(np.uint8)
    pac = host.pack_pac(genome)
    n = 24000
    reads, seeds = _reads_and_seeds(host, rng, genome, n)
    p = host.default_params()
    tasks, keep = host.seeds_to_tasks(p, pac, lp, reads, seeds)
    want = oracle.pair_batch(p, tasks, nthreads=8)
    total = sum(len(r) for r in reads)
    arena = host.HostArena(total + 64) if mode != "pageable" else None
    buf = arena.u8 if arena else np.zeros(total + 64, np.uint8)
    rt = np.zeros(n, dtype=host.REF_TASK)
    rmax = np.zeros(2, dtype=np.int64)
    off = 0
    for i in range(n):
        q = reads[i]
        buf[off:off + len(q)] = q
        host.lib().bsw_chain_window(p.ctypes.data, seeds[i:i + 1].ctypes.data, 1, len(q), lp, rmax.ctypes.data)
        rt[i]["query"], rt[i]["l_query"], rt[i]["init_score"] = buf.ctypes.data + off, len(q), -1
        rt[i]["seed"] = seeds[i]
        rt[i]["rmax0"], rt[i]["rmax1"], rt[i]["tag"] = rmax[0], rmax[1], i
        off += len(q)
    kw = dict(devices=[0, 0], streams=2) if mode == "two devices" else dict(device=0, streams=3)
    try:
        with host.BswContext(kernel=host.KERNEL_LANE, chunk_tasks=5000, **kw) as ctx:
            ref = ctx.ref_upload(pac, lp)
            got = ctx.submit_ref(p, ref, rt)
            ctx.wait()
            assert_same(got, want, tasks)
            assert_same(ctx.extend_ref(p, ref, rt[:777]), want[:777])
            bad = rt.copy()
            bad["rmax1"][n // 2] = 2 * lp + 10                                  # one bad window in the middle: an error, not a fallback
            with pytest.raises(host.BswError):
                ctx.submit_ref(p, ref, bad)
                ctx.wait()
            ctx.ref_free(ref)
    finally:
        if arena:
            arena.free()


def test_synthetic_genome_reads_through_the_streaming_path(host, oracle):
    """bsw_synth_ref_generate (the bench's e2e_device_reference workload) with seeds anywhere in the read:
    bsw_submit_ref on its tasks == the oracle on the host-extracted tasks (bns_get_seq + reversal on the CPU)."""
    p = host.default_params()
    lp, n, L = 300_000, 30_000, 150
    arena = host.HostArena(n * L + 64)
    try:
        pac, rt, _ = host.synth_ref_tasks(n, lp, p, arena=arena.u8, seed=11, read_len=L, seed_len_min=19, seed_len_max=60,
                                          seed_at_start=0, sub_rate=0.02, indel_rate=0.004, n_rate=0.002, junk_frac=0.05)
        reads = [arena.u8[i * L:(i + 1) * L] for i in range(n)]
        tasks, keep = host.seeds_to_tasks(p, pac, lp, reads, rt["seed"].copy())
        want = oracle.pair_batch(p, tasks, nthreads=8)
        with host.BswContext(device=0, chunk_tasks=8192, streams=3) as ctx:
            ref = ctx.ref_upload(pac, lp)
            got = ctx.submit_ref(p, ref, rt)
            ctx.wait()
            assert_same(got, want, tasks)
            ctx.ref_free(ref)
    finally:
        arena.free()


@pytest.mark.parametrize("n", [30_000, 70_000])
def test_one_chunk_against_the_resident_reference_takes_the_mid_sized_paths(host, oracle, n):
    """bsw_submit_ref as ONE chunk under BSW_KERNEL_AUTO: the chunk does not fill the machine, so both sides of a seed run in one
    launch (group kernel at 30 000 seeds, lane kernel at 70 000) and the seeds with an N in a query go to the general kernel —
    decided from a sample of the reads, whose left queries sit FORWARDS in the caller's memory in this mode (the host pass reads
    them backwards from the seed).  Same results as the oracle on the host-extracted tasks."""
    p = host.default_params()
    lp, L = 400_000, 150
    arena = host.HostArena(n * L + 64)
    try:
        pac, rt, _ = host.synth_ref_tasks(n, lp, p, arena=arena.u8, seed=17, read_len=L, seed_len_min=19, seed_len_max=60,
                                          seed_at_start=0, sub_rate=0.02, indel_rate=0.004, n_rate=0.001, junk_frac=0.05)
        reads = [arena.u8[i * L:(i + 1) * L] for i in range(n)]
        tasks, keep = host.seeds_to_tasks(p, pac, lp, reads, rt["seed"].copy())
        want = oracle.pair_batch(p, tasks, nthreads=8)
        with host.BswContext(device=0, chunk_tasks=n, streams=2) as ctx:
            ref = ctx.ref_upload(pac, lp)
            got = ctx.submit_ref(p, ref, rt)
            ctx.wait()
            assert_same(got, want, tasks)
            ctx.ref_free(ref)
    finally:
        arena.free()
