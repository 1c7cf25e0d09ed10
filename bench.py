#!/usr/bin/env python3
"""bench.py — GCUPS of the seed-extension hot path on N MI355X GPUs (one process per GPU).

A "step" = one pass of the hot path over one GPU's seeds, inputs resident in HBM as they crossed PCIe (byte per base): the
device side of the batch manager (pack: bytes -> 16 bases per uint64; bin: the counting sort into the launch lists) and
the DP kernels (left + right ksw_extend2 with band retry and the mem_chain2aln decision) — bsw_run_staged.  At N=1 the
workload is the configuration BASELINE.json's metric is quoted on: 150 bp PE seeds, mixed (qlen, tlen) bins through the batch
manager, configs[2] size (10 M seeds per GPU, resident batches of 32 x 128 Ki seeds).  --scaling weak (default): every rank
owns its own seeds of that shape.  --scaling strong: ONE pool of --pool seeds is cut into chunks, chunk c belongs to rank
c mod N (the per-read task shard of SURVEY.md §8e); no data-path collective either way.

Prints ONE JSON line on rank 0.  `value` = DP cells actually evaluated (exactly as the CPU algorithm iterates them) / wall
time / 1e9, summed over ranks, WITH the staging kernels in the step; `kernels_only` = the same batches through bsw_run (the DP
kernels alone: what rounds 1-5 printed as `value`).  `roofline` prices the dominant DP kernel over the DP part of the step
(HIP events on the library's stream split every step into pack + bin / DP).
`e2e*` = the same seeds pushed through bsw_submit* (host buffers in, host buffers out) — PCIe-inclusive, never `value`; run
on every rank (timed between barriers, MAX over ranks), with what the host side cost (bsw_host_stats: CPU seconds of the slot
threads per million seeds) and how many cores 8 GPUs would need at that rate.
`other_workloads` (N=1): configs[1] (150 bp single bin, 1 M seeds) and configs[4]'s shape (250 bp, w = 500, 1 M seeds).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

WORKLOADS = {
    # BASELINE.json configs[1]
    "150bp_w100_single_bin": dict(read_len=150, seed_len_min=19, seed_len_max=19, seed_at_start=1,
                                  sub_rate=0.01, indel_rate=0.001, junk_frac=0.0, n_rate=0.0, w=100),
    # configs[2]/[3] shape: PE mixed bins via the batch manager
    "150bp_w100_mixed_bins": dict(read_len=150, seed_len_min=19, seed_len_max=60, seed_at_start=0,
                                  sub_rate=0.01, indel_rate=0.001, junk_frac=0.05, n_rate=0.0005, w=100),
    # configs[4] shape: 250 bp, ~5 % error, w=500
    "250bp_w500": dict(read_len=250, seed_len_min=19, seed_len_max=40, seed_at_start=0,
                       sub_rate=0.04, indel_rate=0.01, junk_frac=0.05, n_rate=0.0005, w=500),
}

VALU_OPS_PER_CELL = 15          # integer VALU ops of one DP cell
# The binding roof is VALU issue (integer max/add DP: ~0.02 B per cell, nothing is a contraction).
# Peak from MI355X_MICROARCH.md: 157.3 TFLOP/s fp32 vector = 256 CUs x 4 SIMD-32 x 32 lanes x 2.4 GHz = 78.6 T lane-ops/s
# (a wave64 op at the full rate holds its SIMD for 2 cycles).  Measured on gfx950 (profiles/r3/ubench_fetch_align.txt):
# every VALU opcode the DP cell needs — packed 16-bit ones included — issues at 2 + 2.1 / W cycles per instruction and SIMD
# with W waves resident, so 2 cycles is the many-wave limit; `roofline.issue_ceiling` prices the kernel against what ITS
# occupancy (two waves for a 136-register row) can issue.
PEAK_VALU_TOPS = 256 * 4 * 32 * 2.4e9 / 1e12
PEAK_HBM_GBS = 8000.0
PMC_FILE = os.path.join(ROOT, "profiles", "pmc_latest.json")   # rocprofv3 --pmc summary of this same command


def side_paths(host, device):
    """The paths either side of the extension kernels, one short measurement each, outside the timed
    region: wall time of the whole library call (host layout + H2D + kernels + D2H), inputs in registered host memory."""
    rng = np.random.default_rng(5)
    p = host.default_params()
    res = {}
    n, L = 100_000, 150
    ref = rng.integers(0, 4, 2_000_000).astype(np.uint8)
    starts = rng.integers(1000, len(ref) - 2000, n)
    reads = ref[starts[:, None] + np.arange(L)]
    reads = np.where(rng.random(reads.shape) < 0.02, (reads + rng.integers(1, 4, reads.shape)) % 4, reads).astype(np.uint8)
    arena = host.HostArena(len(ref) + n * 152 + 4096)
    arena.u8[:len(ref)] = ref
    arena.u8[len(ref):len(ref) + n * 152].reshape(n, 152)[:, :L] = reads
    base = arena.u8.ctypes.data
    with host.BswContext(device=device) as c:
        # F4: bwa ksw_align2, mate-rescue shapes (150 bp mate in a 600 bp window that holds it, bwa's flags, 8-bit mode)
        at = np.zeros(n, dtype=host.ATASK)
        at["query"], at["qlen"] = base + len(ref) + 152 * np.arange(n), L
        at["target"], at["tlen"] = base + starts - 200, 600
        at["xtra"] = host.KSW_XBYTE | host.KSW_XSUBO | host.KSW_XSTART | 19
        c.align_batch(p, at[:2000])
        t0 = time.perf_counter(); r = c.align_batch(p, at); dt = time.perf_counter() - t0
        res["ksw_align2"] = {"alignments_per_s": round(n / dt), "first_pass_gcups": round(n * L * 600 / dt / 1e9, 1), "mean_score": round(float(r["score"].mean()), 1)}
        # F4: bwa ksw_global2 with CIGAR, w = 25
        gt = np.zeros(n, dtype=host.GTASK)
        gt["query"], gt["qlen"] = at["query"], L
        gt["target"], gt["tlen"], gt["w"] = base + starts, L, 25
        c.global_batch(p, gt[:2000], max_cigar=32)
        t0 = time.perf_counter(); g, _ = c.global_batch(p, gt, max_cigar=32); dt = time.perf_counter() - t0
        res["ksw_global2"] = {"alignments_per_s": round(n / dt), "band_w": 25, "mean_score": round(float(g["score"].mean()), 1)}
        # F1: the reference's 256 KiB task batches, 128 queued per wait (the figure of the earlier rounds) and 256 (the most the
        # API keeps in flight)
        pz = host.default_params(zdrop=0)
        NB = 256
        wt, _ = host.synth_tasks(NB * 819, seed=51, seed_len_min=19, seed_len_max=60, seed_at_start=0, junk_frac=0.05)
        # the task batches sit back to back in registered (DMA-able) host memory, as the reference's host keeps them in its
        # pinned workspace (batch_manager.v:745-773): the library DMAs them where they are
        warena = host.HostArena(NB * host.REFBATCH_IN_WORDS * 4)
        wview = warena.view(np.uint32, NB * host.REFBATCH_IN_WORDS).reshape(NB, host.REFBATCH_IN_WORDS)
        oarena = host.HostArena(NB * host.REFBATCH_OUT_WORDS * 4)         # ... and the result batches go back into registered memory (rbb.v:150-166)
        oview = oarena.view(np.uint32, NB * host.REFBATCH_OUT_WORDS).reshape(NB, host.REFBATCH_OUT_WORDS)
        ins, outs, ends, lo = [], [], [], 0
        while lo < len(wt) and len(ins) < NB:
            w, k = host.refbatch_encode(pz, wt[lo:lo + 819]); wview[len(ins)] = w; ins.append(wview[len(ins)]); outs.append(oview[len(outs)]); lo += k; ends.append(lo)
        rate = {}
        for nb in (128, len(ins)):
            best = 1e9
            for _ in range(4):
                t0 = time.perf_counter()
                for a, b in zip(ins[:nb], outs[:nb]):
                    c.refbatch_submit(a, b)
                c.refbatch_wait(0, 0)
                best = min(best, time.perf_counter() - t0)
            rate[nb] = round(ends[nb - 1] / best)
        res["wire_format"] = {"seeds_per_s": rate[128], "batches_in_flight": 128, "seeds_per_s_256_in_flight": rate[len(ins)],
                              "task_batches_in": "registered host memory (DMA'd where they are)", "result_batches": "written on the device, DMA'd into registered host memory"}
        # (a spot check that the direct path delivers: the first and the last result batch decode to the tags that went in)
        for bi in (0, len(ins) - 1):
            n_b = ends[bi] - (ends[bi - 1] if bi else 0)
            got = host.refbatch_decode_results(outs[bi], n_b)
            assert (got["tag"] == wt["tag"][ends[bi] - n_b:ends[bi]]).all(), "wire format: result batch %d does not carry its tasks' tags" % bi
        oarena.free()
        warena.free()
        # batches that do not fill the machine (what a host that hands over a few thousand to a few hundred thousand seeds at a time gets;
        # the wire format's own operating point): resident PE mixed-bin batches through BSW_KERNEL_AUTO, median of 8 runs each, reads with
        # the headline workload's N rate.  A launch lasts as long as its slowest wavefront there: both sides of a seed run in ONE launch,
        # queries with an N go to the general kernel beside it
        mid = {}
        mspec = dict(WORKLOADS["150bp_w100_mixed_bins"])
        mt, marena = host.synth_tasks(262144, seed=77, **mspec)
        pm = host.default_params(w=mspec["w"])
        for nmid in (4096, 16384, 32768, 65536, 131072, 262144):
            b = c.upload(pm, mt[:nmid])
            for _ in range(2):
                c.run(b)
            c.sync(); c.run_history()
            for _ in range(8):
                c.run(b)
            c.sync()
            ms = float(np.median(c.run_history()))
            r = c.download(b)
            nl = int(b.info()["launches"])
            b.free()
            mid[str(nmid)] = {"ms": round(ms, 4), "gcups": round(cells_of(r) / ms / 1e6, 1), "kernel_launches": nl}
        res["mid_sized_batches"] = {"workload": "150bp_w100_mixed_bins, resident, BSW_KERNEL_AUTO, DP kernels of one batch (bsw_run)", "by_seeds": mid}
        del marena
    arena.free()
    return res


PRESETS = {
    # BASELINE.json configs[3]: 100M 150 bp PE reads, per-read task shard over the ranks, one pool (strong scaling)
    "configs3": dict(workload="150bp_w100_mixed_bins", scaling="strong", pool=100_000_000),
    # BASELINE.json configs[4]: 250 bp PE reads at ~5 % error, w=500 (weak: every rank owns a batch)
    "configs4": dict(workload="250bp_w500", scaling="weak"),
}


def rank_cpu_sets(n, avail=None, sys_root="/sys"):
    """One CPU set per rank: the CPUs this process may use, split among the ranks along NUMA nodes where the machine has
    several (>= 6x at 8 GPUs needs NUMA-local host threads; the reference's single manager sits next to its
    4 PE arrays, batch_manager.v:343-348).  With a PCI bus id per GPU (amdgpu sysfs, no GPU call) rank r gets the CPUs of
    ITS card's node; without, the nodes are dealt round-robin.  Returns a list of sorted CPU lists (None = leave alone).
    `avail` / `sys_root`: the CPUs to deal out and the sysfs tree to read (tests hand in a fake tree)."""
    if avail is None:
        try:
            avail = sorted(os.sched_getaffinity(0))
        except AttributeError:
            return [None] * n
    avail = sorted(avail)
    nodes = []
    try:
        for d in sorted(os.listdir(sys_root + "/devices/system/node")):
            if d.startswith("node") and d[4:].isdigit():
                cpus = parse_cpulist(open(sys_root + "/devices/system/node/%s/cpulist" % d).read())
                cpus = [c for c in cpus if c in set(avail)]
                if cpus:
                    nodes.append((int(d[4:]), cpus))
    except OSError:
        pass
    if not nodes:
        nodes = [(0, avail)]
    gpu_node = gpu_numa_nodes(sys_root)
    by_node = {}
    for r in range(n):
        node = gpu_node[r] if r < len(gpu_node) and gpu_node[r] in dict(nodes) else nodes[r % len(nodes)][0]
        by_node.setdefault(node, []).append(r)
    sets = [None] * n
    for node, ranks in by_node.items():
        cpus = dict(nodes)[node]
        per = max(1, len(cpus) // len(ranks))
        for k, r in enumerate(ranks):
            mine = cpus[k * per:(k + 1) * per] if k * per < len(cpus) else cpus
            sets[r] = mine or cpus
    return sets


def parse_cpulist(text):
    out = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        out.extend(range(int(lo), int(hi or lo) + 1))
    return out


def gpu_numa_nodes(sys_root="/sys"):
    """NUMA node of every amdgpu render node in enumeration order, from sysfs (no HIP call: the launcher must not touch the
    GPU).  [] when the layout is unknown; a node of -1 (the kernel does not say) is kept and dealt round-robin by the caller.
    This ORDER is an assumption — HIP may enumerate the cards differently — which every rank checks against its own device's
    PCI address once it may touch the GPU (verify_rank_placement)."""
    out = []
    try:
        cards = sorted((d for d in os.listdir(sys_root + "/class/drm") if d.startswith("renderD")), key=lambda d: int(d[7:]))
        for c in cards:
            dev = os.path.realpath(sys_root + "/class/drm/%s/device" % c)
            drv = os.path.basename(os.path.realpath(dev + "/driver")) if os.path.exists(dev + "/driver") else ""
            if drv != "amdgpu":
                continue
            out.append(int(open(dev + "/numa_node").read().strip()))
    except (OSError, ValueError):
        return []
    return out


def self_launch(n):
    """One fresh child process per GPU (RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* set as torchrun would), rank 0's stdout relayed.
    The parent never initialises the GPU and never execs: the reference's one manager feeding several PE arrays
    (batch_manager.v:343-348) becomes one launcher feeding N per-GPU ranks, each pinned to the CPUs of its share
    (BSW_RANK_CPUS, applied by the child before it imports torch).  All children are polled against one deadline; the
    first rank that fails takes the others down.  Exit code: non-zero if any rank failed."""
    import socket
    import subprocess
    import threading
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    sets = rank_cpu_sets(n)
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if sets[r] and "BSW_RANK_CPUS" not in os.environ:
            env["BSW_RANK_CPUS"] = ",".join(str(c) for c in sets[r])
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    deadline = time.time() + float(os.environ.get("BSW_LAUNCH_TIMEOUT", "900"))
    rcs = [None] * n
    while any(c is None for c in rcs):
        for r, p in enumerate(procs):
            if rcs[r] is None:
                rcs[r] = p.poll()
        failed = any(c not in (None, 0) for c in rcs)
        if failed or time.time() > deadline:
            for r, p in enumerate(procs):       # a rank that died before the rendezvous leaves the others in a collective
                if rcs[r] is None:
                    p.kill()
                    p.wait()
                    rcs[r] = -9
            break
        time.sleep(0.05)
    reader.join(timeout=5)
    sys.stdout.write(b"".join(chunks).decode())
    sys.stdout.flush()
    bad = [(r, c) for r, c in enumerate(rcs) if c != 0]
    if bad:
        print("bench.py: ranks failed (rank, exit code): %s" % bad, file=sys.stderr)
        return 1
    return 0


def apply_rank_affinity():
    """BSW_RANK_CPUS (set by self_launch, or by whoever starts the ranks) -> this process's CPU affinity, before torch and
    the library start their threads.  Returns the CPU list in effect."""
    spec = os.environ.get("BSW_RANK_CPUS")
    try:
        global ORIG_AFFINITY
        ORIG_AFFINITY = sorted(os.sched_getaffinity(0))
        if spec:
            os.sched_setaffinity(0, set(parse_cpulist(spec)))
        return sorted(os.sched_getaffinity(0))
    except (AttributeError, OSError, ValueError):
        return None


ORIG_AFFINITY = None


def device_bdf(props):
    """torch's device properties -> the PCI address sysfs uses ("0000:c1:00.0")"""
    return "%04x:%02x:%02x.0" % (props.pci_domain_id, props.pci_bus_id, props.pci_device_id)


def verify_rank_placement(bdf, orig_affinity, node_peers, sys_root="/sys"):
    """The launcher pinned this rank by the ORDER of the render nodes in sysfs, before anything could ask HIP which card
    `local_rank` is.  Now the rank knows its card's PCI address: read that card's NUMA node and CPUs, and if the CPUs the
    rank runs on are not the card's, re-pin onto the card's share of what the process was allowed before the launcher
    narrowed it.  node_peers = (my index, how many) among the ranks whose cards sit on the same node.
    Returns (numa_node, repinned, cpus in effect)."""
    node, local = -1, []
    try:
        node = int(open("%s/bus/pci/devices/%s/numa_node" % (sys_root, bdf)).read().strip())
        local = parse_cpulist(open("%s/bus/pci/devices/%s/local_cpulist" % (sys_root, bdf)).read())
    except (OSError, ValueError):
        pass
    try:
        now = sorted(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        return node, False, None
    if not local or all(c in set(local) for c in now):
        return node, False, now                       # already next to the card (or nothing known about it)
    cand = [c for c in (orig_affinity or now) if c in set(local)]
    if not cand:
        return node, False, now                       # the process may not run there at all
    k, m = node_peers
    per = max(1, len(cand) // max(m, 1))
    mine = cand[k * per:(k + 1) * per] or cand
    try:
        os.sched_setaffinity(0, set(mine))
    except OSError:
        return node, False, now
    return node, True, mine


def cgroup_cpu_quota():
    """CPUs' worth of time the cgroup grants this process (cgroup v2 cpu.max, v1 cfs quota), or None when unlimited / unknown"""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else float(q) / float(per)
    except (OSError, ValueError):
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else q / per
    except (OSError, ValueError):
        return None


def kernel_source_hash():
    """sha256 (16 hex digits) over the kernel and batch-manager sources the library is built from: identifies the code a
    counter file was collected on (tools/make_pmc_latest.py stores it, pmc_summary() compares it)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "bwa-mem-sw_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(d, "*.hip")) + glob.glob(os.path.join(d, "*.h")) + glob.glob(os.path.join(d, "*.c"))
                    + glob.glob(os.path.join(d, "*.inc")) + glob.glob(os.path.join(d, "Makefile"))):
        h.update(os.path.basename(f).encode() + b"\0" + open(f, "rb").read())
    return h.hexdigest()[:16]


def issue_ceiling(pmc, cells, kavg_ms, waves=2):
    """What W waves per SIMD can issue at all: a stream of independent VALU instructions — packed 16-bit and 32-bit alike —
    runs at 2 + 2.1 / W cycles per instruction and SIMD with W waves resident (tools/ubench/fetch_align.hip,
    profiles/r3/ubench_fetch_align.txt: 4.18 / 3.05 / 2.69 / 2.52 cycles at W = 1 / 2 / 3 / 4) — the guide's 2 cycles (the
    `peak`) need many waves; a 136-register row allows two, a 72-register row three, a 232-register row one.
    The kernel's own figure from the counters of the same kernel sources (None without them)."""
    if not pmc.get("valu_lane_insts_per_cell") or not pmc.get("clock_ghz"):
        return None
    per_simd = cells * pmc["valu_lane_insts_per_cell"] / 64.0 / (kavg_ms * 1e-3) / 1024.0        # VALU instructions / s / SIMD
    cyc = pmc["clock_ghz"] * 1e9 / per_simd
    ceil = 2.0 + 2.1 / waves
    return {"waves_per_simd": waves, "ceiling_cycles_per_valu_inst": round(ceil, 2), "kernel_cycles_per_valu_inst": round(cyc, 3),
            "frac_of_ceiling": round(ceil / cyc, 3), "law": "2 + 2.1 / W cycles per VALU instruction and SIMD", "source": "tools/ubench/fetch_align.hip"}


def cpu_list_str(cpus):
    """[0,1,2,3,8,9] -> '0-3,8-9'"""
    if not cpus:
        return None
    out, lo, prev = [], cpus[0], cpus[0]
    for c in list(cpus[1:]) + [None]:
        if c is not None and c == prev + 1:
            prev = c
            continue
        out.append(str(lo) if lo == prev else "%d-%d" % (lo, prev))
        if c is not None:
            lo = prev = c
    return ",".join(out)


def pmc_summary(workload, tasks, key=None):
    """(counters, source) of the committed rocprofv3 PMC passes of this same command (profiles/pmc_latest.json).  The counters
    come from an EARLIER run of this command under rocprofv3 --pmc, not from the run that prints them: they are quoted only
    when the file was collected on the same workload AND the same kernel sources as this tree; otherwise ({}, why)."""
    src = {"file": "profiles/pmc_latest.json", "collected": "earlier rocprofv3 --pmc passes of this command (tools/profile.sh), not this run"}
    try:
        j = json.load(open(PMC_FILE))
        j = j if "workload" in j else (j.get(key) or j.get(workload, {}))        # one entry per workload (round 4: a single entry); key: a pass at another batch size
    except Exception:
        return {}, dict(src, status="absent")
    src["source_hash"] = j.get("source_hash")
    src["tree_hash"] = kernel_source_hash()
    if j.get("workload") != workload or (tasks is not None and j.get("seeds_per_gpu") != tasks):
        return {}, dict(src, status="other workload: counters omitted")
    if j.get("source_hash") != src["tree_hash"]:
        return {}, dict(src, status="stale (kernel sources changed since the counters were collected): counters omitted")
    return j, dict(src, status="matches this tree")


def spread(runs, n):
    """min / median / max of the timed repetitions of a PCIe-inclusive leg, as seeds per second (best first)"""
    r = sorted(runs)
    return {"seeds_per_s_max_median_min": [round(n / r[0], 1), round(n / float(np.median(r)), 1), round(n / r[-1], 1)],
            "max_over_min": round(r[-1] / r[0], 3), "reps": len(r), "reps_ms_in_order": [round(x * 1e3, 3) for x in runs]}


def cells_of(res):
    return int(res["left"]["cells"].astype(np.int64).sum() + res["right"]["cells"].astype(np.int64).sum())


def generate_seeds(host, spec, chunk_ids, sizes, seed_of, threads, arena=None, tag_base=None, chunk=131072):
    """The seeds of the given chunks, generated side by side on `threads` CPUs (every chunk has its own generator seed;
    bsw_synth_generate is 4.25 us per mixed-bin seed on one core) straight into ONE registered (DMA-able) arena.
    Returns (tasks, arena)."""
    from concurrent.futures import ThreadPoolExecutor
    bounds = [host.synth_arena_bound(sz, **spec) for sz in sizes]
    offs = np.concatenate([[0], np.cumsum(bounds)]).astype(np.int64)
    if arena is None:
        arena = host.HostArena(int(offs[-1]) + 4096)
    assert arena.nbytes >= int(offs[-1])
    starts = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    tg = np.zeros(int(starts[-1]), dtype=host.TASK)

    def one(k):
        t, _ = host.synth_tasks(sizes[k], arena=arena.u8[int(offs[k]):], seed=seed_of(chunk_ids[k]), **spec)      # (ctypes releases the GIL)
        if tag_base is not None:
            t["tag"] = np.arange(chunk_ids[k] * chunk, chunk_ids[k] * chunk + sizes[k], dtype=np.uint32)
        else:
            t["tag"] = np.arange(int(starts[k]), int(starts[k]) + sizes[k], dtype=np.uint32)
        tg[int(starts[k]):int(starts[k]) + sizes[k]] = t

    with ThreadPoolExecutor(max_workers=max(1, threads)) as pool:
        for f in [pool.submit(one, k) for k in range(len(sizes))]:
            f.result()
    # The chunks were generated side by side into slots of their arena BOUND (3.2x what a PE chunk uses): close the gaps, so that
    # the arena is as compact as one a host would fill sequentially — bsw_submit DMAs a chunk's span as it lies only when the
    # span holds little besides the chunk's sequences (<= 2x), and gathers through pinned staging otherwise.
    base = arena.u8.ctypes.data
    run = 0
    for k in range(len(sizes)):
        a, b = int(starts[k]), int(starts[k]) + sizes[k]
        t = tg[a:b]
        end = 0
        for pf, lf in (("lquery", "lqlen"), ("ltarget", "ltlen"), ("rquery", "rqlen"), ("rtarget", "rtlen")):
            m = t[pf] != 0
            if m.any():
                end = max(end, int((t[pf][m] + t[lf][m].astype(np.uint64)).max()))
        lo = base + int(offs[k])
        used = max(end - lo, 0)
        shift = int(offs[k]) - run
        if shift and used:
            arena.u8[run:run + used] = arena.u8[int(offs[k]):int(offs[k]) + used]          # (slots never overlap what is already packed: run <= offs[k])
            for pf in ("lquery", "ltarget", "rquery", "rtarget"):
                m = t[pf] != 0
                t[pf][m] -= np.uint64(shift)
        run += (used + 63) & ~63
    return tg, arena


def sides_of(tasks):
    return int((tasks["lqlen"] > 0).sum() + (tasks["rqlen"] > 0).sum())


def nominal_of(tasks):
    return int((tasks["lqlen"].astype(np.int64) * tasks["ltlen"]).sum() + (tasks["rqlen"].astype(np.int64) * tasks["rtlen"]).sum())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--tasks", type=int, default=10_000_000, help="seeds per GPU per step (weak scaling); BASELINE configs[2]: 10 M")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--pool", type=int, default=8_000_000, help="--scaling strong: seeds in the one pool all ranks share (BASELINE configs[3]: 100000000)")
    ap.add_argument("--gen-threads", type=int, default=0, help="threads that generate this rank's chunks (0 = the rank's CPU set, at most 32)")
    ap.add_argument("--sample-stride", type=int, default=97, help="--scaling strong: the line carries the exact cell count of every k-th chunk of rank 0")
    ap.add_argument("--resident-chunks", type=int, default=32, help="128Ki-seed chunks per resident batch (a batch holds < 4 GiB of bases)")
    ap.add_argument("--workload", default="150bp_w100_mixed_bins", choices=sorted(WORKLOADS))
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--zdrop", type=int, default=100)
    ap.add_argument("--gaps", default=None, help="o_del,e_del,o_ins,e_ins (default: bwa's 6,1,6,1)")
    ap.add_argument("--kernels-only-steps", type=int, default=-1, help="steps of the second timed region (bsw_run: DP kernels alone); -1 = --steps, 0 = skip")
    ap.add_argument("--cpu-sample", type=int, default=250_000, help="seeds timed on the scalar CPU oracle (rank 0, N=1): 3 runs of ~9 s on 16 threads")
    ap.add_argument("--cpu-threads", type=int, default=0, help="0 = min(affinity, 16): the 1-GPU box's CPU share; an all-core leg is added when the affinity holds more")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the single-bin / 250 bp / side-path measurements (N=1 only, outside the timed region)")
    ap.add_argument("--other-seeds", type=int, default=1_000_000, help="seeds of the other_workloads legs")
    ap.add_argument("--packed-chunk", type=int, default=0, help="seeds per chunk of the packed-input legs (0 = the library's choice: sized by the seeds' work, bsw_config.chunk_tasks = 0)")
    ap.add_argument("--e2e-chunk", type=int, default=0, help="seeds per chunk of the byte-input and device-reference legs (0 = the library's choice)")
    ap.add_argument("--e2e-slots", type=int, default=4, help="slot threads (= streams) of the PCIe-inclusive legs")
    ap.add_argument("--e2e-pack-threads", type=int, default=8, help="bsw_config.pack_threads of the PCIe-inclusive legs: a chunk's host pass runs on 1 + this / slots threads")
    ap.add_argument("--no-e2e", action="store_true", help="skip the bsw_submit (PCIe-inclusive) measurements")
    ap.add_argument("--e2e-reps", type=int, default=5, help="bsw_submit passes timed (median reported)")
    ap.add_argument("--stream-reps", type=int, default=8, help="batches of a stream leg (two submits in flight in ONE context)")
    ap.add_argument("--ref-mbp", type=int, default=64, help="synthetic genome size (Mbp) of the device-resident-reference e2e leg; 0 = skip")
    ap.add_argument("--check", type=int, default=100_000, help="seeds checked bit-exact against the oracle after timing")
    ap.add_argument("--spec", action="append", default=[], help="override a generator field, e.g. --spec n_rate=0 (experiments)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend for the barrier / timing reduction (nccl = RCCL; gloo only to rehearse N>1 on a 1-GPU box)")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal: every rank uses device 0 (needs --backend gloo)")
    ap.add_argument("--kernel", type=int, default=0, help="0 auto (batch manager picks per bin), 1 wave-per-task only, 2 force lane bins")
    ap.add_argument("--dry-run", action="store_true", help="launcher check without a GPU: the ranks rendezvous over gloo, count themselves and rank 0 prints the count")
    ap.add_argument("--preset", default=None, choices=sorted(PRESETS),
                    help="BASELINE.json configs[3] / configs[4] as one flag (sets --workload/--scaling/--pool; explicit flags still win)")
    args = ap.parse_args()
    if args.preset:
        given = {a.split("=")[0] for a in sys.argv[1:] if a.startswith("--")}
        for key, val in PRESETS[args.preset].items():
            if "--" + key not in given:
                setattr(args, key, val)
    if args.kernels_only_steps < 0:
        args.kernels_only_steps = args.steps

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without torchrun: start N fresh per-GPU ranks BEFORE anything here touches the GPU
        sys.exit(self_launch(args.gpus))

    cpu_affinity = apply_rank_affinity()             # before torch / the library start threads
    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if args.dry_run:
        seen = 1
        if world > 1:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group(backend="gloo")
            v = torch.ones(1, dtype=torch.float64)
            dist.all_reduce(v)
            seen = int(v.item())
            dist.barrier()
            dist.destroy_process_group()
        if rank == 0:
            print(json.dumps({"dry_run": True, "n_gpus": world, "ranks_seen": seen, "gpus_flag": args.gpus,
                              "scaling": args.scaling, "workload": args.workload, "pool": args.pool,
                              "rank_cpus": os.environ.get("BSW_RANK_CPUS")}), flush=True)
        return
    if args.share_gpu:
        local_rank = 0
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend="gloo")
    assert torch.cuda.is_available(), "bench.py needs a GPU (the library has no CPU path)"
    torch.cuda.set_device(local_rank)
    red_dev = "cuda" if args.backend == "nccl" else "cpu"

    # ---- is this rank next to ITS card?  (the launcher assumed render-node order = HIP order; bwa_mem_sw.v:162 /
    # batch_manager.v:343-348: one manager next to its arrays) ----
    props = torch.cuda.get_device_properties(local_rank)
    try:
        my_bdf = device_bdf(props)
    except AttributeError:                           # a torch build whose device properties carry no PCI address: nothing to check
        class _NoPci:
            pci_domain_id = pci_bus_id = pci_device_id = 0
        props, my_bdf = _NoPci(), "unknown"
    try:
        my_node = int(open("/sys/bus/pci/devices/%s/numa_node" % my_bdf).read().strip())
    except (OSError, ValueError):
        my_node = -1
    peers = (0, 1)
    if dist is not None:
        nt = torch.tensor([float(my_node)], dtype=torch.float64, device=red_dev)
        alln = [torch.zeros_like(nt) for _ in range(world)]
        dist.all_gather(alln, nt)
        same = [r for r in range(world) if int(alln[r].item()) == my_node]
        peers = (same.index(rank), len(same))
    _node, repinned, cpus_now = verify_rank_placement(my_bdf, ORIG_AFFINITY, peers)
    if repinned:
        cpu_affinity = cpus_now

    pkg = graft.load_package()
    host = pkg.host
    spec = dict(WORKLOADS[args.workload])
    for kv in args.spec:
        key, val = kv.split("=")
        spec[key] = type(spec[key])(float(val))
    gaps = dict(zip(("o_del", "e_del", "o_ins", "e_ins"), (int(x) for x in args.gaps.split(",")))) if args.gaps else {}
    params = host.default_params(variant=args.variant, zdrop=args.zdrop, w=spec["w"], **gaps)
    gen_threads = max(1, min(args.gen_threads or len(cpu_affinity or [0]), 32))

    # ---- this rank's seeds, generated straight into pinned (DMA-able) host memory, then resident in HBM AS BYTES ----
    chunk = 131072
    ctx = host.BswContext(device=local_rank, kernel=args.kernel)
    t_gen0 = time.perf_counter()
    sample_cells, groups = {}, []
    if args.scaling == "strong":
        # one pool, chunk c -> rank c mod N; a rank's chunks become resident batches of <= --resident-chunks chunks, generated,
        # uploaded and dropped on the host one group at a time (group g+1 is generated while group g uploads)
        from concurrent.futures import ThreadPoolExecutor
        nchunks = (args.pool + chunk - 1) // chunk
        mine = [c for c in range(nchunks) if c % world == rank]
        groups = [mine[i:i + args.resident_chunks] for i in range(0, len(mine), args.resident_chunks)]
        gb = args.resident_chunks * host.synth_arena_bound(chunk, **spec) + 4096
        arenas = [host.HostArena(gb) for _ in range(2 if len(groups) > 1 else 1)]
        bg = ThreadPoolExecutor(max_workers=1)

        def gen_group(gi):
            grp = groups[gi]
            return generate_seeds(host, spec, grp, [min(chunk, args.pool - c * chunk) for c in grp], lambda c: 5000 + c, gen_threads,
                                  arena=arenas[gi % len(arenas)], tag_base=0, chunk=chunk)[0]

        batches, n_local, gstats, tasks = [], 0, [], None
        nxt = bg.submit(gen_group, 0) if groups else None
        for gi in range(len(groups)):
            tg = nxt.result()
            nxt = bg.submit(gen_group, gi + 1) if gi + 1 < len(groups) else None     # ... overlaps this group's upload
            batches.append(ctx.upload_raw(params, tg))                                 # inputs resident in HBM before the timed region
            n_local += len(tg)
            gstats.append((sides_of(tg), nominal_of(tg)))
            if rank == 0 and len(groups) > 1:
                print("bench.py: rank 0 resident batch %d/%d (%d seeds, %.1f s)" % (len(batches), len(groups), len(tg), time.perf_counter() - t_gen0),
                      file=sys.stderr, flush=True)
            if len(groups) == 1:
                tasks = tg                                                              # e2e legs only when the rank's share is one batch
        bg.shutdown()
        harena = arenas[0]
        for ar in arenas[1:]:
            ar.free()
    else:
        n_local = args.tasks
        nchunks = max(1, (n_local + chunk - 1) // chunk)
        ids = list(range(nchunks))
        tasks, harena = generate_seeds(host, spec, ids, [min(chunk, n_local - c * chunk) for c in ids],
                                       lambda c: 7000 + 100000 * rank + c, gen_threads)
        per = args.resident_chunks * chunk
        batches, gstats = [], []
        for lo in range(0, max(n_local, 1), per):
            tg = tasks[lo:lo + per]
            batches.append(ctx.upload_raw(params, tg))          # inputs resident in HBM before the timed region
            gstats.append((sides_of(tg), nominal_of(tg)))
        groups = [list(range(i, min(i + args.resident_chunks, nchunks))) for i in range(0, nchunks, args.resident_chunks)]
    setup_s = time.perf_counter() - t_gen0
    if tasks is None:
        args.no_e2e = True

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def timed_region(step_fn, steps, warmup):
        """W warm-up steps, then exactly K steps between barrier + synchronize; (wall seconds, [(total_ms, staging_ms)] per bsw_run*)"""
        for _ in range(warmup):
            for b in batches:
                step_fn(b)
        ctx.sync()
        ctx.run_history2()                           # reset the per-run event history
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            for b in batches:
                step_fn(b)
        ctx.sync()
        barrier()
        return time.perf_counter() - t0, ctx.run_history2()

    # ---- the timed step: pack + bin + DP kernels of every resident batch (HIP events on the library's stream split it) ----
    dt, hist = timed_region(ctx.run_staged, args.steps, args.warmup)
    nb = len(batches)
    step_ms = [float(sum(h[0] for h in hist[i:i + nb])) for i in range(0, len(hist), nb)]
    stage_ms = [float(sum(h[1] for h in hist[i:i + nb])) for i in range(0, len(hist), nb)]
    dp_ms = [a - b for a, b in zip(step_ms, stage_ms)]
    # ---- the same batches, DP kernels alone (what rounds 1-5 timed) ----
    dt_k, kern_ms = None, []
    if args.kernels_only_steps > 0:
        dt_k, hk = timed_region(ctx.run, args.kernels_only_steps, min(args.warmup, 1))
        kern_ms = [float(sum(h[0] for h in hk[i:i + nb])) for i in range(0, len(hk), nb)]

    info = {"in_bytes": 0, "out_bytes": 0, "launches": 0}
    cells = ext_calls = nominal = 0
    res_parts = []
    for gi, b in enumerate(batches):
        r = ctx.download(b)
        bi = b.info()
        for key in info:
            info[key] += bi[key]
        b.free()
        cells += cells_of(r)
        if args.scaling == "strong":                    # exact cells of a strided sample of this rank's chunks (tests re-derive them with the oracle)
            lo = 0
            for c in groups[gi]:
                sz = min(chunk, args.pool - c * chunk)
                if c % args.sample_stride == 0:
                    sample_cells[str(c)] = cells_of(r[lo:lo + sz])
                lo += sz
        retry = int((r["left"]["aw"] > spec["w"]).sum() + (r["right"]["aw"] > spec["w"]).sum())
        ext_calls += gstats[gi][0] + retry
        nominal += gstats[gi][1]
        if tasks is not None:
            res_parts.append(r)
    res = np.concatenate(res_parts) if res_parts else None
    del res_parts

    # ---- PCIe-inclusive legs: the same seeds through bsw_submit* (host buffers in, host buffers out), on EVERY rank ----
    legs = {}                                             # name -> dict(seconds=[per rep], extra fields); times are rank-local
    stats_legs = {}

    def host_cost(c, n_seeds_total, before):
        st = c.host_stats()
        d = {k: st[k] - before.get(k, 0) for k in st}
        cpu_s = (d["slot_cpu_ns"] + d["helper_cpu_ns"]) / 1e9
        return {"cpu_s_per_M_seeds": round(cpu_s / max(d["seeds"], 1) * 1e6, 4), "slot_threads": st["slot_threads"],
                "h2d_bytes_per_seed": round(d["h2d_bytes"] / max(d["seeds"], 1), 1), "d2h_bytes_per_seed": round(d["d2h_bytes"] / max(d["seeds"], 1), 1),
                "seeds_accounted": d["seeds"]}

    def single_submits(c, call, reps):
        """`reps` single submits, each between barriers; rank-local seconds (submit -> wait returned)"""
        for _ in range(2):                                # warm up twice: staging allocations and code load, then the pipeline's steady layout
            call()
        before = c.host_stats()
        runs = []
        for _ in range(reps):
            barrier()
            t1 = time.perf_counter()
            call()
            runs.append(time.perf_counter() - t1)
        barrier()
        return runs, before

    def stream_two_deep(c, submit, bufs, reps):
        """A stream of batches through ONE context, two submits in flight (ABI 6 tickets): submit k+2 is issued as soon as k has
        been waited for — how an aligner that keeps producing seed batches uses the library.  Rank-local seconds per batch."""
        submit(bufs[0]); ta = c.last_ticket
        c.wait_ticket(ta)                                 # warm up
        before = c.host_stats()
        barrier()
        t1 = time.perf_counter()
        submit(bufs[0]); ta = c.last_ticket
        submit(bufs[1]); tb = c.last_ticket
        for _ in range(reps - 1):
            c.wait_ticket(ta); submit(bufs[0]); ta = c.last_ticket
            c.wait_ticket(tb); submit(bufs[1]); tb = c.last_ticket
        c.wait_ticket(ta); c.wait_ticket(tb)
        d = (time.perf_counter() - t1) / (2 * reps)
        barrier()
        return d, before

    STREAM_CHUNK = {"bytes": args.e2e_chunk, "packed": args.packed_chunk, "ref": args.e2e_chunk}
    for kind in list(STREAM_CHUNK):                       # (measurements) BENCH_STREAM_PACKED=chunk etc.
        ov = os.environ.get("BENCH_STREAM_" + kind.upper())
        if ov:
            STREAM_CHUNK[kind] = int(ov.split(",")[-1])
    if not args.no_e2e:
        hout = host.HostArena(max(n_local, 1) * host.RESULT.itemsize)
        hout2 = host.HostArena(max(n_local, 1) * host.RESULT.itemsize)
        out_buf = hout.view(host.RESULT, max(n_local, 1))[:n_local]
        out_buf2 = hout2.view(host.RESULT, max(n_local, 1))[:n_local]
        # -- byte per base in a registered arena: DMA'd as it is, packed and binned on the GPU
        with host.BswContext(device=local_rank, kernel=args.kernel, streams=args.e2e_slots, pack_threads=args.e2e_pack_threads, chunk_tasks=args.e2e_chunk) as c:
            runs, before = single_submits(c, lambda: c.extend_pairs(params, tasks, out=out_buf), args.e2e_reps)
            legs["e2e"] = {"runs": runs, "same": bool(out_buf.tobytes() == res.tobytes()), "host": host_cost(c, n_local, before)}
        with host.BswContext(device=local_rank, kernel=args.kernel, streams=args.e2e_slots, pack_threads=args.e2e_pack_threads, chunk_tasks=STREAM_CHUNK["bytes"]) as c:
            d2, before = stream_two_deep(c, lambda o: c.submit(params, tasks, o), (out_buf, out_buf2), args.stream_reps)
            legs["e2e_stream"] = {"runs": [d2], "same": bool(out_buf.tobytes() == res.tobytes() and out_buf2.tobytes() == res.tobytes()), "host": host_cost(c, n_local, before)}
        # -- the same seeds handed over 4-BIT PACKED (the device layout; bsw_submit_packed): no pack kernel, ~0.6x the PCIe bytes
        need = int(host.lib().bsw_pack_tasks_bound(tasks.ctypes.data, len(tasks)))
        parena = host.HostArena(need + 64)
        ptasks, _w = host.pack_tasks(tasks, parena.view(np.uint64, need // 8 + 1))
        with host.BswContext(device=local_rank, kernel=args.kernel, streams=args.e2e_slots, pack_threads=args.e2e_pack_threads, chunk_tasks=args.packed_chunk) as c:
            runs, before = single_submits(c, lambda: c.extend_pairs_packed(params, ptasks, out=out_buf), args.e2e_reps)
            legs["packed"] = {"runs": runs, "same": bool(out_buf.tobytes() == res.tobytes()), "host": host_cost(c, n_local, before), "bytes": need}
        with host.BswContext(device=local_rank, kernel=args.kernel, streams=args.e2e_slots, pack_threads=args.e2e_pack_threads, chunk_tasks=STREAM_CHUNK["packed"]) as c:
            d2, before = stream_two_deep(c, lambda o: c.submit_packed(params, ptasks, o), (out_buf, out_buf2), args.stream_reps)
            legs["packed_stream"] = {"runs": [d2], "same": bool(out_buf.tobytes() == res.tobytes() and out_buf2.tobytes() == res.tobytes()), "host": host_cost(c, n_local, before)}
        # -- the same submit handing back the RTL's 5-word record alone (BSW_RESULT_PAIR: 32 of the 96 result bytes per seed)
        pout = host.HostArena(max(n_local, 1) * host.PAIR.itemsize)
        pair_buf = pout.view(host.PAIR, max(n_local, 1))[:n_local]
        with host.BswContext(device=local_rank, kernel=args.kernel, streams=args.e2e_slots, pack_threads=args.e2e_pack_threads, chunk_tasks=args.packed_chunk, result_format=host.RESULT_PAIR) as c:
            runs, before = single_submits(c, lambda: c.extend_pairs_packed(params, ptasks, out=pair_buf), args.e2e_reps)
            legs["packed_pairs"] = {"runs": runs, "same": all(bool((pair_buf[f] == res[f]).all()) for f in host.PAIR.names), "host": host_cost(c, n_local, before)}
        pout.free()
        parena.free()
        # -- same shape of work, seeds against a DEVICE-RESIDENT reference: only the reads cross PCIe
        if args.ref_mbp > 0 and args.scaling == "weak":
            lp = args.ref_mbp * 1_000_000
            hreads = host.HostArena(spec["read_len"] * n_local + 4096)
            pac, rtasks, _ = host.synth_ref_tasks(n_local, lp, params, arena=hreads.u8, seed=3000 + rank, **spec)
            with host.BswContext(device=local_rank, kernel=args.kernel, streams=args.e2e_slots, pack_threads=args.e2e_pack_threads, chunk_tasks=args.e2e_chunk) as c:
                gref = c.ref_upload(pac, lp)

                def ref_call():
                    c.submit_ref(params, gref, rtasks, out=out_buf); c.wait()
                runs, before = single_submits(c, ref_call, args.e2e_reps)
                nchk = min(50_000, n_local)
                ref_first = out_buf.copy()
                legs["ref"] = {"runs": runs, "cells": cells_of(out_buf), "host": host_cost(c, n_local, before), "lp": lp,
                               "bytes": int(rtasks["l_query"].astype(np.int64).sum()),
                               "same": bool(c.extend_ref(params, gref, rtasks[:nchk]).tobytes() == out_buf[:nchk].tobytes())}
                c.ref_free(gref)
            with host.BswContext(device=local_rank, kernel=args.kernel, streams=args.e2e_slots, pack_threads=args.e2e_pack_threads, chunk_tasks=STREAM_CHUNK["ref"]) as c:
                gref = c.ref_upload(pac, lp)
                d2, before = stream_two_deep(c, lambda o: c.submit_ref(params, gref, rtasks, out=o), (out_buf, out_buf2), args.stream_reps)
                legs["ref_stream"] = {"runs": [d2], "host": host_cost(c, n_local, before),
                                      "same": bool(out_buf.tobytes() == ref_first.tobytes() and out_buf2.tobytes() == ref_first.tobytes())}
                c.ref_free(gref)
            del ref_first
            hreads.free()
        hout2.free()

    # ---- reductions: times MAX over ranks, volumes SUM; the per-rank rows travel whole ----
    LEG_NAMES = ["e2e", "e2e_stream", "packed", "packed_stream", "packed_pairs", "ref", "ref_stream"]
    place = ctx.placement()
    leg_med = [float(np.median(legs[k]["runs"])) if k in legs else 0.0 for k in LEG_NAMES]
    leg_cpu = [float(legs[k]["host"]["cpu_s_per_M_seeds"]) if k in legs else 0.0 for k in LEG_NAMES]
    row = [float(rank), float(props.pci_domain_id), float(props.pci_bus_id), float(props.pci_device_id), float(my_node),
           float(repinned), float(place["pinned_cpus"]), dt / args.steps * 1e3, float(len(cpu_affinity or [])),
           (dt_k or 0.0) / max(args.kernels_only_steps, 1) * 1e3, float(n_local)] + leg_med + leg_cpu
    rank_rows = [row]
    if dist is not None:
        mine_t = torch.tensor(row, dtype=torch.float64, device=red_dev)
        allr = [torch.zeros_like(mine_t) for _ in range(world)]
        dist.all_gather(allr, mine_t)
        rank_rows = [t.tolist() for t in allr]
        v = torch.tensor([float(cells), float(ext_calls), float(n_local), float(nominal), float(legs["ref"]["cells"]) if "ref" in legs else 0.0],
                         dtype=torch.float64, device=red_dev)
        tmax = torch.tensor([dt, dt_k or 0.0], dtype=torch.float64, device=red_dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(v, op=dist.ReduceOp.SUM)
        dt_all, dtk_all = float(tmax[0].item()), float(tmax[1].item())
        cells_all, ext_all, tasks_all, nominal_all, refcells_all = (float(x) for x in v.tolist())
    else:
        dt_all, dtk_all = dt, dt_k or 0.0
        cells_all, ext_all, tasks_all, nominal_all = float(cells), float(ext_calls), float(n_local), float(nominal)
        refcells_all = float(legs["ref"]["cells"]) if "ref" in legs else 0.0
    L0 = 11                                              # first leg column of a rank row

    out = None
    if rank == 0:
        gcups = cells_all * args.steps / dt_all / 1e9
        gcups_k = cells_all * args.kernels_only_steps / dtk_all / 1e9 if dtk_all else None
        dp_avg = float(np.mean(dp_ms)) if dp_ms else float("nan")      # DP kernels of one step of THIS rank (HIP events)
        st_avg = float(np.mean(stage_ms)) if stage_ms else 0.0
        alg_bytes = info["in_bytes"] + info["out_bytes"]          # per step: packed seq + task records + order + results
        pmc, pmc_src = pmc_summary(args.workload, n_local, key="%s@%d" % (args.workload, n_local))
        traffic = int((2 * pmc["FETCH_SIZE_KiB"] + pmc["WRITE_SIZE_KiB"]) * 1024) if "FETCH_SIZE_KiB" in pmc else None
        tops = cells * VALU_OPS_PER_CELL / (dp_avg * 1e-3) / 1e12
        ipc = pmc.get("valu_lane_insts_per_cell")
        waves = 1 if spec["read_len"] > 150 else 2
        out = {
            "metric": "GCUPS (seed-extension DP cells/s, %d bp %s, w=%d)" % (spec["read_len"], "PE seeds, left + right extension" if not spec["seed_at_start"] else "reads, right extension", spec["w"]),
            "value": round(gcups, 3), "unit": "GCUPS",
            "n_gpus": world, "world_size": (dist.get_world_size() if dist is not None else 1),
            "collective_backend": (("rccl" if args.backend == "nccl" else args.backend) if dist is not None else None),
            "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt_all / args.steps * 1e3, 4), "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "u8 scores in packed u16 VALU ops (v_pk_*_u16: two seeds per lane), range-checked per seed; wider scores run 16-bit rows / int32",
            "data": "synthetic",
            "step": "with_staging: pack (bytes -> 16 bases per uint64) + bin (counting sort into launch lists) + DP kernels of every resident batch (bsw_run_staged); inputs resident in HBM as byte-per-base sequences",
            "with_staging": {"gcups": round(gcups, 3), "ms_per_step": round(dt_all / args.steps * 1e3, 4),
                             "pack_bin_ms_per_step": round(st_avg, 4), "dp_kernels_ms_per_step": round(dp_avg, 4),
                             "pack_bin_share": round(st_avg / (st_avg + dp_avg), 4) if dp_ms else None, "timing": "HIP events on the library's stream (rank 0), wall clock for the step"},
            "kernels_only": ({"gcups": round(gcups_k, 3), "ms_per_step": round(dtk_all / args.kernels_only_steps * 1e3, 4), "steps": args.kernels_only_steps,
                              "kernel_ms_avg": round(float(np.mean(kern_ms)), 4), "step": "bsw_run: the DP kernels of the same resident batches alone (the `value` of rounds 1-5)"}
                             if gcups_k else None),
            "config": {"workload": args.workload, "seeds_per_gpu": n_local if args.scaling == "weak" else None,
                       "pool_seeds": args.pool if args.scaling == "strong" else None, "read_len": spec["read_len"],
                       "band_w": spec["w"], "zdrop": args.zdrop, "variant": "H" if args.variant == 0 else "M", "gaps": args.gaps or "6,1,6,1",
                       "baseline_config": ("BASELINE.json configs[2] shape: 150 bp PE seeds (left + right extension each), mixed (qlen, tlen) bins via the batch manager"
                                           if args.workload == "150bp_w100_mixed_bins" else None),
                       "sharding": "per-read task shard (chunk c -> rank c mod N), no collective" if world > 1 else "single GPU",
                       "kernel_launches_per_step": info["launches"], "resident_batches_per_rank": nb, "preset": args.preset,
                       "cpu_affinity": cpu_list_str(cpu_affinity), "rank_cpus_pinned": bool(os.environ.get("BSW_RANK_CPUS")) or repinned,
                       "ranks": [{"rank": int(r[0]), "gpu_bdf": "%04x:%02x:%02x.0" % (int(r[1]), int(r[2]), int(r[3])), "numa_node": int(r[4]),
                                  "repinned_after_hip_check": bool(r[5]), "library_slot_threads_pinned_cpus": int(r[6]),
                                  "ms_per_step": round(r[7], 4), "rank_cpus": int(r[8]), "ms_per_step_kernels_only": round(r[9], 4)} for r in rank_rows],
                       "ms_per_step_min_max_over_ranks": [round(min(r[7] for r in rank_rows), 4), round(max(r[7] for r in rank_rows), 4)],
                       "setup_s": round(setup_s, 1), "gen_threads": gen_threads,
                       "rank0_chunk_cells_sample": ({"every": args.sample_stride, "chunk_seeds": chunk, "generator_seed": "5000 + chunk index", "cells": sample_cells}
                                                    if args.scaling == "strong" else None)},
            "extensions_per_s": round(ext_all * args.steps / dt_all, 1),
            "seeds_per_s": round(tasks_all * args.steps / dt_all, 1),
            "cells_per_step": cells_all,
            "nominal_gcups_qlen_x_tlen": round(nominal_all * args.steps / dt_all / 1e9, 3),
            "roofline": {
                "bound": "valu", "kernel": "bsw_lane2_kernel / bsw_lane2l_kernel (the DP launches of a step)", "ops_per_cell": VALU_OPS_PER_CELL,
                "achieved": round(tops, 4), "peak": round(PEAK_VALU_TOPS, 2), "unit": "T lane-ops/s",
                "frac": round(tops / PEAK_VALU_TOPS, 5),
                "frac_is": "the algorithmic model: 15 integer lane-ops per counted cell / the DP kernels' time / peak",
                "frac_measured": (round(cells * ipc / (dp_avg * 1e-3) / 1e12 / PEAK_VALU_TOPS, 5) if ipc else None),
                "frac_measured_is": "VALU lane-instructions the kernels actually issued per counted cell (rocprofv3 SQ_INSTS_VALU x 64 / cells) in place of the 15-op model",
                "valu_insts_per_cell": ipc, "valu_issue_busy": pmc.get("valu_issue_busy"),
                "waves_per_simd_avg": pmc.get("waves_per_simd_avg"),
                "traffic": traffic, "traffic_kernels": pmc.get("traffic_kernels"),
                "counters_source": pmc_src,
                "issue_ceiling": issue_ceiling(pmc, cells, dp_avg, waves),
                "kernel_ms_avg": round(dp_avg, 4), "kernel_ms_is": "DP part of one step on rank 0: all DP launches of its %d resident batches (HIP events)" % nb,
                "note": "integer max/add DP at ~0.02 B/cell: VALU issue binds, not HBM and not MFMA; see roofline_hbm",
            },
            "roofline_staging": (None if tasks is None else (lambda raw, seqb, nn, orderb: {
                "bound": "hbm", "kernels": "bsw_pack_kernel + bsw_bin_count / scan / scatter (the pack + bin part of a step)",
                "achieved": round((raw + seqb + nn * (44 + 16 + 1) + nn * (44 + 1 + 8) + nn * 8 + orderb) / (st_avg * 1e-3) / 1e9, 1) if st_avg else None,
                "peak": PEAK_HBM_GBS, "unit": "GB/s",
                "frac": round((raw + seqb + nn * (44 + 16 + 1) + nn * (44 + 1 + 8) + nn * 8 + orderb) / (st_avg * 1e-3) / 1e9 / PEAK_HBM_GBS, 4) if st_avg else None,
                "algorithmic_bytes_per_step": raw + seqb + nn * (44 + 16 + 1) + nn * (44 + 1 + 8) + nn * 8 + orderb,
                "bytes_are": "pack: raw bases in, packed words out, task + offset records in, N flags out; bins: task records + flags in, keys out and in, launch lists out",
                "ms_per_step": round(st_avg, 4)})(harena_used(tasks), info["in_bytes"] - n_local * 44 - 0, n_local, 0)),
            "roofline_hbm": {
                "bound": "hbm", "achieved": round(alg_bytes / (dp_avg * 1e-3) / 1e9, 3), "peak": PEAK_HBM_GBS,
                "unit": "GB/s", "frac": round(alg_bytes / (dp_avg * 1e-3) / 1e9 / PEAK_HBM_GBS, 6),
                "algorithmic_bytes_per_step": alg_bytes, "traffic": traffic,
            },
        }
        # ---- PCIe-inclusive legs: job rate = all ranks' seeds / the slowest rank's time; per-rank rates beside it ----
        if legs:
            n_job = tasks_all
            cores_box = os.cpu_count()

            def leg_obj(name, path, cells_job, extra=None):
                k = LEG_NAMES.index(name)
                ts = [r[L0 + k] for r in rank_rows]
                cpu = [r[L0 + len(LEG_NAMES) + k] for r in rank_rows]
                t = max(ts)
                rate = n_job / t
                o = {"seeds_per_s": round(rate, 1), "gcups": round(cells_job / t / 1e9, 1),
                     "ratio_to_hbm_resident": round((cells_job / t / 1e9) / gcups, 3) if cells_job == cells_all else round(rate / (tasks_all * args.steps / dt_all), 3),
                     "ratio_to_kernels_only": (round((cells_job / t / 1e9) / gcups_k, 3) if gcups_k and cells_job == cells_all else None),
                     "path": path, "bit_exact": legs[name]["same"],
                     "per_rank_seeds_per_s": [round(r[10] / x, 1) if x else None for r, x in zip(rank_rows, ts)],
                     "host_cost": dict(legs[name]["host"], cpu_s_per_M_seeds_max_over_ranks=round(max(cpu), 4),
                                       cores_to_feed_one_gpu_at_this_rate=round(max(cpu) * (rate / world) / 1e6, 2),
                                       cores_to_feed_8_gpus_at_this_rate=round(max(cpu) * (rate / world) / 1e6 * 8, 1), cores_of_this_box=cores_box)}
                if len(legs[name]["runs"]) > 1:
                    o["reps_median_of"] = len(legs[name]["runs"])
                    o["spread"] = spread(legs[name]["runs"], n_local)
                else:
                    o["batches_timed"] = 2 * args.stream_reps
                    o["in_flight"] = "two submits in ONE context (tickets: bsw_submit_t / bsw_wait_ticket), %d slot threads + %d helpers each for the host pass" % (args.e2e_slots, args.e2e_pack_threads // max(args.e2e_slots, 1))
                if extra:
                    o.update(extra)
                return o
            bps = harena_used(tasks) / max(len(tasks), 1)
            out["e2e"] = leg_obj("e2e", "bsw_submit: registered host arena DMA'd as is, pack + bin on the GPU, results DMA'd into registered host memory",
                                 cells_all, {"bytes_per_seed_h2d": round(bps + 60, 1), "pcie_h2d_GBps": round((bps + 60) * n_job / max(r[L0] for r in rank_rows) / 1e9, 1)})
            out["e2e"]["stream_two_in_flight"] = leg_obj("e2e_stream", "the same, a stream of submits kept two deep", cells_all)
            out["e2e_packed_input"] = leg_obj("packed", "bsw_submit_packed: sequences 4-bit packed by the caller (16 bases per uint64, the device layout) in a registered arena, "
                                              "DMA'd straight into the sequence buffer, no pack kernel; packing itself is not timed (the caller keeps its reads packed)",
                                              cells_all, {"bytes_per_seed_h2d": round((legs["packed"]["bytes"] + len(tasks) * 44) / max(len(tasks), 1), 1)})
            out["e2e_packed_input"]["stream_two_in_flight"] = leg_obj("packed_stream", "the same, a stream of submits kept two deep", cells_all)
            out["e2e_packed_input"]["pair_records"] = leg_obj("packed_pairs", "bsw_config.result_format = BSW_RESULT_PAIR: the RTL's 5-word record alone comes back (32 of 96 bytes per seed)",
                                                              cells_all, {"bytes_per_seed_d2h": 32})
            if "ref" in legs:
                out["e2e_device_reference"] = leg_obj("ref", "bsw_submit_ref: %d Mbp synthetic genome resident in HBM (2 bits/base), reads DMA'd from registered host memory, "
                                                      "targets fetched and left flanks mirrored on the GPU" % (legs["ref"]["lp"] // 1_000_000), refcells_all,
                                                      {"bytes_per_seed_h2d": round((legs["ref"]["bytes"] + n_local * (44 + 16 + 16)) / max(n_local, 1), 1)})
                out["e2e_device_reference"]["stream_two_in_flight"] = leg_obj("ref_stream", "the same, a stream of submits kept two deep", refcells_all)
        if world == 1 and not args.no_cpu_baseline and tasks is not None:
            out.update(cpu_legs(host, params, tasks, res, args))
    if out is not None and world == 1 and not args.no_extra and not args.spec and args.scaling == "weak":
        # BASELINE.json configs[1] (single bin) and the 250 bp shape of configs[4]; reported beside the headline, never part of `value`
        extra = {}
        for wl in ("150bp_w100_single_bin", "250bp_w500", "150bp_w100_mixed_bins"):
            if wl == args.workload:
                continue
            sp2 = dict(WORKLOADS[wl])
            p2 = host.default_params(variant=args.variant, zdrop=args.zdrop, w=sp2["w"])
            a2 = host.HostArena(host.synth_arena_bound(args.other_seeds, **sp2) + 4096)
            t2, _ = host.synth_tasks(args.other_seeds, arena=a2.u8, seed=2000, **sp2)
            b2 = ctx.upload_raw(p2, t2)
            ctx.run(b2); ctx.run_staged(b2); ctx.sync(); ctx.run_history2()
            for _ in range(3):
                ctx.run(b2)
            for _ in range(3):
                ctx.run_staged(b2)
            h2 = ctx.run_history2()
            ms, ms_st, ms_pb = float(np.mean([h[0] for h in h2[:3]])), float(np.mean([h[0] for h in h2[3:]])), float(np.mean([h[1] for h in h2[3:]]))
            r2 = ctx.download(b2)
            c2 = cells_of(r2)
            g2 = c2 / (ms * 1e-3) / 1e9
            pm2, src2 = pmc_summary(wl, args.other_seeds)
            extra[wl] = {"gcups": round(c2 / (ms_st * 1e-3) / 1e9, 1), "ms_per_step": round(ms_st, 3), "pack_bin_ms": round(ms_pb, 3), "seeds": args.other_seeds,
                         "gcups_kernels_only": round(g2, 1), "ms_per_step_kernels_only": round(ms, 3),
                         "roofline_frac": round(c2 * VALU_OPS_PER_CELL / ((ms_st - ms_pb) * 1e-3) / 1e12 / PEAK_VALU_TOPS, 4), "kernel_launches_per_step": b2.info()["launches"],
                         "valu_insts_per_cell": pm2.get("valu_lane_insts_per_cell"), "valu_issue_busy": pm2.get("valu_issue_busy"),
                         "traffic": (int((2 * pm2["FETCH_SIZE_KiB"] + pm2["WRITE_SIZE_KiB"]) * 1024) if "FETCH_SIZE_KiB" in pm2 else None),
                         "counters": src2.get("status")}
            b2.free()
            a2.free()
        out["other_workloads"] = extra
        out["other_paths"] = side_paths(host, local_rank)
        if not args.no_e2e:
            # the same submit path when the caller's memory is NOT registered: host threads gather into pinned staging
            n3 = min(n_local, args.other_seeds)
            t3, a3 = host.synth_tasks(n3, seed=1000, **spec)
            with host.BswContext(device=local_rank, kernel=args.kernel, streams=4, pack_threads=4, chunk_tasks=chunk) as c3:
                o3 = np.ones(len(t3), dtype=host.RESULT)
                c3.extend_pairs(params, t3, out=o3)
                s0 = c3.host_stats()
                t1 = time.perf_counter()
                r3 = c3.extend_pairs(params, t3, out=o3)
                d3 = time.perf_counter() - t1
                s1 = c3.host_stats()
            out["e2e_unregistered_memory"] = {"seeds_per_s": round(len(t3) / d3, 1), "gcups": round(cells_of(r3) / d3 / 1e9, 1), "seeds": n3,
                                              "pack_threads": 4, "path": "bsw_submit: pageable host memory, 4 threads gather into pinned staging",
                                              "cpu_s_per_M_seeds": round((s1["slot_cpu_ns"] + s1["helper_cpu_ns"] - s0["slot_cpu_ns"] - s0["helper_cpu_ns"]) / 1e9 / n3 * 1e6, 4)}
    ctx.close()
    harena.free()
    if not args.no_e2e:
        hout.free()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if out is not None:
        print(json.dumps(out), flush=True)


def cpu_legs(host, params, tasks, res, args):
    """The CPU beside the GPU (rank 0, N = 1): the scalar oracle on a bounded sample, the inter-task AVX2 port on the whole batch
    at the 1-GPU box's CPU share (16 threads) and, when the process may use more CPUs than that, on ALL of them."""
    orc = graft.load_oracle()
    avail = len(os.sched_getaffinity(0))
    ncpu = args.cpu_threads or min(avail, 16)
    ns = min(args.cpu_sample, len(tasks))
    runs = []
    for _ in range(3):
        t1 = time.perf_counter()
        ref = orc.pair_batch(params, tasks[:ns], nthreads=ncpu)
        runs.append(time.perf_counter() - t1)
    dcpu = float(np.median(runs))
    ccells = cells_of(ref)
    n1 = min(ns, 30_000)
    t1 = time.perf_counter()
    orc.pair_batch(params, tasks[:n1], nthreads=1)
    d1 = time.perf_counter() - t1
    c1 = cells_of(ref[:n1])
    # the strong CPU baseline: the whole batch through the inter-task AVX2 kernel (16 seeds per __m256i), checked byte for byte
    # against the scalar oracle's result batch and against the GPU's
    nsv = len(tasks)

    def avx(nth):
        sr = []
        for _ in range(3):
            t1 = time.perf_counter()
            sref = orc.pair_batch_avx2(params, tasks[:nsv], nthreads=nth)
            sr.append(time.perf_counter() - t1)
        return sref, sr
    sref, sruns = avx(ncpu)
    dsimd = float(np.median(sruns))
    out = {"cpu_baseline": {
        "value": round(cells_of(sref) / dsimd / 1e9, 4), "unit": "GCUPS", "cores": ncpu, "kind": "port",
        "impl": "ours-avx2: inter-task SIMD ksw_extend2, 16 seeds per __m256i (int16 lanes), oracle/ksw_extend_avx2.c",
        "sample": "all %d seeds of the same batch, -O3 -march=x86-64-v3, %d pthreads, median of 3 runs (%s s)"
                  % (nsv, ncpu, "/".join("%.2f" % r for r in sruns)),
        "bit_exact_vs_scalar_oracle": bool(sref[:ns].tobytes() == ref.tobytes()),
        "bit_exact_vs_gpu": bool(sref.tobytes() == res.tobytes()),
        "cpus_this_process_may_use": avail, "cpus_of_the_box": os.cpu_count(),
        "scalar": {"value": round(ccells / dcpu / 1e9, 4), "cores": ncpu, "kind": "port",
                   "impl": "scalar C oracle (bwa's ksw_extend is scalar code too)", "sample_seeds": ns,
                   "runs_s": "/".join("%.2f" % r for r in runs), "single_thread_gcups": round(c1 / d1 / 1e9, 4)},
    }}
    quota = cgroup_cpu_quota()
    out["cpu_baseline"]["cgroup_cpu_quota_cpus"] = quota
    if quota is not None and quota <= ncpu + 0.5:
        # the box grants this process `quota` CPUs' worth of time (cgroup cpu.max) however many CPUs its affinity names: more threads
        # than that are throttled, not added (tools/diag/cpu_scaling.py, profiles/r6/cpu_baseline_scaling.txt: 41.9 GCUPS on 16
        # threads, 38 / 35 / 30 on 32 / 64 / 128 with nr_throttled counting up)
        out["cpu_baseline"]["all_cores"] = {"value": out["cpu_baseline"]["value"], "unit": "GCUPS", "cores": ncpu, "kind": "port",
                                            "note": "cgroup cpu.max grants %.4g CPUs of time (affinity names %d): the %d-thread figure IS this box's all-core figure; "
                                                    "threads beyond the quota are throttled" % (quota, avail, ncpu)}
    elif avail > ncpu and not args.cpu_threads:
        sall, aruns = avx(avail)
        out["cpu_baseline"]["all_cores"] = {"value": round(cells_of(sall) / float(np.median(aruns)) / 1e9, 4), "unit": "GCUPS", "cores": avail, "kind": "port",
                                            "impl": "the same AVX2 port on every CPU the process may use", "runs_s": "/".join("%.2f" % r for r in aruns),
                                            "bit_exact_vs_gpu": bool(sall.tobytes() == res.tobytes())}
    else:
        out["cpu_baseline"]["all_cores"] = {"value": out["cpu_baseline"]["value"], "cores": ncpu, "note": "the process may use %d CPUs: the %d-thread figure IS the all-core figure of this box's share" % (avail, ncpu)}
    nchk = min(args.check, ns)
    out["parity_spot_check"] = {"seeds": nchk, "bit_exact": bool(res[:nchk].tobytes() == ref[:nchk].tobytes()),
                                "cells_gpu_eq_cpu_on_sample": bool(cells_of(res[:ns]) == ccells), "sample_seeds": ns,
                                "whole_batch_vs_avx2_port": bool(sref.tobytes() == res.tobytes())}
    return out


def harena_used(tasks):
    """bytes of byte-per-base sequence the tasks reference (what one submit moves over PCIe besides the records)"""
    return int(tasks["lqlen"].astype(np.int64).sum() + tasks["ltlen"].astype(np.int64)[tasks["lqlen"] > 0].sum()
               + tasks["rqlen"].astype(np.int64).sum() + tasks["rtlen"].astype(np.int64)[tasks["rqlen"] > 0].sum())


if __name__ == "__main__":
    main()
