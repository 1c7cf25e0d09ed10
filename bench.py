#!/usr/bin/env python3
"""bench.py — GCUPS of the seed-extension hot path on N MI355X GPUs (one process per GPU).

A "step" = one pass of the hot path (left+right ksw_extend2 with band retry and the
mem_chain2aln decision) over one device-resident task batch.  At N=1 the workload is
BASELINE.json configs[1]: 1M synthetic 150 bp reads, w=100, single (qlen,tlen)=(131,257)
bin.  --scaling weak (default): every rank owns its own batch of that shape.  --scaling strong:
ONE pool of --pool seeds is cut into chunks, chunk c belongs to rank c mod N (the per-read task
shard of SURVEY.md §8e); no data-path collective either way.

Prints ONE JSON line on rank 0.  `value` = DP cells actually evaluated (exactly as the CPU
algorithm iterates them) / wall time / 1e9, summed over ranks, inputs already in HBM.
`e2e` = the same batch pushed through bsw_submit (host buffers in, host buffers out: DMA out of
registered host memory, packing and binning on the GPU) — PCIe-inclusive, never `value`.
`pe_mixed_bins` (N=1) = the workload BASELINE.json's metric names — 150 bp PE seeds, left + right
extension each, mixed (qlen, tlen) bins through the batch manager — at configs[2] size (10 M seeds,
--pe-seeds), resident in HBM, with its own roofline object and the counters of that workload's own
rocprofv3 passes (profiles/pmc_latest.json holds one entry per workload).  Never part of `value`.
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--tasks", type=int, default=1_000_000, help="seeds per GPU per step (weak scaling)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--pool", type=int, default=8_000_000, help="--scaling strong: seeds in the one pool all ranks share (BASELINE configs[3]: 100000000)")
    ap.add_argument("--gen-threads", type=int, default=0, help="--scaling strong: threads that generate this rank's chunks (0 = the rank's CPU set, at most 32)")
    ap.add_argument("--sample-stride", type=int, default=97, help="--scaling strong: the line carries the exact cell count of every k-th chunk of rank 0")
    ap.add_argument("--resident-chunks", type=int, default=32, help="--scaling strong: 128Ki-seed chunks per resident batch")
    ap.add_argument("--workload", default="150bp_w100_single_bin", choices=sorted(WORKLOADS))
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--zdrop", type=int, default=100)
    ap.add_argument("--gaps", default=None, help="o_del,e_del,o_ins,e_ins (default: bwa's 6,1,6,1)")
    ap.add_argument("--cpu-sample", type=int, default=250_000, help="seeds timed on the CPU oracle (rank 0, N=1): 3 runs of ~9 s on 16 threads")
    ap.add_argument("--cpu-threads", type=int, default=0, help="0 = min(affinity, 16): the 1-GPU box's CPU share")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the 250 bp and mixed-bin side measurements (N=1 only, outside the timed region)")
    ap.add_argument("--pe-seeds", type=int, default=10_000_000, help="seeds of the 150 bp PE mixed-bin measurement beside the headline (BASELINE configs[2]: 10 M; 0 = skip)")
    ap.add_argument("--packed-chunk", type=int, default=114688, help="seeds per chunk of the packed-input single-submit legs (112 Ki: 9 chunks per 1 M seeds; re-swept with round 5 kernels, profiles/r5/e2e_packed_chunk_sweep.txt: 64 / 80 / 96 / 112 / 128 / 160 Ki -> 103 / 113 / 115 / 123 / 122 / 95 M seeds/s)")
    ap.add_argument("--e2e-chunk", type=int, default=131072, help="seeds per chunk of the byte-input and device-reference single-submit legs")
    ap.add_argument("--e2e-slots", type=int, default=4, help="slot threads (= streams) of the single-submit PCIe-inclusive legs")
    ap.add_argument("--no-e2e", action="store_true", help="skip the bsw_submit (PCIe-inclusive) measurement")
    ap.add_argument("--e2e-reps", type=int, default=5, help="bsw_submit passes timed (median reported)")
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
                              "scaling": args.scaling, "workload": args.workload, "pool": args.pool}), flush=True)
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

    # ---- this rank's seeds, generated straight into pinned (DMA-able) host memory ----
    chunk = 131072
    ctx = host.BswContext(device=local_rank, kernel=args.kernel)
    if args.scaling == "strong":
        # one pool, chunk c -> rank c mod N; a rank's chunks become resident batches of <= --resident-chunks chunks (a resident
        # batch holds < 4 GiB of bases), generated, uploaded and dropped on the host one group at a time
        nchunks = (args.pool + chunk - 1) // chunk
        mine = [c for c in range(nchunks) if c % world == rank]
        groups = [mine[i:i + args.resident_chunks] for i in range(0, len(mine), args.resident_chunks)]
        gb = args.resident_chunks * host.synth_arena_bound(chunk, **spec) + 4096
        # Every chunk has its own generator seed, so the chunks of a group are generated side by side on this rank's CPUs
        # (bsw_synth_generate is 4.25 us per mixed-bin seed on one core: 100 M seeds would be 7 minutes of it), and group g+1
        # is generated into a second arena while group g is uploaded (validate + lay out on the host, pack + bin on the GPU).
        from concurrent.futures import ThreadPoolExecutor
        gen_threads = max(1, min(args.gen_threads or len(cpu_affinity or [0]), 32))
        arenas = [host.HostArena(gb) for _ in range(2 if len(groups) > 1 else 1)]
        harena = arenas[0]
        pool = ThreadPoolExecutor(max_workers=gen_threads)

        def generate(gi):
            grp = groups[gi]
            ar = arenas[gi % len(arenas)]
            sizes = [min(chunk, args.pool - c * chunk) for c in grp]
            tg = np.zeros(int(sum(sizes)), dtype=host.TASK)
            offs = np.concatenate([[0], np.cumsum([host.synth_arena_bound(sz, **spec) for sz in sizes])]).astype(np.int64)
            starts = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)

            def one(k):
                c, sz = grp[k], sizes[k]
                t, _ = host.synth_tasks(sz, arena=ar.u8[int(offs[k]):], seed=5000 + c, **spec)      # (ctypes releases the GIL)
                t["tag"] = np.arange(c * chunk, c * chunk + sz, dtype=np.uint32)
                tg[int(starts[k]):int(starts[k]) + sz] = t
            return tg, [pool.submit(one, k) for k in range(len(grp))]

        batches, n_local, gstats = [], 0, []
        t_gen0 = time.perf_counter()
        nxt = generate(0) if groups else None
        for gi in range(len(groups)):
            tg, futs = nxt
            for f in futs:
                f.result()
            nxt = generate(gi + 1) if gi + 1 < len(groups) else None     # ... overlaps this group's upload
            batches.append(ctx.upload(params, tg))                          # inputs resident in HBM before the timed region
            n_local += len(tg)
            gstats.append((int((tg["lqlen"] > 0).sum() + (tg["rqlen"] > 0).sum()),
                           int((tg["lqlen"].astype(np.int64) * tg["ltlen"]).sum() + (tg["rqlen"].astype(np.int64) * tg["rtlen"]).sum())))
            if rank == 0 and len(groups) > 1:
                print("bench.py: rank 0 resident batch %d/%d (%d seeds, %.1f s)" % (len(batches), len(groups), len(tg), time.perf_counter() - t_gen0),
                      file=sys.stderr, flush=True)
        pool.shutdown()
        setup_s = time.perf_counter() - t_gen0
        for ar in arenas[1:]:
            ar.free()
        tasks = tg if len(groups) == 1 else None                            # e2e legs only when the rank's share is one batch
        hout = host.HostArena(max(n_local if tasks is not None else 1, 1) * host.RESULT.itemsize)
    else:
        setup_s = None
        n_local = args.tasks
        harena = host.HostArena(host.synth_arena_bound(max(n_local, 1), **spec) + 4096)
        hout = host.HostArena(max(n_local, 1) * host.RESULT.itemsize)
        tasks, _ = host.synth_tasks(n_local, arena=harena.u8, seed=1000 + rank, **spec)
        batches = [ctx.upload(params, tasks)]        # inputs resident in HBM before the timed region
    if tasks is None:
        args.no_e2e = True
    out_buf = hout.view(host.RESULT, max(n_local, 1))[:n_local] if tasks is not None else None
    hout2 = host.HostArena(max(n_local, 1) * host.RESULT.itemsize) if (world == 1 and not args.no_e2e) else None
    out_buf2 = hout2.view(host.RESULT, max(n_local, 1))[:n_local] if hout2 is not None else None

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        for b in batches:
            ctx.run(b)
    ctx.sync()
    ctx.run_history()                            # reset per-run event history
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        for b in batches:
            ctx.run(b)
    ctx.sync()
    barrier()
    dt = time.perf_counter() - t0
    kern_ms = ctx.run_history()                  # HIP events on the library's own stream, one pair per bsw_run

    info = {"in_bytes": 0, "out_bytes": 0, "launches": 0}
    sample_cells = {}
    cells = ext_calls = nominal = 0
    res = None
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
        if tasks is not None:
            res = r
            ext_calls += int((tasks["lqlen"] > 0).sum() + (tasks["rqlen"] > 0).sum()) + retry
            nominal += int((tasks["lqlen"].astype(np.int64) * tasks["ltlen"]).sum() + (tasks["rqlen"].astype(np.int64) * tasks["rtlen"]).sum())
        else:
            ext_calls += gstats[gi][0] + retry
            nominal += gstats[gi][1]
    n_tasks_local = n_local
    if len(batches) > 1:                         # kernel time per step = the step's bsw_run calls together
        kern_ms = [float(sum(kern_ms[i:i + len(batches)])) for i in range(0, len(kern_ms), len(batches))]

    # streams kept two deep: two contexts; slots per context and chunk size per input format from the sweeps in
    # profiles/r3/e2e_hw_queues.txt (the HIP runtime maps streams onto GPU_MAX_HW_QUEUES = 4 hardware queues by default)
    # (re-swept with round 5's kernels and the late result DMAs, profiles/r5/e2e_stream_cfg_sweep.txt: two contexts of TWO slots each —
    # four streams, one per hardware queue — beat two of four: packed 146 vs 124, device reference 142 vs 123 M seeds/s)
    STREAM_CFG = {"bytes": (2, 2 * chunk), "packed": (2, 131072), "ref": (2, chunk)}
    for kind in list(STREAM_CFG):                     # (measurements) BENCH_STREAM_PACKED="slots,chunk" etc.
        ov = os.environ.get("BENCH_STREAM_" + kind.upper())
        if ov:
            STREAM_CFG[kind] = tuple(int(x) for x in ov.split(","))
    def stream_threads(kind):
        return "%d slot threads (2 contexts x %d slots, %d-seed chunks)" % (2 * STREAM_CFG[kind][0], STREAM_CFG[kind][0], STREAM_CFG[kind][1])

    def stream_two_in_flight(make_ctx, submit, reps=8):
        """A stream of batches, two in flight: two contexts, submit k+2 issued as soon as k is waited for — how an aligner
        that keeps producing seed batches uses the library.  Seconds per batch."""
        ca, cb = make_ctx(), make_ctx()
        sa, sb = submit(ca, out_buf), submit(cb, out_buf2)
        sa(); ca.wait(); sb(); cb.wait()                          # warm up both
        t1 = time.perf_counter()
        sa(); sb()
        for _ in range(reps - 1):
            ca.wait(); sa()
            cb.wait(); sb()
        ca.wait(); cb.wait()
        d = (time.perf_counter() - t1) / (2 * reps)
        same = bool(out_buf.tobytes() == out_buf2.tobytes())
        return d, same, (ca, cb)

    # ---- the same seeds through bsw_submit: host buffers in (registered arena), host buffers out ----
    e2e_dt = None
    if not args.no_e2e:
        sctx = host.BswContext(device=local_rank, kernel=args.kernel, streams=args.e2e_slots, pack_threads=4, chunk_tasks=args.e2e_chunk)
        for _ in range(2):                                       # warm up twice: staging allocations and code load, then the pipeline's steady layout (the first timed pass after ONE warm-up ran 16 ms against 10.4)
            sctx.extend_pairs(params, tasks, out=out_buf)
        e2e_runs = []
        for _ in range(args.e2e_reps):
            barrier()
            t1 = time.perf_counter()
            got = sctx.extend_pairs(params, tasks, out=out_buf)
            barrier()
            e2e_runs.append(time.perf_counter() - t1)
        e2e_dt = float(np.median(e2e_runs))
        e2e_same = bool(got.tobytes() == res.tobytes())
        sctx.close()
        e2e_stream = None
        if world == 1:
            d2, same2, (ca, cb) = stream_two_in_flight(
                lambda: host.BswContext(device=local_rank, kernel=args.kernel, streams=STREAM_CFG["bytes"][0], pack_threads=2, chunk_tasks=STREAM_CFG["bytes"][1]),
                lambda c, o: (lambda: c.submit(params, tasks, o)))
            e2e_stream = (d2, same2 and bool(out_buf.tobytes() == res.tobytes()))
            ca.close(); cb.close()

    # ---- the same seeds handed over 4-BIT PACKED (the device layout; bsw_submit_packed): no pack kernel, ~0.6x the PCIe bytes ----
    packed_leg = None
    if not args.no_e2e and world == 1:
        need = int(host.lib().bsw_pack_tasks_bound(tasks.ctypes.data, len(tasks)))
        parena = host.HostArena(need + 64)
        ptasks, _w = host.pack_tasks(tasks, parena.view(np.uint64, need // 8 + 1))
        # 96 Ki chunks: with half the bytes per seed the input DMAs are short and smaller chunks start the GPU sooner
        # (sweep: profiles/r3/e2e_packed_sweep.txt)
        pctx = host.BswContext(device=local_rank, kernel=args.kernel, streams=args.e2e_slots, pack_threads=4, chunk_tasks=args.packed_chunk)
        for _ in range(2):
            pctx.extend_pairs_packed(params, ptasks, out=out_buf)
        runs = []
        for _ in range(args.e2e_reps):
            barrier()
            t1 = time.perf_counter()
            gotp = pctx.extend_pairs_packed(params, ptasks, out=out_buf)
            barrier()
            runs.append(time.perf_counter() - t1)
        psame = bool(gotp.tobytes() == res.tobytes())
        pctx.close()
        d2, same2, (ca, cb) = stream_two_in_flight(
            lambda: host.BswContext(device=local_rank, kernel=args.kernel, streams=STREAM_CFG["packed"][0], pack_threads=2, chunk_tasks=STREAM_CFG["packed"][1]),
            lambda c, o: (lambda: c.submit_packed(params, ptasks, o)))
        pstream = (d2, same2 and bool(out_buf.tobytes() == res.tobytes()))
        ca.close(); cb.close()
        # the same submit handing back the RTL's 5-word record alone (BSW_RESULT_PAIR: 32 of the 96 result bytes per seed)
        pout = host.HostArena(max(n_local, 1) * host.PAIR.itemsize)
        pair_buf = pout.view(host.PAIR, max(n_local, 1))[:n_local]
        qctx = host.BswContext(device=local_rank, kernel=args.kernel, streams=args.e2e_slots, pack_threads=4, chunk_tasks=args.packed_chunk, result_format=host.RESULT_PAIR)
        for _ in range(2):
            qctx.extend_pairs_packed(params, ptasks, out=pair_buf)
        qruns = []
        for _ in range(args.e2e_reps):
            barrier()
            t1 = time.perf_counter()
            gotq = qctx.extend_pairs_packed(params, ptasks, out=pair_buf)
            barrier()
            qruns.append(time.perf_counter() - t1)
        qsame = all(bool((gotq[f] == res[f]).all()) for f in host.PAIR.names)
        qctx.close()
        pout.free()
        packed_leg = (float(np.median(runs)), psame, need, pstream, float(np.median(qruns)), qsame, list(runs), list(qruns))
        parena.free()

    # ---- same shape of work, seeds against a DEVICE-RESIDENT reference: only the reads cross PCIe ----
    ref_leg = None
    if not args.no_e2e and args.ref_mbp > 0 and args.scaling == "weak":
        lp = args.ref_mbp * 1_000_000
        hreads = host.HostArena(spec["read_len"] * n_local + 4096)
        pac, rtasks, _ = host.synth_ref_tasks(n_local, lp, params, arena=hreads.u8, seed=3000 + rank, **spec)
        rctx = host.BswContext(device=local_rank, kernel=args.kernel, streams=args.e2e_slots, pack_threads=4, chunk_tasks=args.e2e_chunk)
        gref = rctx.ref_upload(pac, lp)
        for _ in range(2):                                                      # warm up
            rctx.submit_ref(params, gref, rtasks, out=out_buf); rctx.wait()
        runs = []
        for _ in range(args.e2e_reps):
            barrier()
            t1 = time.perf_counter()
            rctx.submit_ref(params, gref, rtasks, out=out_buf); rctx.wait()
            barrier()
            runs.append(time.perf_counter() - t1)
        rcells = cells_of(out_buf)
        nchk = min(50_000, n_local)
        same = bool(rctx.extend_ref(params, gref, rtasks[:nchk]).tobytes() == out_buf[:nchk].tobytes())
        ref_first = out_buf.copy() if world == 1 else None
        ref_leg = (float(np.median(runs)), rcells, same, int(rtasks["l_query"].astype(np.int64).sum()), lp, list(runs))
        rctx.ref_free(gref)
        rctx.close()
        ref_stream = None
        if world == 1:
            refs = {}
            def mk():
                c = host.BswContext(device=local_rank, kernel=args.kernel, streams=STREAM_CFG["ref"][0], pack_threads=2, chunk_tasks=STREAM_CFG["ref"][1])
                refs[id(c)] = c.ref_upload(pac, lp)
                return c
            d2, same2, (ca, cb) = stream_two_in_flight(mk, lambda c, o: (lambda: c.submit_ref(params, refs[id(c)], rtasks, out=o)))
            ref_stream = (d2, same2 and bool(out_buf.tobytes() == ref_first.tobytes()))
            for c in (ca, cb):
                c.ref_free(refs[id(c)]); c.close()
        hreads.free()

    place = ctx.placement()
    rank_rows = [[float(rank), float(props.pci_domain_id), float(props.pci_bus_id), float(props.pci_device_id), float(my_node),
                  float(repinned), float(place["pinned_cpus"]), dt / args.steps * 1e3, float(len(cpu_affinity or []))]]
    if dist is not None:
        mine_t = torch.tensor(rank_rows[0], dtype=torch.float64, device=red_dev)
        allr = [torch.zeros_like(mine_t) for _ in range(world)]
        dist.all_gather(allr, mine_t)
        rank_rows = [t.tolist() for t in allr]
    if dist is not None:
        v = torch.tensor([dt, float(cells), float(ext_calls), float(n_tasks_local), float(nominal), e2e_dt or 0.0],
                         dtype=torch.float64, device=red_dev)
        tmax = v[[0, 5]].clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(v, op=dist.ReduceOp.SUM)
        dt_all, e2e_all = float(tmax[0].item()), float(tmax[1].item())
        cells_all, ext_all, tasks_all, nominal_all = (float(x) for x in v[1:5].tolist())
    else:
        dt_all, e2e_all = dt, e2e_dt or 0.0
        cells_all, ext_all, tasks_all, nominal_all = float(cells), float(ext_calls), float(n_tasks_local), float(nominal)

    out = None
    if rank == 0:
        gcups = cells_all * args.steps / dt_all / 1e9
        kavg_ms = float(np.mean(kern_ms)) if kern_ms else float("nan")
        alg_bytes = info["in_bytes"] + info["out_bytes"]          # per launch: packed seq + task records + order + results
        pmc, pmc_src = pmc_summary(args.workload, n_local)
        traffic = int((2 * pmc["FETCH_SIZE_KiB"] + pmc["WRITE_SIZE_KiB"]) * 1024) if "FETCH_SIZE_KiB" in pmc else None
        tops = cells * VALU_OPS_PER_CELL / (kavg_ms * 1e-3) / 1e12
        out = {
            "metric": "GCUPS (seed-extension DP cells/s, %d bp reads, w=%d)" % (spec["read_len"], spec["w"]), "value": round(gcups, 3), "unit": "GCUPS",
            "n_gpus": world, "world_size": (dist.get_world_size() if dist is not None else 1),
            "collective_backend": (("rccl" if args.backend == "nccl" else args.backend) if dist is not None else None),
            "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt_all / args.steps * 1e3, 4), "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "u8 scores in packed u16 VALU ops (v_pk_*_u16: two seeds per lane), range-checked per seed; wider scores run 16-bit rows / int32",
            "data": "synthetic",
            "config": {"workload": args.workload, "seeds_per_gpu": n_local if args.scaling == "weak" else None,
                       "pool_seeds": args.pool if args.scaling == "strong" else None, "read_len": spec["read_len"],
                       "band_w": spec["w"], "zdrop": args.zdrop, "variant": "H" if args.variant == 0 else "M", "gaps": args.gaps or "6,1,6,1",
                       "sharding": "per-read task shard (chunk c -> rank c mod N), no collective" if world > 1 else "single GPU",
                       "kernel_launches_per_step": info["launches"], "resident_batches_per_rank": len(batches), "preset": args.preset,
                       "cpu_affinity": cpu_list_str(cpu_affinity), "rank_cpus_pinned": bool(os.environ.get("BSW_RANK_CPUS")) or repinned,
                       "ranks": [{"rank": int(r[0]), "gpu_bdf": "%04x:%02x:%02x.0" % (int(r[1]), int(r[2]), int(r[3])), "numa_node": int(r[4]),
                                  "repinned_after_hip_check": bool(r[5]), "library_slot_threads_pinned_cpus": int(r[6]),
                                  "ms_per_step": round(r[7], 4), "rank_cpus": int(r[8])} for r in rank_rows],
                       "ms_per_step_min_max_over_ranks": [round(min(r[7] for r in rank_rows), 4), round(max(r[7] for r in rank_rows), 4)],
                       "setup_s": round(setup_s, 1) if args.scaling == "strong" else None,
                       "rank0_chunk_cells_sample": ({"every": args.sample_stride, "chunk_seeds": chunk, "generator_seed": "5000 + chunk index", "cells": sample_cells}
                                                    if args.scaling == "strong" else None)},
            "extensions_per_s": round(ext_all * args.steps / dt_all, 1),
            "seeds_per_s": round(tasks_all * args.steps / dt_all, 1),
            "cells_per_step": cells_all,
            "nominal_gcups_qlen_x_tlen": round(nominal_all * args.steps / dt_all / 1e9, 3),
            "roofline": {
                "bound": "valu", "ops_per_cell": VALU_OPS_PER_CELL,
                "achieved": round(tops, 4), "peak": round(PEAK_VALU_TOPS, 2), "unit": "T lane-ops/s",
                "frac": round(tops / PEAK_VALU_TOPS, 5),
                "valu_insts_per_cell": pmc.get("valu_lane_insts_per_cell"), "valu_issue_busy": pmc.get("valu_issue_busy"),
                "traffic": traffic, "traffic_kernels": pmc.get("traffic_kernels"),
                "counters_source": pmc_src,
                "issue_ceiling": issue_ceiling(pmc, cells, kavg_ms, 1 if spec["read_len"] > 150 else 2),
                "kernel_ms_avg": round(kavg_ms, 4),
                "note": "integer max/add DP at ~0.02 B/cell: VALU issue binds, not HBM and not MFMA; see roofline_hbm",
            },
            "roofline_hbm": {
                "bound": "hbm", "achieved": round(alg_bytes / (kavg_ms * 1e-3) / 1e9, 3), "peak": PEAK_HBM_GBS,
                "unit": "GB/s", "frac": round(alg_bytes / (kavg_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 6),
                "algorithmic_bytes_per_launch": alg_bytes, "traffic": traffic,
            },
        }
        if e2e_dt is not None:
            out["e2e"] = {
                "seeds_per_s": round(tasks_all / e2e_all, 1), "gcups": round(cells_all / e2e_all / 1e9, 1),
                "ratio_to_hbm_resident": round((cells_all / e2e_all / 1e9) / gcups, 3),
                "pack_threads": 4, "host_threads": "4 slot threads (validate + count), no host packing", "reps_median_of": args.e2e_reps,
                "pcie_h2d_GBps": round((harena_used(tasks) + len(tasks) * 60) * world / e2e_all / 1e9, 1),
                "path": "bsw_submit: registered host arena DMA'd as is, pack + bin on the GPU, results DMA'd into registered host memory",
                "bytes_per_seed_h2d": round((harena_used(tasks) + len(tasks) * 60) / max(len(tasks), 1), 1),
                "bit_exact_vs_resident_run": e2e_same,
                "spread": spread(e2e_runs, len(tasks)),
            }
            if e2e_stream is not None:
                out["e2e"]["stream_two_in_flight"] = {
                    "seeds_per_s": round(len(tasks) / e2e_stream[0], 1), "gcups": round(cells / e2e_stream[0] / 1e9, 1),
                    "ratio_to_hbm_resident": round((cells / e2e_stream[0] / 1e9) / gcups, 3), "host_threads": stream_threads("bytes"),
                    "batches_timed": 16, "bit_exact_vs_resident_run": e2e_stream[1]}
        if packed_leg is not None:
            pdt, psame, pbytes, pstream, qdt, qsame, pruns, qruns_ = packed_leg
            out["e2e_packed_input"] = {
                "seeds_per_s": round(len(tasks) / pdt, 1), "gcups": round(cells / pdt / 1e9, 1),
                "ratio_to_hbm_resident": round((cells / pdt / 1e9) / gcups, 3), "host_threads": "4 slot threads", "reps_median_of": args.e2e_reps,
                "path": "bsw_submit_packed: sequences 4-bit packed by the caller (16 bases per uint64, the device layout) in a registered arena, "
                        "DMA'd straight into the sequence buffer, no pack kernel; packing itself is not timed (the caller keeps its reads packed)",
                "bytes_per_seed_h2d": round((pbytes + len(tasks) * 44) / max(len(tasks), 1), 1),
                "bit_exact_vs_resident_run": psame, "spread": spread(pruns, len(tasks)),
                "stream_two_in_flight": {"seeds_per_s": round(len(tasks) / pstream[0], 1), "gcups": round(cells / pstream[0] / 1e9, 1),
                                         "ratio_to_hbm_resident": round((cells / pstream[0] / 1e9) / gcups, 3),
                                         "host_threads": stream_threads("packed"), "batches_timed": 16,
                                         "bit_exact_vs_resident_run": pstream[1]},
                "pair_records": {"seeds_per_s": round(len(tasks) / qdt, 1), "gcups": round(cells / qdt / 1e9, 1),
                                 "ratio_to_hbm_resident": round((cells / qdt / 1e9) / gcups, 3), "bytes_per_seed_d2h": 32,
                                 "path": "bsw_config.result_format = BSW_RESULT_PAIR: the RTL's 5-word record alone comes back (32 of 96 bytes per seed)",
                                 "eight_fields_equal_full_records": qsame, "spread": spread(qruns_, len(tasks))}}
        if ref_leg is not None and world == 1:
            rdt, rcells, rsame, rbytes, rlp, rruns = ref_leg
            out["e2e_device_reference"] = {
                "seeds_per_s": round(n_local / rdt, 1), "gcups": round(rcells / rdt / 1e9, 1),
                "ratio_to_hbm_resident": round((n_local / rdt) / (tasks_all * args.steps / dt_all), 3),
                "host_threads": "4 slot threads", "reps_median_of": args.e2e_reps,
                "path": "bsw_submit_ref: %d Mbp synthetic genome resident in HBM (2 bits/base), reads DMA'd from registered host memory, "
                        "targets fetched and left flanks mirrored on the GPU" % (rlp // 1_000_000),
                "bytes_per_seed_h2d": round((rbytes + n_local * (44 + 16 + 16)) / max(n_local, 1), 1),
                "bit_exact_vs_resident_fetch_path": rsame, "spread": spread(rruns, n_local),
            }
            if ref_stream is not None:
                out["e2e_device_reference"]["stream_two_in_flight"] = {
                    "seeds_per_s": round(n_local / ref_stream[0], 1), "gcups": round(rcells / ref_stream[0] / 1e9, 1),
                    "ratio_to_hbm_resident": round((n_local / ref_stream[0]) / (tasks_all * args.steps / dt_all), 3),
                    "host_threads": stream_threads("ref"), "batches_timed": 16, "bit_exact_vs_single_submit": ref_stream[1]}
        if world == 1 and not args.no_cpu_baseline and tasks is not None:
            orc = graft.load_oracle()
            ncpu = args.cpu_threads or min(len(os.sched_getaffinity(0)), 16)
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
            # the strong CPU baseline: the same sample through the inter-task AVX2 kernel (16 seeds per __m256i), checked
            # byte for byte against the scalar oracle's result batch
            nsv = len(tasks)                                   # the whole batch: the vector kernel needs < 0.5 s for it
            sruns = []
            for _ in range(3):
                t1 = time.perf_counter()
                sref = orc.pair_batch_avx2(params, tasks[:nsv], nthreads=ncpu)
                sruns.append(time.perf_counter() - t1)
            dsimd = float(np.median(sruns))
            out["cpu_baseline"] = {
                "value": round(cells_of(sref) / dsimd / 1e9, 4), "unit": "GCUPS", "cores": ncpu, "kind": "port",
                "impl": "ours-avx2: inter-task SIMD ksw_extend2, 16 seeds per __m256i (int16 lanes), oracle/ksw_extend_avx2.c",
                "sample": "all %d seeds of the same batch, -O3 -march=x86-64-v3, %d pthreads, median of 3 runs (%s s)"
                          % (nsv, ncpu, "/".join("%.2f" % r for r in sruns)),
                "bit_exact_vs_scalar_oracle": bool(sref[:ns].tobytes() == ref.tobytes()),
                "bit_exact_vs_gpu": bool(sref.tobytes() == res.tobytes()),
                "scalar": {"value": round(ccells / dcpu / 1e9, 4), "cores": ncpu, "kind": "port",
                           "impl": "scalar C oracle (bwa's ksw_extend is scalar code too)", "sample_seeds": ns,
                           "runs_s": "/".join("%.2f" % r for r in runs), "single_thread_gcups": round(c1 / d1 / 1e9, 4)},
            }
            nchk = min(args.check, ns)
            out["parity_spot_check"] = {"seeds": nchk, "bit_exact": bool(res[:nchk].tobytes() == ref[:nchk].tobytes()),
                                        "cells_gpu_eq_cpu_on_sample": bool(cells_of(res[:ns]) == ccells), "sample_seeds": ns}
    if out is not None and world == 1 and not args.no_extra and not args.spec and args.scaling == "weak":
        # BASELINE.json also asks for 250 bp batches; reported beside the headline, never part of `value`
        extra = {}
        for wl in ("250bp_w500", "150bp_w100_mixed_bins"):
            if wl == args.workload:
                continue
            sp2 = dict(WORKLOADS[wl])
            p2 = host.default_params(variant=args.variant, zdrop=args.zdrop, w=sp2["w"])
            t2, a2 = host.synth_tasks(args.tasks, seed=2000, **sp2)
            b2 = ctx.upload(p2, t2)
            ctx.run(b2); ctx.sync(); ctx.run_history()
            for _ in range(3):
                ctx.run(b2)
            ctx.sync()
            ms = float(np.mean(ctx.run_history()))
            r2 = ctx.download(b2)
            g2 = cells_of(r2) / (ms * 1e-3) / 1e9
            pm2, src2 = pmc_summary(wl, args.tasks)
            extra[wl] = {"gcups": round(g2, 1), "ms_per_step": round(ms, 3), "seeds": args.tasks,
                         "roofline_frac": round(g2 * 1e9 * VALU_OPS_PER_CELL / 1e12 / PEAK_VALU_TOPS, 4), "kernel_launches_per_step": b2.info()["launches"],
                         "valu_insts_per_cell": pm2.get("valu_lane_insts_per_cell"), "valu_issue_busy": pm2.get("valu_issue_busy"),
                         "traffic": (int((2 * pm2["FETCH_SIZE_KiB"] + pm2["WRITE_SIZE_KiB"]) * 1024) if "FETCH_SIZE_KiB" in pm2 else None),
                         "counters": src2.get("status")}
            b2.free()
        out["other_workloads"] = extra
        if args.pe_seeds > 0 and args.workload != "150bp_w100_mixed_bins":
            out["pe_mixed_bins"] = pe_mixed_leg(host, ctx, args, cpu_affinity, args.pe_seeds)
        out["other_paths"] = side_paths(host, local_rank)
        if not args.no_e2e:
            # the same submit path when the caller's memory is NOT registered: host threads gather into pinned staging
            t3, a3 = host.synth_tasks(args.tasks, seed=1000, **spec)
            with host.BswContext(device=local_rank, kernel=args.kernel, streams=4, pack_threads=4, chunk_tasks=chunk) as c3:
                o3 = np.ones(len(t3), dtype=host.RESULT)
                c3.extend_pairs(params, t3, out=o3)
                t1 = time.perf_counter()
                r3 = c3.extend_pairs(params, t3, out=o3)
                d3 = time.perf_counter() - t1
            out["e2e_unregistered_memory"] = {"seeds_per_s": round(len(t3) / d3, 1), "gcups": round(cells_of(r3) / d3 / 1e9, 1),
                                              "pack_threads": 4, "path": "bsw_submit: pageable host memory, 4 threads gather into pinned staging"}
    ctx.close()
    harena.free()
    hout.free()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if out is not None:
        print(json.dumps(out), flush=True)


def pe_mixed_leg(host, ctx, args, cpu_affinity, n_seeds, steps=3):
    """BASELINE.json's metric is quoted on 150 bp PE batches (configs[2]: 10 M PE reads, mixed (qlen, tlen) bins through the
    batch manager).  The same measurement as the headline — inputs resident in HBM, HIP events around every bsw_run — on
    n_seeds left + right seeds with seed length ~U[19, 60] at a uniform position, 5 % junk reads, Ns: resident batches of
    32 x 128 Ki seeds (a batch holds < 4 GiB of bases), every chunk its own generator seed, generated side by side on this
    rank's CPUs while the previous group uploads.  Never part of `value`."""
    from concurrent.futures import ThreadPoolExecutor
    wl = "150bp_w100_mixed_bins"
    spec = dict(WORKLOADS[wl])
    params = host.default_params(variant=args.variant, zdrop=args.zdrop, w=spec["w"])
    chunk, per_batch = 131072, 32
    nchunks = (n_seeds + chunk - 1) // chunk
    groups = [list(range(i, min(i + per_batch, nchunks))) for i in range(0, nchunks, per_batch)]
    gb = per_batch * host.synth_arena_bound(chunk, **spec) + 4096
    arenas = [host.HostArena(gb) for _ in range(2 if len(groups) > 1 else 1)]
    pool = ThreadPoolExecutor(max_workers=max(1, min(len(cpu_affinity or [0]), 32)))
    t0 = time.perf_counter()

    def generate(gi):
        grp, ar = groups[gi], arenas[gi % len(arenas)]
        sizes = [min(chunk, n_seeds - c * chunk) for c in grp]
        tg = np.zeros(int(sum(sizes)), dtype=host.TASK)
        offs = np.concatenate([[0], np.cumsum([host.synth_arena_bound(sz, **spec) for sz in sizes])]).astype(np.int64)
        starts = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)

        def one(k):
            t, _ = host.synth_tasks(sizes[k], arena=ar.u8[int(offs[k]):], seed=7000 + grp[k], **spec)
            tg[int(starts[k]):int(starts[k]) + sizes[k]] = t
        return tg, [pool.submit(one, k) for k in range(len(grp))]

    batches, sides = [], 0
    nxt = generate(0)
    for gi in range(len(groups)):
        tg, futs = nxt
        for f in futs:
            f.result()
        nxt = generate(gi + 1) if gi + 1 < len(groups) else None
        batches.append(ctx.upload(params, tg))
        sides += int((tg["lqlen"] > 0).sum() + (tg["rqlen"] > 0).sum())
    pool.shutdown()
    setup_s = time.perf_counter() - t0
    for ar in arenas:
        ar.free()
    for b in batches:
        ctx.run(b)
    ctx.sync(); ctx.run_history()
    t1 = time.perf_counter()
    for _ in range(steps):
        for b in batches:
            ctx.run(b)
    ctx.sync()
    wall = time.perf_counter() - t1
    kms = ctx.run_history()
    step_ms = [float(sum(kms[i:i + len(batches)])) for i in range(0, len(kms), len(batches))]
    cells, launches = 0, 0
    for b in batches:
        r = ctx.download(b)
        cells += cells_of(r)
        launches += b.info()["launches"]
        b.free()
    ms = float(np.mean(step_ms))
    gc = cells / (ms * 1e-3) / 1e9
    pmc, pmc_src = pmc_summary(wl, None, key=wl + "@4194304")      # the pass on one 32 x 128 Ki-seed resident batch (tools/profile_r5.sh), else the 1 M-seed one
    tops = cells * VALU_OPS_PER_CELL / (ms * 1e-3) / 1e12
    return {"workload": wl, "config": "BASELINE.json configs[2] shape: %d PE seeds (left + right extension each), mixed bins via the batch manager" % n_seeds,
            "seeds": n_seeds, "extensions": sides, "resident_batches": len(batches), "gcups": round(gc, 1), "ms_per_step": round(ms, 3),
            "ms_per_step_wall": round(wall / steps * 1e3, 3), "steps": steps, "seeds_per_s": round(n_seeds / (ms * 1e-3), 1),
            "cells_per_step": cells, "kernel_launches_per_step": launches, "setup_s": round(setup_s, 1),
            "roofline": {"bound": "valu", "ops_per_cell": VALU_OPS_PER_CELL, "achieved": round(tops, 4), "peak": round(PEAK_VALU_TOPS, 2),
                         "unit": "T lane-ops/s", "frac": round(tops / PEAK_VALU_TOPS, 5),
                         "valu_insts_per_cell": pmc.get("valu_lane_insts_per_cell"), "valu_issue_busy": pmc.get("valu_issue_busy"),
                         "waves_per_simd_avg": pmc.get("waves_per_simd_avg"),
                         "traffic_per_counted_step": (int((2 * pmc["FETCH_SIZE_KiB"] + pmc["WRITE_SIZE_KiB"]) * 1024) if "FETCH_SIZE_KiB" in pmc else None),
                         "counters_source": dict(pmc_src, note="collected on a %d-seed resident batch of this workload (bench.py --workload %s --tasks %d)%s"
                                                 % (pmc.get("seeds_per_gpu", 0), wl, pmc.get("seeds_per_gpu", 0),
                                                    ": the size of this leg's batches (32 x 128 Ki seeds)" if pmc.get("seeds_per_gpu") == 4194304 else ""))}}


def harena_used(tasks):
    """bytes of byte-per-base sequence the tasks reference (what one submit moves over PCIe besides the records)"""
    return int(tasks["lqlen"].astype(np.int64).sum() + tasks["ltlen"].astype(np.int64)[tasks["lqlen"] > 0].sum()
               + tasks["rqlen"].astype(np.int64).sum() + tasks["rtlen"].astype(np.int64)[tasks["rqlen"] > 0].sum())


if __name__ == "__main__":
    main()
