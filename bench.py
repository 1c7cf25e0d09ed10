#!/usr/bin/env python3
"""bench.py — GCUPS of the seed-extension hot path on N MI355X GPUs (one process per GPU).

A "step" = one pass of the hot path (left+right ksw_extend2 with band retry and the
mem_chain2aln decision) over one device-resident task batch.  At N=1 the workload is
BASELINE.json configs[1]: 1M synthetic 150 bp reads, w=100, single (qlen,tlen)=(131,257)
bin.  For N>1 every rank owns its own batch of the same shape (weak scaling, the
per-read task pool is sharded, no data-path collective).

Prints ONE JSON line on rank 0.  `value` = DP cells actually evaluated (exactly as the CPU
algorithm iterates them) / wall time / 1e9, summed over ranks, inputs already in HBM.
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
# int32 VALU roof: 256 CUs x 4 SIMDs x 16 lanes/clk x 2.4 GHz = 39.3 T lane-ops/s.  (A wave64 integer
# VALU op holds its SIMD for 4 cycles: measured, SQ_ACTIVE_INST_VALU == SQ_INSTS_VALU quad-cycles in
# profiles/r1/*_pmc_summary.json; the 157 TFLOP/s fp32 figure counts packed FMA.)
PEAK_INT32_TOPS = 256 * 64 * 2.4e9 / 1e12
PEAK_HBM_GBS = 8000.0
PMC_FILE = os.path.join(ROOT, "profiles", "pmc_latest.json")   # rocprofv3 --pmc summary of this same command


def pmc_traffic(workload, tasks):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (FETCH_SIZE/WRITE_SIZE in KiB;
    gfx950 correction of MI355X_MICROARCH.md §HBM: FETCH_SIZE x 2), or None if no matching profile."""
    try:
        j = json.load(open(PMC_FILE))
        if j.get("workload") != workload or j.get("seeds_per_gpu") != tasks:
            return None
        return int((2 * j["FETCH_SIZE_KiB"] + j["WRITE_SIZE_KiB"]) * 1024)
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--tasks", type=int, default=1_000_000, help="seeds per GPU per step")
    ap.add_argument("--workload", default="150bp_w100_single_bin", choices=sorted(WORKLOADS))
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--zdrop", type=int, default=100)
    ap.add_argument("--cpu-sample", type=int, default=1_000_000, help="seeds timed on the CPU oracle (rank 0, N=1): ~35 s of CPU work on 16 threads")
    ap.add_argument("--cpu-threads", type=int, default=0, help="0 = min(affinity, 16): the 1-GPU box's CPU share")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the 250 bp and mixed-bin side measurements (N=1 only, outside the timed region)")
    ap.add_argument("--check", type=int, default=100_000, help="seeds checked bit-exact against the oracle after timing")
    ap.add_argument("--spec", action="append", default=[], help="override a generator field, e.g. --spec n_rate=0 (experiments)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend for the barrier / timing reduction (nccl = RCCL; gloo only to rehearse N>1 on a 1-GPU box)")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal: every rank uses device 0 (needs --backend gloo)")
    ap.add_argument("--kernel", type=int, default=0, help="0 auto (batch manager picks per bin), 1 wave-per-task only, 2 force lane bins")
    args = ap.parse_args()

    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
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

    pkg = graft.load_package()
    host = pkg.host
    spec = dict(WORKLOADS[args.workload])
    for kv in args.spec:
        key, val = kv.split("=")
        spec[key] = type(spec[key])(float(val))
    params = host.default_params(variant=args.variant, zdrop=args.zdrop, w=spec["w"])
    tasks, arena = host.synth_tasks(args.tasks, seed=1000 + rank, **spec)

    ctx = host.BswContext(device=local_rank, kernel=args.kernel)
    batch = ctx.upload(params, tasks)            # inputs resident in HBM before the timed region

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        ctx.run(batch)
    ctx.sync()
    ctx.run_history()                            # reset per-run event history
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ctx.run(batch)
    ctx.sync()
    barrier()
    dt = time.perf_counter() - t0
    kern_ms = ctx.run_history()                  # HIP events on the library's own stream, one pair per step

    res = ctx.download(batch)
    info = batch.info()
    cells = int(res["left"]["cells"].astype(np.int64).sum() + res["right"]["cells"].astype(np.int64).sum())
    ext_calls = int((tasks["lqlen"] > 0).sum() + (tasks["rqlen"] > 0).sum()
                    + (res["left"]["aw"] > spec["w"]).sum() + (res["right"]["aw"] > spec["w"]).sum())
    nominal = int((tasks["lqlen"].astype(np.int64) * tasks["ltlen"]).sum() + (tasks["rqlen"].astype(np.int64) * tasks["rtlen"]).sum())

    if dist is not None:
        v = torch.tensor([dt, float(cells), float(ext_calls), float(len(tasks)), float(nominal)], dtype=torch.float64, device=red_dev)
        tmax = v[:1].clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(v, op=dist.ReduceOp.SUM)
        dt_all = float(tmax.item())
        cells_all, ext_all, tasks_all, nominal_all = (float(x) for x in v[1:].tolist())
    else:
        dt_all, cells_all, ext_all, tasks_all, nominal_all = dt, float(cells), float(ext_calls), float(len(tasks)), float(nominal)

    out = None
    if rank == 0:
        gcups = cells_all * args.steps / dt_all / 1e9
        kavg_ms = float(np.mean(kern_ms)) if kern_ms else float("nan")
        alg_bytes = info["in_bytes"] + info["out_bytes"]          # per launch: packed seq + task records + results
        out = {
            "metric": "GCUPS (seed-extension DP cells/s, 150 bp PE)", "value": round(gcups, 3), "unit": "GCUPS",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt_all / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "int32", "data": "synthetic",
            "config": {"workload": args.workload, "seeds_per_gpu": args.tasks, "read_len": spec["read_len"],
                       "band_w": spec["w"], "zdrop": args.zdrop, "variant": "H" if args.variant == 0 else "M",
                       "sharding": "per-read task shard, no collective" if world > 1 else "single GPU",
                       "kernel_launches_per_step": info["launches"]},
            "extensions_per_s": round(ext_all * args.steps / dt_all, 1),
            "seeds_per_s": round(tasks_all * args.steps / dt_all, 1),
            "cells_per_step": cells_all,
            "nominal_gcups_qlen_x_tlen": round(nominal_all * args.steps / dt_all / 1e9, 3),
            "roofline": {
                "bound": "hbm", "achieved": round(alg_bytes / (kavg_ms * 1e-3) / 1e9, 3), "peak": PEAK_HBM_GBS,
                "unit": "GB/s", "frac": round(alg_bytes / (kavg_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 6),
                "traffic": pmc_traffic(args.workload, args.tasks),
                "kernel_ms_avg": round(kavg_ms, 4), "algorithmic_bytes_per_launch": alg_bytes,
                "note": "integer max/add DP at ~0.02 B/cell: neither HBM nor MFMA binds; see roofline_valu",
            },
            "roofline_valu": {
                "bound": "valu-int32", "ops_per_cell": VALU_OPS_PER_CELL,
                "achieved": round(cells * VALU_OPS_PER_CELL / (kavg_ms * 1e-3) / 1e12, 4), "peak": round(PEAK_INT32_TOPS, 2),
                "unit": "Tops/s", "frac": round(cells * VALU_OPS_PER_CELL / (kavg_ms * 1e-3) / 1e12 / PEAK_INT32_TOPS, 5),
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            orc = graft.load_oracle()
            ncpu = args.cpu_threads or min(len(os.sched_getaffinity(0)), 16)
            ns = min(args.cpu_sample, len(tasks))
            t1 = time.perf_counter()
            ref = orc.pair_batch(params, tasks[:ns], nthreads=ncpu)
            dcpu = time.perf_counter() - t1
            ccells = int(ref["left"]["cells"].astype(np.int64).sum() + ref["right"]["cells"].astype(np.int64).sum())
            n1 = min(ns, 50_000)
            t1 = time.perf_counter()
            orc.pair_batch(params, tasks[:n1], nthreads=1)
            d1 = time.perf_counter() - t1
            c1 = int(ref["left"]["cells"][:n1].astype(np.int64).sum() + ref["right"]["cells"][:n1].astype(np.int64).sum())
            out["cpu_baseline"] = {
                "value": round(ccells / dcpu / 1e9, 4), "unit": "GCUPS", "cores": ncpu, "kind": "port",
                "sample": "first %d seeds of the same batch, scalar C oracle (-O3 -march=x86-64-v3), %d pthreads" % (ns, ncpu),
                "single_thread_gcups": round(c1 / d1 / 1e9, 4),
            }
            nchk = min(args.check, ns)
            out["parity_spot_check"] = {"seeds": nchk, "bit_exact": bool(res[:nchk].tobytes() == ref[:nchk].tobytes())}
    batch.free()
    if out is not None and world == 1 and not args.no_extra and not args.spec:
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
            c2 = int(r2["left"]["cells"].astype(np.int64).sum() + r2["right"]["cells"].astype(np.int64).sum())
            extra[wl] = {"gcups": round(c2 / (ms * 1e-3) / 1e9, 1), "ms_per_step": round(ms, 3), "seeds": args.tasks}
            b2.free()
        out["other_workloads"] = extra
    ctx.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if out is not None:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
