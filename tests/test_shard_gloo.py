"""N>1 path on CPU: 2 gloo ranks shard the per-read task pool (no data-path collective), the
gathered shards must equal the unsharded result batch and the bench-style aggregation must add up.
The per-rank compute here is the CPU oracle standing in for the GPU (test infrastructure)."""
import os
import socket
import sys

import numpy as np
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, n, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import __graft_entry__ as graft
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pkg = graft.load_package()
    host, orc = pkg.host, graft.load_oracle()
    tasks, arena = host.synth_tasks(n, seed=5, seed_at_start=0, seed_len_min=19, seed_len_max=60, junk_frac=0.1)
    p = host.default_params()
    mine = host.shard_indices(n, world, rank, chunk=64)
    res = orc.pair_batch(p, tasks[mine])
    cells = int(res["left"]["cells"].sum() + res["right"]["cells"].sum())
    v = torch.tensor([float(cells), float(len(mine))], dtype=torch.float64)
    dist.all_reduce(v)
    gathered = [None] * world
    dist.all_gather_object(gathered, (mine, res.tobytes()))
    dist.barrier()
    if rank == 0:
        full = np.zeros(n, dtype=host.RESULT)
        for idx, blob in gathered:
            full[idx] = np.frombuffer(blob, dtype=host.RESULT)
        ref = orc.pair_batch(p, tasks)
        q.put((full.tobytes() == ref.tobytes(), v.tolist(),
               int(ref["left"]["cells"].sum() + ref["right"]["cells"].sum())))
    dist.destroy_process_group()


def test_two_rank_shard_matches_unsharded():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    n = 1000
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    same, agg, ref_cells = q.get(timeout=120)
    for pr in procs:
        pr.join(60)
        assert pr.exitcode == 0
    assert same
    assert int(agg[0]) == ref_cells and int(agg[1]) == n


def test_shard_indices_partition(host):
    for n, world, chunk in ((0, 2, 64), (1, 2, 64), (1000, 3, 64), (65536 * 3 + 5, 8, 65536)):
        seen = np.concatenate([host.shard_indices(n, world, r, chunk) for r in range(world)])
        assert len(seen) == n and (np.sort(seen) == np.arange(n)).all()
    # contiguous per-read chunks, round-robin over ranks
    assert (host.shard_indices(300, 2, 1, 100) == np.arange(100, 200)).all()


def _bench(*args, timeout=300):
    import json
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(args), capture_output=True, text=True, timeout=timeout,
                       env={k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")})
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    return r.returncode, (json.loads(lines[-1]) if lines else None), r.stderr


def test_bench_gpus_flag_starts_one_rank_per_gpu():
    """`python bench.py --gpus N` without torchrun must start N ranks itself (it used to measure one GPU):
    the launcher path, rendezvous and relay of rank 0's line, checked without a GPU through --dry-run."""
    rc, line, err = _bench("--gpus", "2", "--dry-run", "--preset", "configs3")
    assert rc == 0, err
    assert line["n_gpus"] == 2 and line["ranks_seen"] == 2
    assert line["scaling"] == "strong" and line["pool"] == 100_000_000 and line["workload"] == "150bp_w100_mixed_bins"
    rc, line, err = _bench("--gpus", "3", "--dry-run", "--preset", "configs4", "--pool", "5")
    assert rc == 0 and line["n_gpus"] == 3 and line["workload"] == "250bp_w500" and line["scaling"] == "weak" and line["pool"] == 5


def test_bench_launcher_reports_a_failed_rank():
    """no GPU here: every rank fails its `needs a GPU` assertion and the launcher must exit non-zero, not print a line"""
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("needs a box without a GPU")
    rc, line, err = _bench("--gpus", "2", "--steps", "1", "--warmup", "0")
    assert rc != 0 and line is None and "ranks failed" in err
