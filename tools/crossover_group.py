#!/usr/bin/env python3
"""The three ways a resident batch can run, by batch size: general kernels (one / four seeds per wavefront), the GROUP
kernel (bsw_lane2g_kernel: 16 seeds per wavefront; one launch per side, or both sides of a seed in one launch) and the lane kernels (128 per wavefront).  PE mixed bins and the
150 bp single bin.  The switches are read once per process: every (mode, workload) runs in a child process.
python tools/crossover_group.py > profiles/r6/crossover_group.json"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import json, os, sys
import numpy as np
sys.path.insert(0, %(root)r)
import __graft_entry__ as graft
host = graft.load_package().host
p = host.default_params()
wl, kern = sys.argv[1], int(sys.argv[2])
spec = dict(seed_len_min=19, seed_len_max=60, seed_at_start=0, junk_frac=0.05) if wl == "pe_mixed" else {}
if os.environ.get("CROSSOVER_READ_LEN"):                            # e.g. 250: BASELINE configs[4] (w = 500)
    spec["read_len"] = int(os.environ["CROSSOVER_READ_LEN"])
    if spec["read_len"] > 200:
        spec["w"] = 500
        p = host.default_params(w=500)
if os.environ.get("CROSSOVER_N_RATE"):
    spec["n_rate"] = float(os.environ["CROSSOVER_N_RATE"])          # (bench.py's workloads have 0.001)
tasks, arena = host.synth_tasks(262144, seed=51, **spec)
row = {}
for n in (1024, 2048, 4096, 8192, 13104, 16384, 24576, 32768, 49152, 65536, 131072, 262144):
    with host.BswContext(device=0, kernel=kern) as ctx:
        b = ctx.upload(p, tasks[:n])
        for _ in range(3):
            ctx.run(b)
        ctx.sync(); ctx.run_history()
        for _ in range(10):
            ctx.run(b)
        ctx.sync()
        row[n] = round(float(np.median(ctx.run_history())), 4)
        b.free()
print(json.dumps(row))
"""
out = {}
for wl in ("pe_mixed", "single_bin"):
    out[wl] = {}
    modes = (("general_ms", 1, {}), ("group_ms", 0, {"BSW_GROUP": "1", "BSW_GROUP_FUSE": "0"}), ("group_fused_ms", 0, {"BSW_GROUP": "1", "BSW_GROUP_FUSE": "1"}),
             ("lane_ms", 2, {"BSW_GROUP": "0", "BSW_LANE_FUSE": "0"}), ("lane_fused_ms", 2, {"BSW_GROUP": "0", "BSW_LANE_FUSE": "1"}), ("auto_ms", 0, {}), ("auto_nosplit_ms", 0, {"BSW_NSPLIT": "0"}))
    if len(sys.argv) > 1:
        modes = tuple(m for m in modes if m[0] in sys.argv[1:])
    for name, kern, env in modes:
        r = subprocess.run([sys.executable, "-c", CHILD % dict(root=ROOT), wl, str(kern)], env=dict(os.environ, **env), capture_output=True, text=True)
        out[wl][name] = json.loads(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 else r.stderr[-500:]
print(json.dumps(out, indent=1))
