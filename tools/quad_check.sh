#!/bin/bash
# parity tests that run through the general (wave / quad) kernels, then the wave-vs-quad-vs-lane crossover and the wire-format rate
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$1; mkdir -p $OUT
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/gputest.log 2>&1; tail -3 $OUT/gputest.log
python3 tools/crossover.py > $OUT/cross_quad.json 2>/dev/null
BSW_QUAD=0 python3 tools/crossover.py > $OUT/cross_wave.json 2>/dev/null
python3 tools/side_rates.py wire 2>/dev/null | tail -1 > $OUT/wire_quad.json
BSW_QUAD=0 python3 tools/side_rates.py wire 2>/dev/null | tail -1 > $OUT/wire_wave.json
python3 - $OUT <<PY
import json, sys
o = sys.argv[1]
a = json.load(open(o + "/cross_quad.json")); b = json.load(open(o + "/cross_wave.json"))
for x, y in zip(a, b): print(x["seeds"], "quad", x["wave_ms"], "wave", y["wave_ms"], "lane", x["lane_ms"], "speedup", round(y["wave_ms"] / x["wave_ms"], 2))
for f in ("wire_quad", "wire_wave"):
    j = json.load(open(o + "/%s.json" % f)); print(f, [(r["batches_in_flight"], r["seeds_per_s"]) for r in j["runs"]])
PY
