#!/usr/bin/env python3
"""Summarise a rocprofv3 kernel + memory-copy trace: per-op totals and the busy/idle timeline of the last pass."""
import csv, glob, os, sys, collections
d = sys.argv[1]
GAP = int(float(os.environ.get("TRACE_GAP_MS", "1")) * 1e6)      # idle time that separates two passes
DUMP = len(sys.argv) > 2 and sys.argv[2] == "--dump"
ev = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K:" + r["Kernel_Name"][:40]))
for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C:" + r.get("Direction", r.get("Name", "copy"))[:40]))
ev.sort()
if not ev:
    print("no events"); sys.exit(0)
# split into passes by gaps > 1 ms
passes, cur = [], [ev[0]]
for e in ev[1:]:
    if e[0] - max(x[1] for x in cur) > GAP:
        passes.append(cur); cur = [e]
    else:
        cur.append(e)
passes.append(cur)
print("passes:", len(passes), [round((max(x[1] for x in p) - p[0][0]) / 1e6, 2) for p in passes])
last = passes[-1]
t0 = last[0][0]; t1 = max(x[1] for x in last)
if DUMP:
    for s_, e_, nme in last:
        print("  +%8.3f ms  %7.3f ms  %s" % ((s_ - t0) / 1e6, (e_ - s_) / 1e6, nme))
tot = collections.defaultdict(lambda: [0, 0])
for s, e, nme in last:
    tot[nme][0] += e - s; tot[nme][1] += 1
for k, v in sorted(tot.items(), key=lambda kv: -kv[1][0]):
    print("%-46s %8.3f ms  x%d" % (k, v[0] / 1e6, v[1]))
def union(evs):
    evs = sorted(evs); tot = 0; cs, ce = evs[0]
    for s, e in evs[1:]:
        if s > ce: tot += ce - cs; cs, ce = s, e
        else: ce = max(ce, e)
    return tot + ce - cs
kern = [(s, e) for s, e, nme in last if nme.startswith("K:")]
lane = [(s, e) for s, e, nme in last if "lane_kernel" in nme]
h2d = [(s, e) for s, e, nme in last if nme.startswith("C:") and "HOST_TO_DEVICE" in nme.upper()]
print("pass span %.3f ms; any-kernel busy %.3f ms; lane-kernel busy %.3f ms; H2D busy %.3f ms" % ((t1 - t0) / 1e6, union(kern) / 1e6, union(lane) / 1e6 if lane else 0, union(h2d) / 1e6 if h2d else 0))
if lane:
    print("first lane kernel starts at +%.3f ms; last ends at +%.3f ms" % ((min(s for s, e in lane) - t0) / 1e6, (max(e for s, e in lane) - t0) / 1e6))
    # concurrency histogram of lane kernels
    pts = sorted([(s, 1) for s, e in lane] + [(e, -1) for s, e in lane]); c = 0; prev = pts[0][0]; hist = collections.Counter()
    for t, dlt in pts:
        hist[c] += t - prev; prev = t; c += dlt
    print("lane-kernel concurrency (ms at k kernels in flight):", {k: round(v / 1e6, 3) for k, v in sorted(hist.items())})
