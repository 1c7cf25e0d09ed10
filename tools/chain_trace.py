#!/usr/bin/env python3
"""Kernel timeline of a few bsw_run steps from a rocprofv3 --kernel-trace of tools/w250_check.py (usage: chain_trace.py DIR [step ...])."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
seq = [(r["Kernel_Name"][5:36], int(r["Grid_Size_X"]) // 64, int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
fin = [i for i, s in enumerate(seq) if "pair_finalize" in s[0]]
for st in [int(x) for x in sys.argv[2:]] or [6]:
    print("step", st, "span %.3f ms" % ((seq[fin[st]][3] - seq[fin[st - 1]][3]) / 1e6))
    t0 = None
    for s in seq[fin[st - 1] + 2:fin[st] + 1]:
        if any(x in s[0] for x in ("fillBuffer", "bin_", "pack", "streamOpsWrite")):
            continue
        if t0 is None:
            t0 = s[2]
        print("  %-32s waves %6d  start %8.3f  end %8.3f  dur %7.3f" % (s[0], s[1], (s[2] - t0) / 1e6, (s[3] - t0) / 1e6, (s[3] - s[2]) / 1e6))
