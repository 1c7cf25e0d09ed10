#!/usr/bin/env python3
"""Summarise a tools/profile_r1.sh output directory: per-kernel mean counter values per launch."""
import collections, csv, glob, json, sys
d = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else "bsw"
out = collections.OrderedDict()
for f in sorted(glob.glob(d + "/trace/*/*_kernel_stats.csv")):
    for r in csv.DictReader(open(f)):
        if pat in r["Name"]:
            out.setdefault(r["Name"][:60], {})["avg_ns"] = float(r["AverageNs"]); out[r["Name"][:60]]["calls"] = int(r["Calls"])
for f in sorted(glob.glob(d + "/pmc_*/*/*_counter_collection.csv")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            agg[(r["Kernel_Name"][:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
            out.setdefault(r["Kernel_Name"][:60], {}).update(vgpr=r.get("VGPR_Count"), sgpr=r.get("SGPR_Count"), lds=r.get("LDS_Block_Size"), scratch=r.get("Scratch_Size"))
    for (kn, c), v in agg.items():
        out.setdefault(kn, {})[c] = sum(v) / len(v)
print(json.dumps(out, indent=1))
