#!/usr/bin/env python3
"""The numbers of a bench.py line that matter while iterating: print_bench.py <file with the JSON line>"""
import json, sys
d = json.loads([l for l in open(sys.argv[1]).read().strip().splitlines() if l.startswith("{")][-1])
print("headline %.1f GCUPS  %.4f ms/step  frac %.4f  launches %s" % (d["value"], d["ms_per_step"], d["roofline"]["frac"], d["config"]["kernel_launches_per_step"]))
if "parity_spot_check" in d:
    print("parity", d["parity_spot_check"]["bit_exact"], d["parity_spot_check"]["cells_gpu_eq_cpu_on_sample"], "avx2==gpu", d["cpu_baseline"]["bit_exact_vs_gpu"], "cpu", d["cpu_baseline"]["value"])
for k, v in d.get("other_workloads", {}).items():
    print("  %-24s %.1f GCUPS  %.3f ms  frac %.4f" % (k, v["gcups"], v["ms_per_step"], v["roofline_frac"]))
if "pe_mixed_bins" in d:
    v = d["pe_mixed_bins"]
    print("  pe_mixed_bins %d seeds   %.1f GCUPS  %.3f ms  frac %.4f" % (v["seeds"], v["gcups"], v["ms_per_step"], v["roofline"]["frac"]))
for k in ("e2e", "e2e_packed_input", "e2e_device_reference"):
    if k in d:
        print("  %-22s %.1f M seeds/s  ratio %.3f  spread %s" % (k, d[k]["seeds_per_s"] / 1e6, d[k]["ratio_to_hbm_resident"], d[k].get("spread", {}).get("max_over_min")))
if "other_paths" in d:
    print("  wire %.1f M seeds/s  align %.2f M/s  global %.2f M/s" % (d["other_paths"]["wire_format"]["seeds_per_s"] / 1e6, d["other_paths"]["ksw_align2"]["alignments_per_s"] / 1e6, d["other_paths"]["ksw_global2"]["alignments_per_s"] / 1e6))
