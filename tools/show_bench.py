#!/usr/bin/env python3
"""Print the headline numbers of a bench.py JSON line."""
import json, sys
d = json.load(open(sys.argv[1]))
print("value %.4g %s  ms_per_step %.4f  k_search ms %.4f  roofline frac %.4g  parity %s" %
      (d["value"], d["unit"], d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["frac"], d.get("parity_sample_ok")))
print("dia %.4g ms/frame %.4f | pcie ms/frame %.4f" % (d["dia_search"]["value"], d["dia_search"]["ms_per_frame"], d["pcie_inclusive"]["full_search"]["ms_per_frame"]))
f = d["full_search"]
print("full: %.4g evals/s  ms/frame %.3f  k_full ms %.3f frac %.3f  tail ms %.3f" %
      (f["value"], f["ms_per_frame"], f["roofline"]["kernel_ms"], f["roofline"]["frac"], f["roofline"]["search_kernel_ms"]))
if "lookahead" in d:
    l = d["lookahead"]
    print("lookahead: %d pairs %.3f ms  (%.0f pairs/s, %.4g evals/s)  single pair %.3f ms  lowres init %.1f us  cpu %s  parity %s" %
          (l["pairs_per_call"], l["kernel_ms_per_call"], l["frame_pairs_per_s"], l["value"], l["single_pair_kernel_ms"],
           l["lowres_init_us_per_picture"], l.get("cpu_baseline", {}).get("ms_per_frame_pair"), l.get("parity_ok")))
for k in ("cpu_baseline", "cpu_baseline_full_search", "cpu_baseline_all_cores"):
    if k in d:
        print(k, "%.4g" % d[k]["value"], d[k]["unit"], d[k]["cores"], d[k]["kind"])
