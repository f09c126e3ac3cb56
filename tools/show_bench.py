#!/usr/bin/env python3
"""Print the headline numbers of a bench.py JSON line."""
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
r = d["roofline"]
print("value %.4g %s  ms_per_step %.4f  exhaustive kernels %.4f ms (frac %.3f, with frames in flight %.3f)  tail %.4f ms  parity %s" %
      (d["value"], d["unit"], d["ms_per_step"], r["kernel_ms"], r["frac"], r.get("frames_in_flight_frac", 0), r.get("search_kernel_ms", 0),
       d.get("parity_sample_ok")))
if "hex_search" in d:
    h = d["hex_search"]
    print("hex: %.4g evals/s  ms/frame %.4f  k_search %.4f ms  valu frac %.4f" % (h["value"], h["ms_per_frame"], h["roofline"]["kernel_ms"], h["roofline"]["frac"]))
    if "lookahead_ordered" in h:
        print("  hex ordered by the lookahead field: ms/frame %.4f  k_search %.4f ms  equal %s" % (h["lookahead_ordered"]["ms_per_frame"], h["lookahead_ordered"]["kernel_ms"], h["lookahead_ordered"]["results_equal"]))
    print("dia: %.4g evals/s  ms/frame %.4f" % (d["dia_search"]["value"], d["dia_search"]["ms_per_frame"]))
    p = d["pcie_inclusive"]
    print("pcie-inclusive ms/frame: full %.4f  hex %.4f" % (p["full_search"]["ms_per_frame"], p["hex_search"]["ms_per_frame"]))
    if "pipelined" in p:
        q = p["pipelined"]
        print("  pipelined: full %.4f (%.3f x resident, field equal %s)  hex %.4f" % (q["full_search"]["ms_per_frame"], q["vs_resident"],
              q["full_search"]["field_equals_resident_run"], q["hex_search"]["ms_per_frame"]))
if "sa8d_full_4k10" in d:
    s = d["sa8d_full_4k10"]
    print("sa8d 4K 10-bit: %.4g evals/s  ms/frame %.3f  k_full_had %.3f ms  tail %.3f ms" %
          (s["value"], s["ms_per_frame"], s["roofline"]["kernel_ms"], s["roofline"]["search_kernel_ms"]))
    if "all_partitions" in s:
        a = s["all_partitions"]
        print("  all %d partitions: %.4g evals/s  ms/frame %.3f  k_full_had %.3f ms  tail %.3f ms" %
              (a["jobs_per_frame"], a["value"], a["ms_per_frame"], a["kernel_ms"], a["search_kernel_ms"]))
    if "all_shapes" in s:
        a = s["all_shapes"]
        print("  all %d shapes' PUs: %.4g evals/s  ms/frame %.3f  k_full_had %.3f ms  tail %.3f ms  %s" %
              (a["jobs_per_frame"], a["value"], a["ms_per_frame"], a["kernel_ms"], a["search_kernel_ms"], a.get("parity_sample")))
if "lookahead" in d:
    l = d["lookahead"]
    print("lookahead: %d pairs %.3f ms  (%.0f pairs/s, %.4g evals/s)  single pair %.3f ms  lowres init %.1f us  cpu %s  parity %s" %
          (l["pairs_per_call"], l["kernel_ms_per_call"], l["frame_pairs_per_s"], l["value"], l["single_pair_kernel_ms"],
           l["lowres_init_us_per_picture"], l.get("cpu_baseline", {}).get("ms_per_frame_pair"), l.get("parity_ok")))
if d.get("cpu_baseline"):
    c = d["cpu_baseline"]
    print("cpu_baseline %.4g %s  cores %d  %s  (%s)" % (c["value"], c["unit"], c["cores"], c["kind"], c.get("path", "")))
