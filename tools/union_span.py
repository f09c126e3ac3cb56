#!/usr/bin/env python3
"""Per-step kernel times from a rocprofv3 --kernel-trace table.

    tools/union_span.py <..._kernel_trace.csv> [kernel-name substrings to group, comma separated]

A step of bench.py launches several kernels, some of them side by side on two streams, so neither the
sum nor the average of the per-kernel durations is the step's kernel time.  This prints, for every group
of dispatches, the UNION of their [start, end) intervals per step (steps are found as runs of dispatches
separated by the group's first kernel name re-appearing) and the mean / median over steps - the figure
that bench.py's HIP-event bracket (`kernel_ms`) measures.
"""
import csv
import statistics
import sys


def union(iv):
    iv = sorted(iv)
    tot, cs, ce = 0, None, None
    for s, e in iv:
        if cs is None:
            cs, ce = s, e
        elif s <= ce:
            ce = max(ce, e)
        else:
            tot += ce - cs
            cs, ce = s, e
    if cs is not None:
        tot += ce - cs
    return tot


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    groups = sys.argv[2].split(",") if len(sys.argv) > 2 else ["k_full_hier,k_pack_fenc,k_full_items", "k_search"]
    if len(sys.argv) > 2:
        groups = sys.argv[2].split(";")
    disp = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda t: t[0])
    for g in groups:
        keys = g.split(",")
        sel = [d for d in disp if any(k in d[2] for k in keys)]
        if not sel:
            continue
        # a new step starts when a dispatch begins after every dispatch of the running step has ended AND
        # its kernel is one the running step already holds (the same launch sequence repeating)
        steps, cur, names = [], [], set()
        for d in sel:
            if cur and d[2] in names and d[0] >= max(c[1] for c in cur) - 0:
                steps.append(cur)
                cur, names = [], set()
            cur.append(d)
            names.add(d[2])
        if cur:
            steps.append(cur)
        spans = [union([(s, e) for s, e, _ in st]) / 1e6 for st in steps]
        per_kernel = {}
        for s, e, n in sel:
            per_kernel.setdefault(n[:70], []).append((e - s) / 1e6)
        print("group [%s]: %d dispatches in %d steps; union span per step: mean %.4f ms, median %.4f ms, min %.4f, max %.4f"
              % (g, len(sel), len(spans), statistics.mean(spans), statistics.median(spans), min(spans), max(spans)))
        for n, v in sorted(per_kernel.items()):
            print("    %-72s n=%4d  mean %.4f ms  median %.4f ms" % (n, len(v), statistics.mean(v), statistics.median(v)))


if __name__ == "__main__":
    main()
