#!/bin/bash
# HBM traffic of the search kernels of one benchmark step: two rocprofv3 counter passes over bench.py
# (FETCH_SIZE and WRITE_SIZE cannot share a pass), summed per step over the kernels of the timed region.
# Writes profiles-style JSON to gpurun_out/pmc_traffic.json.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $R/gpurun_out/pmc_$c
    timeout -s KILL 200 rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/pmc_$c -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu > $R/gpurun_out/pmc_$c.log 2>&1 || { echo "pass $c failed"; tail -5 $R/gpurun_out/pmc_$c.log; exit 1; }
done
python3 - $R <<'PY'
import csv, sys, glob, json, collections
R = sys.argv[1]
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(R + "/gpurun_out/pmc_%s/**/*counter_collection.csv" % c, recursive=True)[0]
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == c:
            per[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    out[c] = {k: {"dispatches": len(v), "sum": sum(v), "mean": sum(v) / len(v)} for k, v in per.items()}
json.dump(out, open(R + "/gpurun_out/pmc_traffic_raw.json", "w"), indent=1)
for c in out:
    for k, v in out[c].items():
        print(c, k[:70], v)
PY
