#!/bin/bash
# usage: tools/pmc_pass.sh <tag> "<counters>" <size> <method> <subme>
# one rocprofv3 counter pass over tools/run_full_once.py; prints per-kernel sums of each counter
tag=$1; ctrs=$2; shift 2
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
rm -rf $out
timeout -s KILL 120 rocprofv3 --pmc $ctrs --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/tools/run_full_once.py "$@" 2 > $out.log 2>&1 || { echo "pass $tag failed"; tail -5 $out.log; exit 1; }
f=$(find $out -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"][:60]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    n[(k, r["Counter_Name"])] += 1
for k in acc:
    if "k_search" in k or "k_full" in k or "k_pack" in k or "k_field" in k or "k_look" in k:
        for c, v in acc[k].items():
            print("%-62s %-24s %14.0f  (%d dispatches)" % (k, c, v, n[(k, c)]))
PY
