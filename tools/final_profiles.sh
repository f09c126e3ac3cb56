#!/bin/bash
# Round profile set, run on the GPU box from the repository root:  bash tools/final_profiles.sh <tag>
#   1. rocprofv3 --kernel-trace --stats of the default bench.py            -> gpurun_out/<tag>_kernel_stats_bench.csv (+ log)
#   2. the same of `bench.py --only-primary` (HEX + subme 5 leg alone)     -> gpurun_out/<tag>_kernel_stats_primary.csv (+ log)
#   3. HBM traffic counters (FETCH_SIZE / WRITE_SIZE, one pass each) of the primary leg -> gpurun_out/<tag>_pmc_traffic.json
# Copy the files you want kept from gpurun_out/ into profiles/.
set -u
TAG=${1:-r01}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
run_stats() {   # name, bench arguments...
    local name=$1; shift
    rm -rf $R/gpurun_out/prof_$name
    timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$name -- python3 $R/bench.py "$@" > $R/gpurun_out/${TAG}_bench_under_rocprof_$name.log 2>&1 || { echo "stats run $name failed"; tail -5 $R/gpurun_out/${TAG}_bench_under_rocprof_$name.log; return 1; }
    local f=$(find $R/gpurun_out/prof_$name -name '*kernel_stats.csv' | head -1)
    cp "$f" $R/gpurun_out/${TAG}_kernel_stats_$name.csv && echo "== $name" && cut -c1-150 $R/gpurun_out/${TAG}_kernel_stats_$name.csv | head -14
}
run_stats bench --steps 20 --warmup 3 --no-cpu || exit 1
run_stats primary --steps 20 --warmup 3 --only-primary || exit 1
for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $R/gpurun_out/pmc_$c
    timeout -s KILL 200 rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/pmc_$c -- python3 $R/bench.py --steps 5 --warmup 1 --only-primary > $R/gpurun_out/pmc_$c.log 2>&1 || { echo "pass $c failed"; tail -5 $R/gpurun_out/pmc_$c.log; exit 1; }
done
python3 - $R $TAG <<'PY'
import csv, sys, glob, json, collections
R, TAG = sys.argv[1], sys.argv[2]
per = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(R + "/gpurun_out/pmc_%s/**/*counter_collection.csv" % c, recursive=True)[0]
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == c:
            d[(r["Kernel_Name"].split("(")[0].replace("void xa::", ""), r.get("Grid_Size", "?"))].append(float(r["Counter_Value"]))
    per[c] = d
kern = {}
for k in sorted(set(per["FETCH_SIZE"]) | set(per["WRITE_SIZE"])):
    fv, wv = per["FETCH_SIZE"].get(k, []), per["WRITE_SIZE"].get(k, [])
    kern["%s grid=%s" % k] = {"fetch_KiB": round(sum(fv) / max(len(fv), 1), 1), "write_KiB": round(sum(wv) / max(len(wv), 1), 1),
                              "dispatches": max(len(fv), len(wv))}
# one step of the primary leg = one launch of every k_search class it uses
step = [v for k, v in kern.items() if k.startswith("k_search")]
raw = sum((v["fetch_KiB"] + v["write_KiB"]) * 1024 for v in step)
dbl = sum((2 * v["fetch_KiB"] + v["write_KiB"]) * 1024 for v in step)
out = {"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over `bench.py --steps 5 --warmup 1 --only-primary` "
               "(tools/final_profiles.sh); KiB per dispatch as reported (mean over the dispatches of one kernel and grid size). "
               "FETCH_SIZE on gfx950 counts 64 B per 128-B request for wide streaming reads; these kernels read dwords, a width "
               "for which the counter is uncalibrated, so both the raw and the doubled figure are given.",
       "kernels": kern, "k_search_u8_bytes_per_launch": int(raw), "k_search_u8_bytes_per_launch_fetch_doubled": int(dbl)}
json.dump(out, open(R + "/gpurun_out/%s_pmc_traffic.json" % TAG, "w"), indent=1)
print(json.dumps({k: v for k, v in kern.items() if k.startswith("k_")}, indent=1))
print("k_search bytes per step (raw / fetch doubled):", int(raw), int(dbl))
PY
