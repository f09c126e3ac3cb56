#!/bin/bash
# Round profile set, run on the GPU box from the repository root:  bash tools/final_profiles.sh <tag>
#   1. rocprofv3 --kernel-trace --stats of the default bench.py (all legs)     -> gpurun_out/<tag>_kernel_stats_bench.csv (+ log)
#   2. the same of `bench.py --only-primary --in-flight 1` (FULL +-64 leg alone, one frame at a time) and of `--only-primary
#      --primary hex --in-flight 1` -> gpurun_out/<tag>_kernel_stats_{full,hex}_leg.csv, <tag>_kernel_trace_{full,hex}_leg.csv,
#      <tag>_union_spans.txt; and of `bench.py --only-primary` (four frames in flight) -> <tag>_kernel_stats_full_leg_in_flight.csv
#   3. HBM traffic counters (FETCH_SIZE / WRITE_SIZE, one pass each) of both legs -> gpurun_out/<tag>_pmc_traffic.json
# Copy the files you want kept from gpurun_out/ into profiles/ (pmc_traffic.json is what bench.py reads).
set -u
TAG=${1:-r03}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
run_stats() {   # name, bench arguments...
    local name=$1; shift
    rm -rf $R/gpurun_out/prof_$name
    timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$name -- python3 $R/bench.py "$@" > $R/gpurun_out/${TAG}_bench_under_rocprof_$name.log 2>&1 || { echo "stats run $name failed"; tail -5 $R/gpurun_out/${TAG}_bench_under_rocprof_$name.log; return 1; }
    local f=$(find $R/gpurun_out/prof_$name -name '*kernel_stats.csv' | head -1)
    cp "$f" $R/gpurun_out/${TAG}_kernel_stats_$name.csv && echo "== $name" && cut -c1-150 $R/gpurun_out/${TAG}_kernel_stats_$name.csv | head -14
    local t=$(find $R/gpurun_out/prof_$name -name '*kernel_trace.csv' | head -1)
    [ -n "$t" ] && cp "$t" $R/gpurun_out/${TAG}_kernel_trace_$name.csv
    return 0
}
run_stats bench --steps 20 --warmup 3 --no-cpu || exit 1
# one frame at a time: a dispatch's duration is then the kernel's own (with frames in flight it shares the GPU)
run_stats full_leg --steps 20 --warmup 3 --only-primary --in-flight 1 || exit 1
run_stats hex_leg --steps 20 --warmup 3 --only-primary --primary hex --in-flight 1 || exit 1
# the headline as bench.py runs it (four frames in flight)
run_stats full_leg_in_flight --steps 40 --warmup 4 --only-primary || exit 1
{
    echo "== FULL +-64 leg: exhaustive kernels (k_pack_fenc + k_full_hier), then the general and group-per-job kernels' launches of one step"
    python3 $R/tools/union_span.py $R/gpurun_out/${TAG}_kernel_trace_full_leg.csv k_pack_fenc,k_full_hier
    python3 $R/tools/union_span.py $R/gpurun_out/${TAG}_kernel_trace_full_leg.csv k_search,k_lane
    echo "== HEX + subme 5 leg: the general kernel's launches and the group-per-job kernel of one step (side by side)"
    python3 $R/tools/union_span.py $R/gpurun_out/${TAG}_kernel_trace_hex_leg.csv k_search,k_lane
    echo "== FULL +-64 leg with four frames in flight: busy span of all search kernels over the dispatches of the run"
    python3 - $R/gpurun_out/${TAG}_kernel_trace_full_leg_in_flight.csv <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if any(k in r["Kernel_Name"] for k in ("k_full_hier", "k_pack_fenc", "k_search", "k_lane"))]
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows)
n_frames = sum(1 for r in rows if "k_full_hier" in r["Kernel_Name"])
# the last 40 frames (the timed region and the five one-at-a-time frames behind it are in the trace too: take the longest gap-free stretch)
busy, cur_s, cur_e, spans = 0, None, None, []
for s, e in iv:
    if cur_e is None or s > cur_e + 20000:
        if cur_e is not None: spans.append((cur_s, cur_e))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
spans.append((cur_s, cur_e))
ls, le = max(spans, key=lambda p: p[1] - p[0])
inside = sum(1 for r in rows if "k_full_hier" in r["Kernel_Name"] and ls <= int(r["Start_Timestamp"]) <= le)
print("%d k_full_hier dispatches in the trace; longest gap-free stretch: %.3f ms with %d of them -> %.4f ms per frame" % (n_frames, (le - ls) / 1e6, inside, (le - ls) / 1e6 / max(inside, 1)))
# the time during which at least one k_full_hier dispatch is running (their union), per frame: the pass's rate while
# several launches share the GPU
hv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if "k_full_hier" in r["Kernel_Name"] and ls <= int(r["Start_Timestamp"]) <= le)
u, cs, ce = 0, None, None
for s, e in hv:
    if ce is None or s > ce:
        if ce is not None: u += ce - cs
        cs, ce = s, e
    else:
        ce = max(ce, e)
u += ce - cs
px = 3.3966937600e10    # executed pixel-SADs of one 1080p frame (the 8x8 cells of every candidate)
print("union of the k_full_hier dispatches in that stretch: %.3f ms = %.4f ms per frame -> %.3f of the packed-SAD peak (157.29 T pixel-SAD/s) while they run" % (u / 1e6, u / 1e6 / max(inside, 1), px / (u / 1e9 / max(inside, 1)) / 157.2864e12))
PY
} > $R/gpurun_out/${TAG}_union_spans.txt 2>&1
cat $R/gpurun_out/${TAG}_union_spans.txt
for leg in full hex; do
    for c in FETCH_SIZE WRITE_SIZE; do
        rm -rf $R/gpurun_out/pmc_${leg}_$c
        timeout -s KILL 200 rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/pmc_${leg}_$c -- python3 $R/bench.py --steps 5 --warmup 1 --only-primary --in-flight 1 --primary $leg > $R/gpurun_out/pmc_${leg}_$c.log 2>&1 || { echo "pass $leg $c failed"; tail -5 $R/gpurun_out/pmc_${leg}_$c.log; exit 1; }
    done
done
python3 - $R $TAG <<'PY'
import csv, sys, glob, json, collections
R, TAG = sys.argv[1], sys.argv[2]
out = {"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over `bench.py --steps 5 --warmup 1 --only-primary "
               "[--primary hex]` (tools/final_profiles.sh); KiB per dispatch as reported (mean over the dispatches of one kernel "
               "and grid size). FETCH_SIZE on gfx950 counts 64 B per 128-B request for wide streaming reads; these kernels read "
               "dwords, a width for which the counter is uncalibrated, so both the raw and the doubled figure are given.",
       "legs": {}}
for leg in ("full", "hex"):
    per = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        f = glob.glob(R + "/gpurun_out/pmc_%s_%s/**/*counter_collection.csv" % (leg, c), recursive=True)[0]
        d = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c:
                d[(r["Kernel_Name"].split("(")[0].replace("void ", "").replace("xa::", ""), r.get("Grid_Size", "?"))].append(float(r["Counter_Value"]))
        per[c] = d
    kern = {}
    for k in sorted(set(per["FETCH_SIZE"]) | set(per["WRITE_SIZE"])):
        if not k[0].startswith("k_"):
            continue
        fv, wv = per["FETCH_SIZE"].get(k, []), per["WRITE_SIZE"].get(k, [])
        kern["%s grid=%s" % k] = {"fetch_KiB": round(sum(fv) / max(len(fv), 1), 1), "write_KiB": round(sum(wv) / max(len(wv), 1), 1),
                                  "dispatches": max(len(fv), len(wv))}
    out["legs"][leg] = kern
    def total(prefixes, mul):
        # bytes of one step: a kernel launched in fewer steps than the others (the large-PU class has two variants, chosen
        # by what runs beside it) counts by its share of the steps
        sel = [v for k, v in kern.items() if k.startswith(prefixes)]
        steps = max([v["dispatches"] for v in sel] + [1])
        return int(sum((mul * v["fetch_KiB"] + v["write_KiB"]) * 1024 * v["dispatches"] / steps for v in sel))
    if leg == "full":
        # one step = one k_pack_fenc + one k_full_hier launch (+ the general kernel's launches, listed above)
        out["k_full_hier_bytes_per_launch"] = total(("k_full_hier", "k_pack_fenc"), 1)
        out["k_full_hier_bytes_per_launch_fetch_doubled"] = total(("k_full_hier", "k_pack_fenc"), 2)
        out["full_leg_k_search_bytes_per_step"] = total(("k_search", "k_lane"), 1)
    else:
        out["k_search_u8_bytes_per_launch"] = total(("k_search", "k_lane"), 1)
        out["k_search_u8_bytes_per_launch_fetch_doubled"] = total(("k_search", "k_lane"), 2)
json.dump(out, open(R + "/gpurun_out/%s_pmc_traffic.json" % TAG, "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k not in ("note",)}, indent=1))
PY
