#!/usr/bin/env python3
"""Motion-estimation benchmark (one JSON line on rank 0).

    python bench.py --gpus N --steps K --warmup W
    (N > 1: one rank per GPU over RCCL; started by torch.distributed.run, or by this script itself
     when it is invoked plainly with --gpus N)

A step = one whole-frame ME pass of the hot path on every rank:
  * rank 0 owns the reconstructed reference plane; with N > 1 it is broadcast to the other
    ranks over RCCL (the path's only exchange step) through a two-buffer ring whose hand-overs are
    stream events (the next broadcast runs beside the current frame's searches), then
  * every rank searches ITS OWN current 1080p frame (frame rank+1 of the synthetic sequence)
    against that reference -- frames are independent, so the work is sharded with no other
    collective (weak scaling: one frame per GPU per step).

Workload of `value` (BASELINE.json's metric "evals/sec and frames/sec, 1080p, +-64 search"; configs[0] on
the GPU): 1920x1080 8-bit, 64x64 CTUs, exhaustive +-64 integer search with the SAD metric for every square
PU of the quadtree (64/32/16/8: 42 900 PUs, 7.0e8 block-cost evaluations per frame), subme 0.
Secondary legs of the same run (N = 1): BASELINE configs[1] (HEX / DIA search + quarter-pel SATD
refinement), the lookahead path, the PCIe-inclusive rate, and the reference's CPU paths on the host cores.

Planes, job lists and result buffers are resident in HBM before the timed region starts.
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from x265_amd import jobs as J  # noqa: E402
from x265_amd import synth  # noqa: E402

W, H = 1920, 1080
QP = 32
MERANGE = 64
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak
SAD_LANES_PER_CLK_PER_CU = 64  # v_sad_u8 issue rate measured with tools/ubench_sad (lane-ops/clk/CU)


def build_job_lists():
    hex_jobs = J.frame_jobs(W, H, sizes=(64, 16), method=J.HEX, subme=5, merange=MERANGE, qp=QP)
    full_jobs = J.frame_jobs(W, H, sizes=(64, 32, 16, 8), method=J.FULL, subme=0, merange=MERANGE, qp=QP)
    return hex_jobs, full_jobs


def pixel_sads(full_jobs):
    return float(sum(J.full_search_evals(full_jobs[full_jobs["w"] == s]) * s * s for s in (64, 32, 16, 8)))


def cpu_baseline(budget_s=6.0):
    """The reference's own CPU paths (oracle/_ref: compiled from the reference's sources) on the host cores,
    on bounded samples of the same job lists (oracle/cpu_baseline_worker.py, one process per core):
    C primitives and the intrinsic (SSE4.1-dispatched) primitives, one core and all cores given to this job."""
    worker = os.path.join(ROOT, "oracle", "cpu_baseline_worker.py")
    have = {k: os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libx265ref%s.so" % k)) for k in ("8", "8v")}
    if not have["8"]:
        return None
    # the job's share of the host: a one-GPU box grants 16 cores of a much larger machine (OMP_NUM_THREADS says how many)
    host_cpus = os.cpu_count() or 1
    cores = max(1, min(host_cpus, int(os.environ.get("X265AMD_BENCH_CORES", os.environ.get("OMP_NUM_THREADS", "16")))))
    env = dict(os.environ, OMP_NUM_THREADS="8")

    def run(kind, leg, procs, seconds, evals=None):
        cmd = [sys.executable, worker, kind, leg, str(seconds)] + ([str(evals)] if evals else [])
        ps = [subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, env=env) for _ in range(procs)]
        return [json.loads(p.communicate(timeout=600)[0].strip().splitlines()[-1]) for p in ps]

    out = {}
    for leg in ("full", "hex"):
        one = run("8", leg, 1, budget_s)[0]
        per_pass = one["evals"] / one["reps"]
        sample = "%d jobs of the %s frame (strided), %d passes, %.1f s" % (one["jobs"], "FULL +-64" if leg == "full" else "HEX+subme5",
                                                                          one["reps"], one["seconds"])
        out[leg] = {"c_1core": {"value": one["evals"] / one["seconds"], "unit": "evals/s", "cores": 1, "kind": "reference",
                                "path": "C primitives (cpuid=1)", "sample": sample,
                                "pixel_sads_per_s": one["pixel_sads"] / one["seconds"] if leg == "full" else None}}
        if have["8v"]:
            v1 = run("8v", leg, 1, budget_s / 2, per_pass)[0]
            assert v1["crc"] == one["crc"], "the reference's C and intrinsic paths disagree"
            out[leg]["vec_1core"] = {"value": v1["evals"] / v1["seconds"], "unit": "evals/s", "cores": 1, "kind": "reference",
                                     "path": "intrinsic primitives, the reference's run-time dispatch (up to SSE4.1)",
                                     "sample": "same sample, %d passes, %.1f s" % (v1["reps"], v1["seconds"]),
                                     "pixel_sads_per_s": v1["pixel_sads"] / v1["seconds"] if leg == "full" else None}
            vn = run("8v", leg, cores, budget_s / 2, per_pass)
            out[leg]["vec_all_cores"] = {"value": sum(o["evals"] for o in vn) / max(o["seconds"] for o in vn), "unit": "evals/s",
                                         "cores": cores, "host_cpus": host_cpus, "kind": "reference",
                                         "path": "intrinsic primitives, one process (one MotionEstimate) per core, each on the same sample; "
                                                 "%d of the machine's %d CPUs (the share this job may use)" % (cores, host_cpus)}
    return out


def sharded_legs(eng, dev, rank, world, args, barrier, dist):
    """BASELINE configs[3] and configs[4] as they shard over N ranks (run at every N, N = 1 included):
      lookahead_batch_4k  8 current 4K frames against the same two references: frame i -> rank i mod N; the references are
                          uploaded on rank 0 and broadcast ONCE per batch (RCCL); every rank searches its frames (HEX +
                          subme 5, 16x16 and 64x64 PUs, both references) - fixed total work, "strong" over the batch;
      sa8d_all_shapes_4k10  4K 10-bit, all 593 PUs of every CTU, exhaustive +-16 SA8D search: one frame per rank against a
                          broadcast reference (18.6 MB) - "weak" like the headline.
    Every rank compares a sample of its own results with the oracle; the line carries the AND over the ranks."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_ffi as F
    from x265_amd import dist as xd
    W4, H4, NF = 3840, 2160, 8
    out = {}
    grouped = dist is not None and dist.is_initialized()
    red_dev = dev if (grouped and dist.get_backend() == "nccl") else "cpu"

    def reduce(v, op):
        if not grouped:
            return float(v)
        t = torch.tensor([float(v)], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=op)
        return float(t.item())

    def bcast(t):
        # one collective per reference plane, ordered against the engine's stream on the GPU (no host wait)
        t0 = time.perf_counter()
        xd.broadcast_plane(t, 0, engine=eng)
        eng.sync()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e6

    oracle = F.Oracle()
    # ---- configs[3]: the 8-frame batch
    seq = synth.make_sequence(W4, H4, NF + 2, seed=99)            # frames 0, 1: the references; 2 .. 9: the batch
    stride = seq[0].shape[1]
    refs = [torch.from_numpy(seq[i]).to(dev) if rank == 0 else torch.zeros(seq[i].shape, dtype=torch.uint8, device=dev) for i in range(2)]
    ref_ids = [eng.wrap_device_plane(t.data_ptr(), W4, H4, stride, 1, 8) for t in refs]
    mine = xd.shard_frames(NF, rank, world)
    encs = {f: torch.from_numpy(seq[2 + f]).to(dev) for f in mine}
    enc_ids = {f: eng.wrap_device_plane(encs[f].data_ptr(), W4, H4, stride, 1, 8) for f in mine}
    jobs = np.concatenate([J.frame_jobs(W4, H4, sizes=(64, 16), method=J.HEX, subme=5, merange=64, qp=32, ref_idx=r) for r in range(2)])
    jls = {f: eng.make_joblist(jobs, n_refs=2) for f in mine}
    times, bus = [], []
    for rep in range(3):
        barrier()
        t0 = time.perf_counter()
        b_us = sum(bcast(t) for t in refs) if grouped else 0.0     # once per batch
        for i, f in enumerate(mine):
            jls[f].run_overlapped(enc_ids[f], ref_ids, i & 1)       # (a rank's frames two in flight: the references are complete here)
        eng.sync()
        barrier()
        times.append(time.perf_counter() - t0)
        bus.append(b_us)
    dt = reduce(min(times), dist.ReduceOp.MAX if grouped else None)
    eng.eval_count(reset=True)
    for f in mine:
        jls[f].run(enc_ids[f], ref_ids, fetch=False)
    evals = reduce(eng.eval_count(), dist.ReduceOp.SUM if grouped else None)
    ok = 1.0
    if mine:
        f = mine[0]
        sel = np.arange(rank, len(jobs), 211)
        exp, _ = oracle.search(seq[2 + f], [seq[0], seq[1]], np.ascontiguousarray(jobs[sel]), threads=8)
        ok = float(np.array_equal(jls[f].results()[sel], exp))
    ok = reduce(ok, dist.ReduceOp.MIN if grouped else None)
    out["lookahead_batch_4k"] = {"workload": "BASELINE configs[3]: 8 current 4K 8-bit frames against the same 2 references, frame i -> rank i mod N, "
                                             "references broadcast from rank 0 once per batch; HEX + subme 5, 16x16 + 64x64 PUs, both references "
                                             "(%d jobs per frame)" % len(jobs),
                                 "scaling": "strong (the batch is fixed)", "frames": NF, "batch_ms": dt * 1e3, "frames_per_s": NF / dt,
                                 "value": evals / dt, "unit": "evals/s", "reference_broadcast_us": reduce(min(bus), dist.ReduceOp.MAX if grouped else None),
                                 "reference_bytes": int(2 * seq[0].nbytes), "parity_sample_ok_all_ranks": bool(ok)}
    for f in mine:
        jls[f].close()
        eng.release_plane(enc_ids[f])
    for i in ref_ids:
        eng.release_plane(i)
    del refs, encs
    # ---- configs[4]: 4K 10-bit, every PU shape, SA8D, one frame per rank
    seq4 = synth.make_sequence(W4, H4, world + 1, seed=5, bit_depth=10)
    # (16-bit planes travel as bytes: the collective library has no 16-bit integer type)
    ref4 = torch.from_numpy(seq4[0].view(np.uint8)).to(dev) if rank == 0 else torch.zeros(seq4[0].view(np.uint8).shape, dtype=torch.uint8, device=dev)
    enc4 = torch.from_numpy(seq4[rank + 1].view(np.uint8)).to(dev)
    rid = eng.wrap_device_plane(ref4.data_ptr(), W4, H4, seq4[0].shape[1], 2, 10)
    eid = eng.wrap_device_plane(enc4.data_ptr(), W4, H4, seq4[0].shape[1], 2, 10)
    shape_jobs = J.all_partition_jobs(W4, H4, method=J.FULL, subme=0, merange=16, qp=27, icost=J.SA8D)
    jl = eng.make_joblist(shape_jobs, bytes_per_px=2)
    b_us = bcast(ref4) if grouped else 0.0
    for _ in range(2):
        jl.run(eid, [rid], fetch=False)
    eng.sync()
    eng.eval_count(reset=True)
    steps = 5
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        if grouped:
            xd.broadcast_plane(ref4, 0, engine=eng)               # a new reference every frame: the exchange step inside the timed region
        jl.run(eid, [rid], fetch=False)
    eng.sync()
    barrier()
    dt = reduce(time.perf_counter() - t0, dist.ReduceOp.MAX if grouped else None)
    evals = reduce(eng.eval_count(), dist.ReduceOp.SUM if grouped else None)
    sel = np.arange(rank, len(shape_jobs), 4801)
    exp, _ = oracle.search(seq4[rank + 1], [seq4[0]], np.ascontiguousarray(shape_jobs[sel]), threads=8, bit_depth=10)
    ok = reduce(float(np.array_equal(jl.results()[sel], exp)), dist.ReduceOp.MIN if grouped else None)
    out["sa8d_all_shapes_4k10"] = {"workload": "BASELINE configs[4]: 4K 10-bit, all 593 PUs of every CTU (24 shapes), exhaustive +-16 SA8D search; one frame "
                                               "per rank per step against a reference broadcast from rank 0 every step (%d jobs per frame)" % len(shape_jobs),
                                   "scaling": "weak", "ms_per_step": dt / steps * 1e3, "frames_per_s": world * steps / dt, "value": evals / dt,
                                   "unit": "evals/s", "reference_broadcast_us": reduce(b_us, dist.ReduceOp.MAX if grouped else None),
                                   "reference_bytes": int(seq4[0].nbytes), "parity_sample_ok_all_ranks": bool(ok)}
    jl.close()
    eng.release_plane(rid)
    eng.release_plane(eid)
    return out


class stdout_to_stderr:
    """The collective library prints a version banner on the process's stdout when it initialises; this script's stdout is ONE
    JSON line, so file descriptor 1 points at stderr while a process group comes up and runs its first collective."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)
        return False


def trace(msg):
    if os.environ.get("X265AMD_BENCH_TRACE"):
        print("[bench rank %s] %s" % (os.environ.get("RANK", "0"), msg), file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--only-primary", action="store_true",
                    help="run only the leg `value` is quoted on (for profiler runs: every launch of the search "
                         "kernels then belongs to that leg) and print a reduced line")
    ap.add_argument("--in-flight", type=int, default=4, choices=(1, 2, 3, 4),
                    help="frames in flight on the GPU (x265amd_joblist_run_overlapped: one job list and one stream set per frame "
                         "in flight; the pattern-search legs use at most 2); 1 = one frame at a time")
    ap.add_argument("--primary", choices=("full", "hex"), default="full",
                    help="with --only-primary: which leg to run alone (profiling aid; `value` is always the FULL leg otherwise)")
    args = ap.parse_args()

    # ---- N > 1 without a launcher: start one rank per GPU ourselves (before anything touches the GPU)
    env_world = os.environ.get("WORLD_SIZE")
    if args.gpus > 1 and env_world is None:
        port = 29400 + (os.getpid() % 500)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))
    if env_world is not None and int(env_world) != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%s ranks" % (args.gpus, env_world))

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP kernels are the only compute path")
    # (rehearsals of the N > 1 path on a one-GPU box: X265AMD_DIST_BACKEND=gloo puts every rank on that GPU)
    backend = os.environ.get("X265AMD_DIST_BACKEND", "nccl")
    local_rank = local_rank % torch.cuda.device_count() if backend != "nccl" else local_rank
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        with stdout_to_stderr():
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=dev)
            else:
                dist.init_process_group(backend)
            warm = torch.zeros(1, device=dev if backend == "nccl" else "cpu")
            dist.all_reduce(warm)                   # the communicator comes up here
            torch.cuda.synchronize()
    group_backend = backend if world > 1 else None

    if os.environ.get("X265AMD_BENCH_TRACE"):       # a stuck rank shows where it is
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ.get("X265AMD_BENCH_TRACE_AFTER", "90")), exit=False)
    trace("process group up")
    from x265_amd.api import Engine, build_info
    from x265_amd import build as xbuild
    eng = Engine(local_rank)
    trace("engine created")
    # which sources the loaded binary was built from, and whether the tree beside it still holds those sources
    library_id = {"build_info": build_info(), "tree_source_hash": xbuild.source_hash(),
                  "binary_matches_tree": build_info().endswith(xbuild.source_hash())}

    # ---- inputs, resident in HBM ---------------------------------------------------------
    base = synth.base_frame(W, H)
    ref_np = synth.pad_plane(base)                                  # frame 0 = reference (owned by rank 0)
    enc_np = synth.pad_plane(synth.moved_frame(base, rank + 1))     # this rank's current frame
    stride = ref_np.shape[1]
    enc_t = torch.from_numpy(enc_np).to(dev)
    # two reference buffers: the broadcast for the next step overlaps the current step's search
    ref_bufs = [torch.from_numpy(ref_np).to(dev) if rank == 0 else torch.zeros(ref_np.shape, dtype=torch.uint8, device=dev)
                for _ in range(2)]
    ref_t = ref_bufs[0]
    enc_id = eng.wrap_device_plane(enc_t.data_ptr(), W, H, stride, 1, 8)
    ref_ids = [eng.wrap_device_plane(t.data_ptr(), W, H, stride, 1, 8) for t in ref_bufs]
    ref_id = ref_ids[0]

    hex_jobs, full_jobs = build_job_lists()
    # planned job lists: jobs, work-item plan and result buffers live in device memory
    jl_hex = eng.make_joblist(hex_jobs)
    jl_full = eng.make_joblist(full_jobs)
    # frames in flight: one list (jobs, plan, result buffers) and one stream set per frame.  The exhaustive frames take the
    # single-stream sets 4 .. 7, the pattern-search frames the two two-stream sets 0 / 1 (measured: the best of each kind)
    depth_full = max(1, args.in_flight)
    depth_hex = min(2, depth_full)
    full_lists = [jl_full] + [eng.make_joblist(full_jobs) for _ in range(depth_full - 1)]
    hex_lists = [jl_hex] + [eng.make_joblist(hex_jobs) for _ in range(depth_hex - 1)]
    FULL_SETS, HEX_SETS = (4, 5, 6, 7)[:depth_full], (0, 1)[:depth_hex]
    eng.set_timing(True)
    trace("job lists planned")
    from x265_amd import dist as xd

    ring = xd.PlaneRing(ref_bufs, engine=eng, src=0, overlap=bool(os.environ.get("X265AMD_OVERLAP")))
    if os.environ.get("X265AMD_BENCH_TRACE"):       # a watchdog: what is busy when a rank stops making progress
        import threading

        def watchdog():
            for _ in range(3):
                time.sleep(int(os.environ.get("X265AMD_BENCH_WATCH_S", "25")))
                cur = torch.cuda.current_stream(dev)
                ev = lambda e: "-" if e is None else ("done" if e.query() else "BUSY")
                trace("watchdog: engine {%s}; torch current stream %s; ring free %s filled %s ready %d"
                      % (eng.debug_state(), "done" if cur.query() else "BUSY", [ev(e) for e in ring.free],
                         [ev(e) for e in ring.filled], ring.ready))
        threading.Thread(target=watchdog, daemon=True).start()

    step_no = [0]
    host_queue = [0.0]

    def step(lists, sets):
        # reference plane: rank 0 -> all (RCCL over xGMI when N > 1), then this rank's frame; no host waits.  Frames in
        # flight: step k takes list k mod depth on that list's stream set, so consecutive frames run side by side
        k = step_no[0]
        step_no[0] += 1
        i = k % len(lists)
        b = ring.acquire(k, sets[i] if len(lists) > 1 else 0)
        trace("step %d: reference in buffer %d" % (k, b))
        if len(lists) > 1:
            lists[i].run_overlapped(enc_id, [ref_ids[b]], sets[i])
        else:
            lists[0].run(enc_id, [ref_ids[b]], fetch=False)
        ring.prefetch(k + 1)
        ring.release(k, sets[i] if len(lists) > 1 else 0)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(lists, warmup, steps, sets=(0,)):
        """K steps queued back to back (no host wait inside the timed region), `len(lists)` frames in flight; then the
        kernel times of frames run one at a time (HIP events around one frame's kernels mean something only then)."""
        if not isinstance(lists, (list, tuple)):
            lists = [lists]
        for _ in range(max(warmup, len(lists))):
            step(lists, sets)
        trace("warm-up queued")
        eng.sync()
        eng.eval_count(reset=True)
        barrier()
        trace("timed region starts")
        eng.set_timing(False)
        t0 = time.perf_counter()
        for k in range(steps):
            step(lists, sets)
        host_queue[0] = (time.perf_counter() - t0) / steps * 1e6      # the host's time to queue a step (no GPU wait in it)
        eng.sync()
        barrier()
        dt = time.perf_counter() - t0
        trace("timed region done")
        evals = eng.eval_count()
        # HIP events on the engine's streams around the kernels of ONE frame on an otherwise idle GPU (timestamp events drain
        # the queue between launches, so they stay out of the timed region): median of five frames
        eng.set_timing(True)
        kt = []
        for _ in range(5):
            lists[0].run(enc_id, [ref_ids[0]], fetch=False)
            eng.sync()
            kt.append(eng.kernel_times())
        # (timing stays on: the legs that time single calls read it)
        k_full = float(np.median([a for a, _ in kt]))
        k_search = float(np.median([b for _, b in kt]))
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
            e = torch.tensor([evals], dtype=torch.float64, device=dev)
            dist.all_reduce(e, op=dist.ReduceOp.SUM)
            evals = float(e.item())
        return dt, evals, k_full, k_search

    def pmc(key):
        f = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        try:
            return json.load(open(f)).get(key)
        except Exception:
            return None

    TRAFFIC_SOURCE = ("profiles/pmc_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (tools/final_profiles.sh) over this "
                      "command on the committed state; read from that file, not measured in this run")

    prop = torch.cuda.get_device_properties(local_rank)
    cus = prop.multi_processor_count
    clk_hz = prop.clock_rate * 1e3 if hasattr(prop, "clock_rate") else 2.4e9
    sad_peak = cus * SAD_LANES_PER_CLK_PER_CU * 4 * clk_hz
    alg_bytes = 2 * ref_np.nbytes                 # enc plane + ref plane read once (jobs / results added per leg)

    # ---- primary: FULL +-64 SAD, every square PU of the quadtree ---------------------------------
    run_full = not (args.only_primary and args.primary == "hex")
    run_hex = (world == 1) and not (args.only_primary and args.primary == "full")
    full = hexr = None
    if run_full:
        fdt, fevals, k_full_ms, k_tail_ms = timed(full_lists, args.warmup, args.steps, FULL_SETS)
        full_host_queue_us = host_queue[0]
        full_res = jl_full.run(enc_id, [ref_id], fetch=True)
        px_alg = pixel_sads(full_jobs)                    # 1 evaluation of a w x h block = w*h pixel-SADs, no credit for reuse
        px_exec = float(J.full_search_evals(full_jobs[full_jobs["w"] == 8]) * 64)      # measured: the 8x8 cells only
        full_bytes = alg_bytes + full_jobs.nbytes + len(full_jobs) * 8
        full = {
            "value": fevals / fdt, "ms_per_step": fdt / args.steps * 1e3, "frames_per_s": world * args.steps / fdt,
            "evals_per_frame": fevals / (args.steps * world), "ms_per_frame_one_at_a_time": k_full_ms + k_tail_ms,
            "roofline": {"bound": "valu", "kernel": "k_full_hier", "achieved": px_exec / (k_full_ms * 1e-3) / 1e12, "peak": sad_peak / 1e12,
                         "unit": "T pixel-SAD/s", "frac": px_exec / (k_full_ms * 1e-3) / sad_peak,
                         "algorithmic_achieved": px_alg / (k_full_ms * 1e-3) / 1e12,
                         "algorithmic_frac": px_alg / (k_full_ms * 1e-3) / sad_peak,
                         "traffic": pmc("k_full_hier_bytes_per_launch"), "traffic_source": TRAFFIC_SOURCE,
                         "kernel_ms": k_full_ms, "search_kernel_ms": k_tail_ms,
                         "kernel_ms_is": "HIP-event bracket on the engine's stream around the exhaustive kernels (k_pack_fenc + k_full_hier) "
                                         "of ONE frame on an otherwise idle GPU (median of five frames run one at a time right after the "
                                         "timed region; with several frames in flight a frame's kernels share the GPU with its neighbours' "
                                         "and a bracket around them measures the mix); search_kernel_ms: the same around the search kernels "
                                         "(predictors, slab combine, half-pel SAD round, final SATD)",
                         "frames_in_flight_frac": px_exec / (fdt / args.steps) / sad_peak,
                         "frames_in_flight_frac_is": "executed pixel-SADs of a frame over ms_per_step (all of a frame's kernels, search "
                                                     "kernels included, %d frames in flight) over the peak: the packed-SAD pipe's utilisation "
                                                     "over the whole timed region" % depth_full,
                         "algorithmic_pixel_sads_per_launch": px_alg,
                         "executed_pixel_sads_per_launch": px_exec,
                         "note": "achieved / frac count the pixel-SADs the kernel EXECUTES (the 8x8 cells of every candidate, 3.4e10 per frame): "
                                 "the utilisation of the packed-SAD pipe, at most 1. algorithmic_* count every PU's candidates x its own pixels "
                                 "(1.35e11 per frame, the survey's definition of work): the kernel sums the cells up the quadtree instead of "
                                 "measuring each level, so that ratio exceeds 1",
                         "peak_def": "%d CUs x %d lanes/clk x 4 px (v_sad_u8 / v_qsad_pk_u16_u8) x %.2f GHz" % (cus, SAD_LANES_PER_CLK_PER_CU, clk_hz / 1e9)},
            "roofline_hbm": {"bound": "hbm", "kernel": "k_full_hier", "achieved": full_bytes / (k_full_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                             "unit": "GB/s", "frac": full_bytes / (k_full_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                             "algorithmic_bytes_per_launch": full_bytes, "traffic": pmc("k_full_hier_bytes_per_launch"),
                             "traffic_source": TRAFFIC_SOURCE},
        }
    if run_hex:
        hdt, hevals, _, k_hex_ms = timed(hex_lists, args.warmup, args.steps, HEX_SETS)
        hdt1, _, _, _ = timed(jl_hex, 2, args.steps) if depth_hex > 1 else (hdt, 0, 0, 0)
        hex_res = jl_hex.run(enc_id, [ref_id], fetch=True)
        hex_bytes = alg_bytes + hex_jobs.nbytes + len(hex_jobs) * 8
        # algorithmic operations (1 SAD evaluation of a w x h block = w*h, 1 SATD evaluation = 5*w*h; the split
        # per PU size comes from the evaluation counter of one run per size)
        alg_ops = 0.0
        for size in (16, 64):
            js = hex_jobs[hex_jobs["w"] == size]
            jl_s = eng.make_joblist(js)
            eng.eval_count(reset=True)
            jl_s.run(enc_id, [ref_id], fetch=False)
            ev = eng.eval_count(reset=True)
            jl_s.close()
            satd = 17 * len(js)
            alg_ops += ((ev - satd) + 5 * satd) * size * size
        eng.set_timing(True)
        hexr = {
            "workload": "1080p 8-bit, HEX search merange 64 + subme 5 (8 half-pel + 8 quarter-pel SATD), PUs 16x16 + 64x64 "
                        "(BASELINE configs[1])",
            "value": hevals / hdt, "unit": "evals/s", "ms_per_frame": hdt / args.steps * 1e3, "frames_per_s": args.steps / hdt,
            "frames_in_flight": depth_hex, "ms_per_frame_one_at_a_time": hdt1 / args.steps * 1e3,
            "jobs_per_frame": int(len(hex_jobs)), "evals_per_frame": hevals / args.steps,
            "satd_evals_per_frame": 17 * int(len(hex_jobs)), "sad_evals_per_frame": hevals / args.steps - 17 * int(len(hex_jobs)),
            # a search is a chain of dependent evaluation calls: instruction issue and per-call latency bind it,
            # so it is priced against the un-packed 32-bit integer-ALU roof (CUs x 64 lanes x clock) and against HBM
            "roofline": {"bound": "valu", "kernel": "k_search<uint8_t>", "achieved": alg_ops / (k_hex_ms * 1e-3) / 1e12,
                         "peak": cus * 64 * clk_hz / 1e12, "unit": "T op/s", "frac": alg_ops / (k_hex_ms * 1e-3) / (cus * 64 * clk_hz),
                         "kernel_ms": k_hex_ms, "algorithmic_ops_per_launch": alg_ops,
                         "frames_in_flight_frac": alg_ops / (hdt / args.steps) / (cus * 64 * clk_hz),
                         "ops_def": "SAD evaluation = w*h, SATD evaluation = 5*w*h integer operations"},
            "roofline_hbm": {"bound": "hbm", "kernel": "k_search<uint8_t>", "achieved": hex_bytes / (k_hex_ms * 1e-3) / 1e9,
                             "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hex_bytes / (k_hex_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                             "algorithmic_bytes_per_launch": hex_bytes, "traffic": pmc("k_search_u8_bytes_per_launch"),
                             "traffic_source": TRAFFIC_SOURCE},
        }

    # ---- the exchange step alone (N > 1): reference plane from rank 0 to every rank ----------------
    bcast_us = None
    if world > 1 or (group_backend and not str(group_backend).startswith("none")):
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            xd.broadcast_plane(ref_bufs[0], 0, engine=eng)
        eng.sync()
        barrier()
        t = torch.tensor([(time.perf_counter() - t0) / args.steps * 1e6], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        bcast_us = float(t.item())

    if args.only_primary and args.primary == "hex":
        if rank == 0:
            print(json.dumps({"metric": "ME block-cost evals/sec", "unit": "evals/s", "n_gpus": world, "steps": args.steps,
                              "warmup": args.warmup, "note": "HEX leg alone (--only-primary --primary hex: profiling aid)", **hexr}), flush=True)
        return

    line = {
        "metric": "ME block-cost evals/sec",
        "value": full["value"],
        "unit": "evals/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": full["ms_per_step"],
        "frames_per_s": full["frames_per_s"],
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u8",
        "data": "synthetic",
        "config": {"workload": "1080p synthetic 8-bit, 64x64 CTUs, exhaustive +-64 integer search (SAD) of every square PU 64/32/16/8, "
                               "subme 0 (BASELINE configs[0] on the GPU); one frame per GPU per step, %d frames in flight per GPU (the searches of "
                               "consecutive frames are independent, as on the reference's frame threads); reference plane broadcast from "
                               "rank 0" % depth_full,
                   "frames_in_flight": depth_full, "ms_per_frame_one_at_a_time": full["ms_per_frame_one_at_a_time"],
                   "jobs_per_frame": int(len(full_jobs)), "evals_per_frame": full["evals_per_frame"],
                   "qp": QP, "merange": MERANGE, "parallelism": "frames sharded over %d GPU(s)" % world,
                   # the collective alone, host-bracketed (it is also inside every timed step, beside the searches)
                   "reference_broadcast_us": bcast_us, "reference_plane_bytes": int(ref_np.nbytes),
                   "process_group": group_backend,
                   # the host's share of a step: ring acquire / prefetch / release (collective + events) and the list's launches
                   "host_queue_us_per_step": full_host_queue_us,
                   "host_queue_frac_of_step": full_host_queue_us / (full["ms_per_step"] * 1e3)},
        "roofline": full["roofline"],
        "roofline_hbm": full["roofline_hbm"],
        "cpu_baseline": None,
        "library": library_id,
    }

    if world == 1 and not args.only_primary and not os.environ.get("X265AMD_NO_GROUP"):
        # A group of ONE rank, so that the collectives of the N > 1 path run at N = 1 too: the headline step again with the
        # reference broadcast in every step (what the host pays to queue a step with the collective on), then the sharded legs.
        try:
            with stdout_to_stderr():
                group_backend = xd.single_rank_group(dev)
                warm = torch.zeros(1, device=dev if group_backend == "nccl" else "cpu")
                dist.all_reduce(warm)               # the communicator comes up here
                torch.cuda.synchronize()
        except Exception as e:      # the legs below then run without collectives and say so
            group_backend = "none (%s)" % type(e).__name__
        if dist.is_initialized():
            ring = xd.PlaneRing(ref_bufs, engine=eng, src=0, overlap=bool(os.environ.get("X265AMD_OVERLAP")))
            step_no[0] = 0
            cdt, _, _, _ = timed(full_lists, 3, 20, FULL_SETS)
            line["config"]["with_one_rank_group"] = {"process_group": group_backend, "ms_per_step": cdt / 20 * 1e3,
                                                     "host_queue_us_per_step": host_queue[0],
                                                     "host_queue_frac_of_step": host_queue[0] / (cdt / 20 * 1e6),
                                                     "note": "the headline step with a reference broadcast in it (one-rank RCCL group): ring acquire / "
                                                             "release, the collective, the list's launches; host_queue = the host's time to queue a step"}
            # (the single-GPU legs below run without a collective in their steps, as `value` did)
            ring = xd.PlaneRing(ref_bufs, engine=eng, src=0, collective=False)
            step_no[0] = 0
    if not args.only_primary:
        try:
            line["sharded"] = sharded_legs(eng, dev, rank, world, args, barrier, dist if (world > 1 or dist.is_initialized()) else None)
        except Exception as e:      # (the headline above stands on its own; a failure here is reported, not hidden)
            import traceback
            line["sharded"] = {"error": "%s: %s" % (type(e).__name__, e), "traceback": traceback.format_exc()[-1500:]}

    # With N > 1 the legs above run (the one `value` is quoted on, the broadcast alone, the two sharded 4K legs): the
    # other secondary legs are single-GPU figures.
    if args.only_primary or world > 1:
        if rank == 0:
            line["note"] = "primary leg only (--only-primary, or N > 1: cpu_baseline and the secondary legs are reported at N = 1)"
            print(json.dumps(line), flush=True)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    line["hex_search"] = hexr

    # ---- the same frame with the diamond search (the other method BASELINE configs[1] names) ----
    dia_jobs = hex_jobs.copy()
    dia_jobs["method"] = J.DIA
    dia_lists = [eng.make_joblist(dia_jobs) for _ in range(depth_hex)]
    ddt, devals, _, k_dia_ms = timed(dia_lists, 2, args.steps, HEX_SETS)
    for j in dia_lists:
        j.close()
    line["dia_search"] = {"workload": "same frame and PUs, DIA search + subme 5", "value": devals / ddt, "unit": "evals/s",
                          "ms_per_frame": ddt / args.steps * 1e3, "frames_in_flight": depth_hex, "kernel_ms": k_dia_ms}

    # ---- PCIe-inclusive: both planes uploaded and the MV/cost field downloaded every frame -------
    enc_pin = torch.from_numpy(enc_np).pin_memory()
    ref_pin = torch.from_numpy(ref_np).pin_memory()

    def pcie_step(jl):
        enc_t.copy_(enc_pin, non_blocking=True)
        ref_t.copy_(ref_pin, non_blocking=True)
        eng.torch_stream().wait_stream(torch.cuda.current_stream(dev))
        return jl.run(enc_id, [ref_id], fetch=True)        # blocks until the results are on the host

    pcie = {}
    for name, jl, ev in (("full_search", jl_full, full["evals_per_frame"]), ("hex_search", jl_hex, hexr["evals_per_frame"])):
        for _ in range(2):
            pcie_step(jl)
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            pcie_step(jl)
        barrier()
        d = time.perf_counter() - t0
        pcie[name] = {"ms_per_frame": d / args.steps * 1e3, "frames_per_s": args.steps / d, "value": ev * args.steps / d, "unit": "evals/s"}
    line["pcie_inclusive"] = {"workload": "enc and ref planes copied host->device and the MV/cost field device->host every frame, "
                                          "frames not overlapped", **pcie}

    # The same with the frames pipelined (the wall-clock figure that includes the host-to-device and device-to-host copies): two sets of planes; the planes of frame
    # k + 1 are refilled through the library (x265amd_plane_refill_async: behind the searches queued so far, beside the search of
    # frame k) and the field of frame k comes down behind its search (x265amd_joblist_results_async into page-locked memory);
    # the host waits once at the end
    PD, PP, PA = (2, 4, 2) if depth_full > 1 else (1, 2, 1)     # frames in flight, plane slots, frames a refill runs ahead
    sets = [{"enc_id": eng.upload_plane(enc_np), "ref_id": eng.upload_plane(ref_np)} for _ in range(PP)]
    piped = {}
    for name, jobs_np, ev, psets in (("full_search", full_jobs, full["evals_per_frame"], (4, 5)), ("hex_search", hex_jobs, hexr["evals_per_frame"], (0, 1))):
        jls = [eng.make_joblist(jobs_np) for _ in range(PD)]
        outs = [torch.zeros(len(jobs_np) * 8, dtype=torch.uint8).pin_memory() for _ in range(PD)]

        def frame(k):
            nxt = sets[(k + PA) % PP]
            eng.refill_plane_async(nxt["enc_id"], enc_pin.data_ptr())
            eng.refill_plane_async(nxt["ref_id"], ref_pin.data_ptr())
            st = sets[k % PP]
            if PD > 1:
                jls[k % PD].run_overlapped(st["enc_id"], [st["ref_id"]], psets[k % PD])
            else:
                jls[0].run(st["enc_id"], [st["ref_id"]], fetch=False)
            jls[k % PD].results_async(outs[k % PD].data_ptr())

        for k in range(2 * PP):
            frame(k)
        eng.sync()
        barrier()
        t0 = time.perf_counter()
        for k in range(args.steps):
            frame(k)
        eng.sync()
        barrier()
        d = time.perf_counter() - t0
        same = all(bool(np.array_equal(o.numpy().view(J.RESULT_DTYPE), full_res if name == "full_search" else hex_res)) for o in outs)
        piped[name] = {"ms_per_frame": d / args.steps * 1e3, "frames_per_s": args.steps / d, "value": ev * args.steps / d, "unit": "evals/s",
                       "field_equals_resident_run": same}
        for j in jls:
            j.close()
    for st in sets:
        eng.release_plane(st["enc_id"])
        eng.release_plane(st["ref_id"])
    line["pcie_inclusive"]["pipelined"] = {"workload": "the same traffic with the frames pipelined: %d frames in flight on their own stream sets, %d plane "
                                                       "slots, the planes of frame k + %d uploaded beside the searches of frames k and k + 1 "
                                                       "(x265amd_plane_refill_async: behind the readers of its slot on every set), the field of "
                                                       "frame k downloaded on its own set behind its search into page-locked memory "
                                                       "(x265amd_joblist_results_async); the host waits once at the end" % (PD, PP, PA),
                                           "frames_in_flight": PD,
                                           **piped, "vs_resident": piped["full_search"]["ms_per_frame"] / full["ms_per_step"],
                                           "vs_resident_one_at_a_time": piped["full_search"]["ms_per_frame"] / full["ms_per_frame_one_at_a_time"],
                                           "note": "`value` has %d resident frames in flight" % depth_full}
    ok_pipe = all(v["field_equals_resident_run"] for v in piped.values())

    # ---- lookahead path (not part of `value`): half-resolution plane sets and whole-frame
    # neighbour-predicted searches with the reference's lookahead settings (HEX, subme 1, merange 16, qp 12)
    LOOK_PAIRS = 16
    lr_enc, lr_ref = eng.lowres_create(enc_id), eng.lowres_create(ref_id)
    eng.sync()
    t0 = time.perf_counter()
    for _ in range(50):
        eng.lowres_update(lr_enc, enc_id)
    eng.sync()
    lowres_us = (time.perf_counter() - t0) / 50 * 1e6
    look = {}
    for n in (1, LOOK_PAIRS):
        for _ in range(2):
            eng.lookahead_me([lr_enc] * n, [lr_ref] * n, fetch=False)
        eng.eval_count(reset=True)
        eng.lookahead_me([lr_enc] * n, [lr_ref] * n, fetch=False)
        look[n] = (eng.kernel_times()[1], eng.eval_count())
    look_field = eng.lookahead_me([lr_enc], [lr_ref])[0]
    _, _, _, look_cw, look_ch = eng.lowres_geometry(lr_enc)
    line["lookahead"] = {"workload": "1080p 8-bit -> 960x544 half-resolution sets; whole-frame searches of %d x %d 8x8 CUs with the "
                                     "neighbour-predicted MV chain, HEX + subme 1, merange 16, qp 12 (the reference's lookahead settings)"
                                     % (look_cw, look_ch),
                         "pairs_per_call": LOOK_PAIRS, "kernel_ms_per_call": look[LOOK_PAIRS][0],
                         "frame_pairs_per_s": LOOK_PAIRS / (look[LOOK_PAIRS][0] * 1e-3),
                         "value": look[LOOK_PAIRS][1] / (look[LOOK_PAIRS][0] * 1e-3), "unit": "evals/s",
                         "single_pair_kernel_ms": look[1][0], "lowres_init_us_per_picture": lowres_us}

    # ---- the HEX leg again with its launches ordered by the lookahead's field of THIS frame pair (computed above from the
    # half-resolution planes, as an encoder's lookahead does before the frame is encoded): the jobs the field predicts to be
    # long start first.  Not `value`, and not the order learnt from a previous run of the same frame.
    jl_hex.order_by_field(look_field, look_cw, look_ch)
    odt, _, _, k_ord_ms = timed(jl_hex, 2, args.steps)
    ord_res = jl_hex.run(enc_id, [ref_id], fetch=True)
    line["hex_search"]["lookahead_ordered"] = {"ms_per_frame": odt / args.steps * 1e3, "kernel_ms": k_ord_ms, "frames_in_flight": 1,
                                               "compare_with": "hex_search.ms_per_frame_one_at_a_time",
                                               "results_equal": bool(np.array_equal(ord_res, hex_res)),
                                               "note": "launch order from x265amd_lookahead_me's half-resolution field of the same frame pair "
                                                       "(x265amd_joblist_order_by_field); the lookahead's own time is in `lookahead`"}

    # ---- 4K 10-bit, Hadamard integer stage (BASELINE configs[4]'s metric): exhaustive +-16 SA8D search of every
    # square PU 64/32/16/8 in one list; the candidate-per-lane kernel prices the 8x8 cells once for all four levels
    W4, H4 = 3840, 2160
    seq4 = synth.make_sequence(W4, H4, 2, seed=5, bit_depth=10)
    ids4 = [eng.upload_plane(f, bit_depth=10) for f in seq4]
    had_jobs = J.frame_jobs(W4, H4, sizes=(64, 32, 16, 8), method=J.FULL, subme=0, merange=16, qp=27, icost=J.SA8D)
    jl_had = eng.make_joblist(had_jobs, bytes_per_px=2)
    for _ in range(2):
        jl_had.run(ids4[1], [ids4[0]], fetch=False)
    eng.sync()
    eng.eval_count(reset=True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        jl_had.run(ids4[1], [ids4[0]], fetch=False)
    eng.sync()
    had_dt = (time.perf_counter() - t0) / args.steps
    had_evals = eng.eval_count() / args.steps
    k_had_ms, k_had_tail_ms = eng.kernel_times()
    had_res = jl_had.results()
    jl_had.close()
    had_ops = float(sum(7.0 * s * s * J.full_search_evals(had_jobs[had_jobs["w"] == s]) for s in (64, 32, 16, 8)))
    had_exec = float(7.0 * 64 * J.full_search_evals(had_jobs[had_jobs["w"] == 8]))
    line["sa8d_full_4k10"] = {"workload": "4K 10-bit synthetic, exhaustive +-16 integer search with the SA8D metric of every square PU "
                                          "64/32/16/8, subme 0 (BASELINE configs[4]'s metric and pixel type)",
                              "jobs_per_frame": int(len(had_jobs)), "evals_per_frame": had_evals, "value": had_evals / had_dt,
                              "unit": "evals/s", "ms_per_frame": had_dt * 1e3,
                              "roofline": {"bound": "valu", "kernel": "k_full_had<uint16_t>", "achieved": had_exec / (k_had_ms * 1e-3) / 1e12,
                                           "peak": cus * 64 * clk_hz / 1e12, "unit": "T op/s",
                                           "frac": had_exec / (k_had_ms * 1e-3) / (cus * 64 * clk_hz),
                                           "algorithmic_achieved": had_ops / (k_had_ms * 1e-3) / 1e12,
                                           "algorithmic_frac": had_ops / (k_had_ms * 1e-3) / (cus * 64 * clk_hz), "kernel_ms": k_had_ms,
                                           "search_kernel_ms": k_had_tail_ms, "algorithmic_ops_per_launch": had_ops,
                                           "executed_ops_per_launch": had_exec,
                                           "ops_def": "SA8D evaluation = 7*w*h integer operations. frac / achieved: the operations EXECUTED (each 8x8 "
                                                      "cell of a candidate priced once, 7*64 per cell); algorithmic_*: every PU's candidates x its own "
                                                      "pixels (the four levels share the cells, so that ratio exceeds 1)"}}
    # the same frame with EVERY partition of the CTUs that is made of 8x8 cells (85 squares + 124 symmetric / asymmetric
    # rectangles per CTU) in one list: one pass over each window prices them all
    parts = [had_jobs]
    for n2 in (16, 32, 64):
        q4 = n2 // 4
        shapes = [(0, 0, n2, n2 // 2), (0, n2 // 2, n2, n2 // 2), (0, 0, n2 // 2, n2), (n2 // 2, 0, n2 // 2, n2)]
        if n2 > 16:
            shapes += [(0, 0, n2, q4), (0, q4, n2, n2 - q4), (0, 0, n2, n2 - q4), (0, n2 - q4, n2, q4),
                       (0, 0, q4, n2), (q4, 0, n2 - q4, n2), (0, 0, n2 - q4, n2), (n2 - q4, 0, q4, n2)]
        cy4, cx4 = np.mgrid[0:H4 - n2 + 1:n2, 0:W4 - n2 + 1:n2]
        for (ox, oy, w, h) in shapes:
            parts.append(J.make_jobs(W4, H4, cx4.ravel() + ox, cy4.ravel() + oy, w, h, method=J.FULL, subme=0, merange=16, qp=27,
                                     icost=J.SA8D))
    all_jobs = np.concatenate(parts)

    def run_4k10(jobs, steps):
        jl = eng.make_joblist(jobs, bytes_per_px=2)
        for _ in range(2):
            jl.run(ids4[1], [ids4[0]], fetch=False)
        eng.sync()
        eng.eval_count(reset=True)
        t0 = time.perf_counter()
        for _ in range(steps):
            jl.run(ids4[1], [ids4[0]], fetch=False)
        eng.sync()
        dt = (time.perf_counter() - t0) / steps
        ev = eng.eval_count() / steps
        k_ms, tail_ms = eng.kernel_times()
        res = jl.results()
        # two frames in flight (a second list, stream sets 0 / 1)
        jl2 = eng.make_joblist(jobs, bytes_per_px=2)
        pair = (jl, jl2)
        for k in range(2):
            pair[k].run_overlapped(ids4[1], [ids4[0]], k)
        eng.sync()
        t0 = time.perf_counter()
        for k in range(steps + (steps & 1)):
            pair[k & 1].run_overlapped(ids4[1], [ids4[0]], k & 1)
        eng.sync()
        dt2 = (time.perf_counter() - t0) / (steps + (steps & 1))
        same = bool(np.array_equal(jl2.results(), res))
        jl.close()
        jl2.close()
        return res, {"jobs_per_frame": int(len(jobs)), "evals_per_frame": ev, "value": ev / dt, "unit": "evals/s",
                     "ms_per_frame": dt * 1e3, "kernel_ms": k_ms, "search_kernel_ms": tail_ms,
                     "two_in_flight": {"ms_per_frame": dt2 * 1e3, "value": ev / dt2, "unit": "evals/s", "results_equal": same}}

    all_res, line["sa8d_full_4k10"]["all_partitions"] = run_4k10(all_jobs, 5)
    line["sa8d_full_4k10"]["all_partitions"]["workload"] = "the 209 partitions of every CTU that are made of 8x8 cells (85 squares + 124 rectangles)"
    # BASELINE configs[4]'s full shape list: all 593 PUs of every CTU, 24 shapes from 64x64 down to 8x4 / 4x8 (those with
    # 4- or 12-pixel sides are priced by sa8d_inter's SATD recipe, from the 4x4 quadrant sums of the same pass)
    shape_jobs = J.all_partition_jobs(W4, H4, method=J.FULL, subme=0, merange=16, qp=27, icost=J.SA8D)
    shape_res, line["sa8d_full_4k10"]["all_shapes"] = run_4k10(shape_jobs, 5)
    line["sa8d_full_4k10"]["all_shapes"]["workload"] = "all 593 PUs of every 64x64 CTU (24 HEVC PU shapes, 64x64 .. 8x4 / 4x8), FULL +-16, SA8D"
    for i in ids4:
        eng.release_plane(i)

    if not args.no_cpu:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_ffi as F
        if F.have_reference(8):
            # the reference's C path on one host core: plane construction is its primitive + numpy borders
            # (not timed), the search is its MotionEstimate driven in the lookahead's order
            r8 = F.Reference(8)
            g_lr, e4 = r8.lowres_init(enc_np)
            _, r4 = r8.lowres_init(ref_np)
            t0 = time.perf_counter()
            cpu_field = r8.lookahead_me(e4, r4, g_lr.cu_w, g_lr.cu_h, 16, J.HEX, 1, 12)
            cpu_dt = time.perf_counter() - t0
            line["lookahead"]["cpu_baseline"] = {"value": look[1][1] / cpu_dt, "unit": "evals/s", "cores": 1, "kind": "reference",
                                                 "ms_per_frame_pair": cpu_dt * 1e3, "sample": "one 1080p frame pair"}
            line["lookahead"]["parity_ok"] = bool(np.array_equal(cpu_field, look_field))
        cb = cpu_baseline()
        if cb is not None:
            # the stated baseline = the fastest of the reference's CPU paths that builds here, on one core;
            # the other rows sit beside it
            best = cb["full"].get("vec_1core", cb["full"]["c_1core"])
            line["cpu_baseline"] = dict(best, workload="FULL +-64 leg")
            line["cpu_baseline_all"] = cb
            line["hex_search"]["cpu_baseline"] = cb["hex"].get("vec_1core", cb["hex"]["c_1core"])
        else:
            o = F.Oracle()
            fs = np.ascontiguousarray(full_jobs[::400])
            t0 = time.perf_counter()
            _, ev = o.search(enc_np, [ref_np], fs, threads=1)
            d = time.perf_counter() - t0
            line["cpu_baseline"] = {"value": ev / d, "unit": "evals/s", "cores": 1, "kind": "port",
                                    "sample": "%d jobs of the FULL +-64 frame (every 400th), %.1f s" % (len(fs), d)}
        # the GPU and CPU legs ran the same jobs: their results must agree (checker, not product)
        o = F.Oracle()
        chk, _ = o.search(enc_np, [ref_np], hex_jobs[::5], threads=8)
        ok = bool(np.array_equal(chk, hex_res[::5]))
        fsel = np.arange(0, len(full_jobs), 400)
        chk, _ = o.search(enc_np, [ref_np], full_jobs[fsel], threads=8)
        ok = ok and bool(np.array_equal(chk, full_res[fsel]))
        hsel = np.arange(0, len(had_jobs), 997)
        chk, _ = o.search(seq4[1], [seq4[0]], had_jobs[hsel], threads=8, bit_depth=10)
        ok = ok and bool(np.array_equal(chk, had_res[hsel]))
        asel = np.arange(len(had_jobs), len(all_jobs), 2503)
        chk, _ = o.search(seq4[1], [seq4[0]], all_jobs[asel], threads=8, bit_depth=10)
        ok = ok and bool(np.array_equal(chk, all_res[asel]))
        ssel = np.arange(0, len(shape_jobs), 1201)
        chk, _ = o.search(seq4[1], [seq4[0]], shape_jobs[ssel], threads=8, bit_depth=10)
        line["sa8d_full_4k10"]["all_shapes"]["parity_sample"] = "%d of %d PUs (every 1201st) equal to the oracle: %s" % (
            len(ssel), len(shape_jobs), bool(np.array_equal(chk, shape_res[ssel])))
        line["parity_sample_ok"] = ok and ok_pipe and bool(np.array_equal(chk, shape_res[ssel]))

    print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
