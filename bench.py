#!/usr/bin/env python3
"""Motion-estimation benchmark (one JSON line on rank 0).

    python bench.py --gpus N --steps K --warmup W
    (N > 1: launched by torch.distributed.run, one rank per GPU, RCCL)

A step = one whole-frame ME pass of the hot path on every rank:
  * rank 0 owns the reconstructed reference plane; with N > 1 it is broadcast to the other
    ranks over RCCL (the path's only exchange step), then
  * every rank searches ITS OWN current 1080p frame (frame rank+1 of the synthetic sequence)
    against that reference -- frames are independent, so the work is sharded with no other
    collective (weak scaling: one frame per GPU per step).

Workload of `value` (BASELINE.json configs[1]): 1920x1080 8-bit, HEX search, merange 64, +-64
window rule of the encoder, subme 5 (8 half-pel + 8 quarter-pel SATD), PUs 16x16 and 64x64.
The same run also measures configs[0] on the GPU (FULL +-64 SAD, PUs 64/32/16/8) and reports it
under "full_search", with the SAD-ALU roofline the exhaustive kernel is bound by.

Planes, job lists and result buffers are resident in HBM before the timed region starts.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
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


def cpu_baseline(enc, ref, hex_jobs, full_jobs, budget_s=20.0):
    """Time the CPU path on the host cores for a bounded sample of the same workloads.
    kind 'reference' = the reference's own C path (oracle/_ref, compiled from the reference's
    sources), 1 thread; falls back to kind 'port' (oracle/libme_oracle.so) if it is absent."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_ffi as F
    oracle = F.Oracle()
    out = {}

    def time_ref(jobs):
        lib = F.Reference(8)
        mv = np.zeros((len(jobs), 2), dtype=np.int16)
        cost = np.zeros(len(jobs), dtype=np.int32)
        jb = np.ascontiguousarray(jobs)
        lib.lib.ref_me_search_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_ssize_t, C.c_void_p, C.c_ssize_t,
                                                C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        t0 = time.perf_counter()
        lib.lib.ref_me_search_batch(lib.me, F.plane_origin(enc), enc.shape[1], F.plane_origin(ref), ref.shape[1],
                                    jb.ctypes.data, len(jb), mv.ctypes.data, cost.ctypes.data)
        dt = time.perf_counter() - t0
        res = np.zeros(len(jobs), dtype=J.RESULT_DTYPE)
        res["mvx"], res["mvy"], res["cost"] = mv[:, 0], mv[:, 1], cost
        return dt, res

    def time_port(jobs, threads):
        t0 = time.perf_counter()
        res, evals = oracle.search(enc, [ref], jobs, threads=threads)
        return time.perf_counter() - t0, res, evals

    kind = "reference" if F.have_reference(8) else "port"
    # HEX workload: the whole frame's job list is ~0.2 s of CPU work; repeat a fixed sample
    hs = hex_jobs[::2]
    _, _, hex_evals = time_port(hs[:64], 1)
    reps, dt = 0, 0.0
    while dt < budget_s * 0.5 and reps < 200:
        d, res = time_ref(hs) if kind == "reference" else time_port(hs, 1)[:2]
        dt += d
        reps += 1
    _, chk, hex_evals = time_port(hs, 8)
    assert np.array_equal(chk, res), "CPU baseline legs disagree"
    out["hex"] = {"value": hex_evals * reps / dt, "unit": "evals/s", "cores": 1, "kind": kind,
                  "sample": "%d of %d jobs of the HEX+subme5 frame (every 2nd), %d repetitions, %.1f s" % (len(hs), len(hex_jobs), reps, dt)}
    # FULL workload: ~0.15 s per 64x64 PU on one core -> a strided sample of PUs of all four sizes
    fs = np.concatenate([full_jobs[full_jobs["w"] == s][:: st] for s, st in ((64, 3), (32, 6), (16, 12), (8, 24))])
    d, res = time_ref(fs) if kind == "reference" else time_port(fs, 1)[:2]
    fe = J.full_search_evals(fs)
    px = float((J.full_search_evals(fs[fs["w"] == 64]) * 4096 + J.full_search_evals(fs[fs["w"] == 32]) * 1024 +
                J.full_search_evals(fs[fs["w"] == 16]) * 256 + J.full_search_evals(fs[fs["w"] == 8]) * 64))
    out["full"] = {"value": fe / d, "unit": "evals/s", "pixel_sads_per_s": px / d, "cores": 1, "kind": kind,
                   "sample": "%d PUs (strided over 64/32/16/8) of the FULL+-64 frame, %.1f s" % (len(fs), d)}
    # all host cores given to this job, on the port (one search state per thread)
    cores = min(16, os.cpu_count() or 1)
    d, _, ev = time_port(hs, cores)
    out["hex_mt"] = {"value": ev / d, "unit": "evals/s", "cores": cores, "kind": "port"}
    return out


def trace(msg):
    if os.environ.get("X265AMD_BENCH_TRACE"):
        print("[bench rank %s] %s" % (os.environ.get("RANK", "0"), msg), file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--full-steps", type=int, default=10)
    ap.add_argument("--only-primary", action="store_true",
                    help="run only the HEX+subme5 leg `value` is quoted on (for profiler runs: every launch of the "
                         "search kernels then belongs to that leg) and print a reduced line")
    args = ap.parse_args()

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
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    if os.environ.get("X265AMD_BENCH_TRACE"):       # a stuck rank shows where it is
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ.get("X265AMD_BENCH_TRACE_AFTER", "90")), exit=False)
    trace("process group up")
    from x265_amd.api import Engine
    eng = Engine(local_rank)
    trace("engine created")

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
    eng.set_timing(True)
    trace("job lists planned")
    from x265_amd import dist as xd

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

    ring = xd.PlaneRing(ref_bufs, engine=eng, src=0, overlap=not os.environ.get("X265AMD_NO_OVERLAP"))
    step_no = [0]

    def step(jl):
        # reference plane: rank 0 -> all (RCCL over xGMI when N > 1), then this rank's frame
        k = step_no[0]
        step_no[0] += 1
        b = ring.acquire(k)
        trace("step %d: reference in buffer %d" % (k, b))
        jl.run(enc_id, [ref_ids[b]], fetch=False)
        ring.prefetch(k + 1)
        ring.release(k)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(jl, warmup, steps):
        """K steps queued back to back (no host wait inside the timed region); kernel times of the last step."""
        for _ in range(warmup):
            step(jl)
        trace("warm-up queued")
        eng.sync()
        eng.eval_count(reset=True)
        barrier()
        trace("timed region starts")
        # HIP events bracket the kernels of the LAST step only: timestamp events drain the queue between
        # launches, and with them in every step the back-to-back frames run about 7 % slower
        eng.set_timing(False)
        t0 = time.perf_counter()
        for k in range(steps):
            if k == steps - 1:
                eng.set_timing(True)
            step(jl)
        eng.sync()
        barrier()
        dt = time.perf_counter() - t0
        trace("timed region done")
        k_full, k_search = eng.kernel_times()     # HIP events on the engine's streams around the last step's kernels
        evals = eng.eval_count()
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
            e = torch.tensor([evals], dtype=torch.float64, device=dev)
            dist.all_reduce(e, op=dist.ReduceOp.SUM)
            evals = float(e.item())
        return dt, evals, k_full, k_search

    # ---- primary: HEX + subme 5 ------------------------------------------------------------
    dt, evals, _, k_search_ms = timed(jl_hex, args.warmup, args.steps)
    hex_res = jl_hex.run(enc_id, [ref_id], fetch=True)
    value = evals / dt
    ms_per_step = dt / args.steps * 1e3
    # algorithmic HBM bytes of one launch: enc plane + ref plane read once, jobs read, results written
    alg_bytes = 2 * ref_np.nbytes + hex_jobs.nbytes + len(hex_jobs) * 8
    achieved_gbs = alg_bytes / (k_search_ms * 1e-3) / 1e9

    # ---- the same kernel against the integer-ALU roof: algorithmic operations of one frame ------------
    # (1 SAD evaluation of a w x h block = w*h operations, 1 SATD evaluation = 5*w*h; the split per PU size
    # comes from the evaluation counter of one run per size)
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

    # ---- the exchange step alone (N > 1): reference plane from rank 0 to every rank ----------------
    bcast_us = None
    if world > 1:
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            xd.broadcast_plane(ref_bufs[0], 0, engine=eng)
        barrier()
        t = torch.tensor([(time.perf_counter() - t0) / args.steps * 1e6], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        bcast_us = float(t.item())

    # With N > 1 only the leg `value` is quoted on runs (plus the broadcast alone): the secondary legs are
    # single-GPU figures, and every extra step on N ranks is one more collective per rank for nothing.
    if args.only_primary or world > 1:
        if rank == 0:
            traffic = None
            pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
            if os.path.exists(pmc):
                try:
                    traffic = json.load(open(pmc)).get("k_search_u8_bytes_per_launch")
                except Exception:
                    traffic = None
            print(json.dumps({
                "metric": "ME block-cost evals/sec", "value": value, "unit": "evals/s", "n_gpus": world,
                "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
                "frames_per_s": world * args.steps / dt, "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "dtype": "u8", "data": "synthetic",
                "config": {"workload": "1080p 8-bit, HEX search merange 64 + subme 5 (8 hpel + 8 qpel SATD), PUs 16x16+64x64 "
                                       "(BASELINE configs[1]); one frame per GPU per step; reference plane broadcast from rank 0",
                           "jobs_per_frame": int(len(hex_jobs)), "evals_per_frame": evals / (args.steps * world),
                           "qp": QP, "parallelism": "frames sharded over %d GPU(s)" % world,
                           "reference_broadcast_us": bcast_us, "reference_plane_bytes": int(ref_np.nbytes)},
                "roofline": {"bound": "hbm", "kernel": "k_search<uint8_t>", "achieved": achieved_gbs,
                             "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved_gbs / HBM_PEAK_GBS, "traffic": traffic,
                             "kernel_ms": k_search_ms, "algorithmic_bytes_per_launch": alg_bytes,
                             "kernel_ms_is": "HIP-event bracket around the search kernels of the last step of the timed region"},
                "cpu_baseline": None,
                "note": "primary leg only (--only-primary, or N > 1: cpu_baseline and the secondary legs are reported at N = 1)"}),
                flush=True)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    # ---- informational: launch order by job length (x265amd_joblist_rebalance).  The bench repeats one
    # frame, so the lengths learnt from a run predict the next run exactly - an upper bound for the
    # technique, not part of `value` (the synthetic sequence draws each frame's local motion afresh, and
    # an order learnt on frame t does not help frame t + 1 there).
    jl_reb = eng.make_joblist(hex_jobs)
    jl_reb.run(enc_id, [ref_id], fetch=False)
    jl_reb.rebalance()
    rdt, revals, _, k_reb_ms = timed(jl_reb, 2, args.steps)
    jl_reb.close()

    # ---- the same frame with the diamond search (the other method BASELINE configs[1] names) ----
    dia_jobs = hex_jobs.copy()
    dia_jobs["method"] = J.DIA
    jl_dia = eng.make_joblist(dia_jobs)
    ddt, devals, _, k_dia_ms = timed(jl_dia, 2, args.steps)
    jl_dia.close()

    # ---- PCIe-inclusive: both planes uploaded and the MV/cost field downloaded every frame -------
    enc_pin = torch.from_numpy(enc_np).pin_memory()
    ref_pin = torch.from_numpy(ref_np).pin_memory()
    def pcie_step():
        enc_t.copy_(enc_pin, non_blocking=True)
        if rank == 0:
            ref_t.copy_(ref_pin, non_blocking=True)
        eng.torch_stream().wait_stream(torch.cuda.current_stream(dev))
        return jl_hex.run(enc_id, [ref_id], fetch=True)        # blocks until the results are on the host
    for _ in range(2):
        pcie_step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        pcie_step()
    barrier()
    pcie_dt = time.perf_counter() - t0

    # ---- secondary: FULL +-64 (configs[0] on the GPU) --------------------------------------
    fdt, fevals, k_full_ms, k_tail_ms = timed(jl_full, 2, args.full_steps)
    full_res = jl_full.run(enc_id, [ref_id], fetch=True)
    prop = torch.cuda.get_device_properties(local_rank)
    cus = prop.multi_processor_count
    clk_hz = prop.clock_rate * 1e3 if hasattr(prop, "clock_rate") else 2.4e9
    px_per_frame = float(sum(J.full_search_evals(full_jobs[full_jobs["w"] == s]) * s * s for s in (64, 32, 16, 8)))
    sad_peak = cus * SAD_LANES_PER_CLK_PER_CU * 4 * clk_hz
    full_px_rate = px_per_frame / (k_full_ms * 1e-3)

    if rank != 0:
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    # ---- lookahead path (rank 0 only, not part of `value`): half-resolution plane sets and whole-frame
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

    traffic = None
    pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(pmc):
        try:
            traffic = json.load(open(pmc)).get("k_search_u8_bytes_per_launch")
        except Exception:
            traffic = None

    line = {
        "metric": "ME block-cost evals/sec",
        "value": value,
        "unit": "evals/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "frames_per_s": world * args.steps / dt,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u8",
        "data": "synthetic",
        "config": {"workload": "1080p 8-bit, HEX search merange 64 + subme 5 (8 hpel + 8 qpel SATD), PUs 16x16+64x64 "
                               "(BASELINE configs[1]); one frame per GPU per step; reference plane broadcast from rank 0",
                   "jobs_per_frame": int(len(hex_jobs)), "evals_per_frame": evals / (args.steps * world),
                   # subme 5 = 1 + 8 + 8 SATD evaluations per PU (encoder/motion.cpp:57); the rest are SADs
                   "satd_evals_per_frame": 17 * int(len(hex_jobs)),
                   "sad_evals_per_frame": evals / (args.steps * world) - 17 * int(len(hex_jobs)),
                   "qp": QP, "parallelism": "frames sharded over %d GPU(s)" % world,
                   # the collective alone, host-bracketed (it is also inside every timed step)
                   "reference_broadcast_us": bcast_us, "reference_plane_bytes": int(ref_np.nbytes)},
        "roofline": {"bound": "hbm", "kernel": "k_search<uint8_t>", "achieved": achieved_gbs, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": achieved_gbs / HBM_PEAK_GBS, "traffic": traffic,
                     "kernel_ms": k_search_ms, "algorithmic_bytes_per_launch": alg_bytes,
                     "kernel_ms_is": "HIP-event bracket around the search kernels of the last step of the timed region"},
        # the bound that actually binds this kernel is instruction issue, not memory: the same launch against
        # the un-packed 32-bit integer-ALU roof (CUs x 64 lanes x clock)
        "roofline_valu": {"bound": "valu", "kernel": "k_search<uint8_t>", "achieved": alg_ops / (k_search_ms * 1e-3) / 1e12,
                          "peak": cus * 64 * clk_hz / 1e12, "unit": "T op/s",
                          "frac": alg_ops / (k_search_ms * 1e-3) / (cus * 64 * clk_hz),
                          "algorithmic_ops_per_launch": alg_ops,
                          "ops_def": "SAD evaluation = w*h, SATD evaluation = 5*w*h integer operations"},
        "rebalanced_same_frame": {"workload": "the primary leg again after x265amd_joblist_rebalance (longest jobs of the previous "
                                              "run start first); the bench repeats one frame, so this is the technique's upper bound",
                                  "value": revals / rdt, "unit": "evals/s", "ms_per_step": rdt / args.steps * 1e3, "kernel_ms": k_reb_ms},
        "dia_search": {"workload": "same frame and PUs, DIA search + subme 5", "value": devals / ddt, "unit": "evals/s",
                       "ms_per_frame": ddt / args.steps * 1e3, "kernel_ms": k_dia_ms},
        "pcie_inclusive": {"workload": "HEX+subme5 frame; enc and ref planes copied host->device and the MV/cost field "
                                       "device->host every frame, frames not overlapped",
                           "frames_per_s": world * args.steps / pcie_dt, "ms_per_frame": pcie_dt / args.steps * 1e3,
                           "value": evals / args.steps / world * (world * args.steps / pcie_dt), "unit": "evals/s"},
        "full_search": {"workload": "1080p 8-bit, FULL +-64 SAD, PUs 64/32/16/8, subme 0 (BASELINE configs[0] on the GPU)",
                        "value": fevals / fdt, "unit": "evals/s", "frames_per_s": world * args.full_steps / fdt,
                        "ms_per_frame": fdt / args.full_steps * 1e3, "jobs_per_frame": int(len(full_jobs)),
                        "roofline": {"bound": "valu", "kernel": "k_full_u8", "achieved": full_px_rate / 1e12, "peak": sad_peak / 1e12,
                                     "unit": "T pixel-SAD/s", "frac": full_px_rate / sad_peak, "kernel_ms": k_full_ms,
                                     "search_kernel_ms": k_tail_ms,
                                     "peak_def": "%d CUs x %d lanes/clk x 4 px (v_sad_u8) x %.2f GHz" % (cus, SAD_LANES_PER_CLK_PER_CU, clk_hz / 1e9)}},
        "lookahead": {"workload": "1080p 8-bit -> 960x544 half-resolution sets; whole-frame searches of %d x %d 8x8 CUs with the "
                                  "neighbour-predicted MV chain, HEX + subme 1, merange 16, qp 12 (the reference's lookahead settings)"
                                  % (look_cw, look_ch),
                      "pairs_per_call": LOOK_PAIRS, "kernel_ms_per_call": look[LOOK_PAIRS][0],
                      "frame_pairs_per_s": LOOK_PAIRS / (look[LOOK_PAIRS][0] * 1e-3),
                      "value": look[LOOK_PAIRS][1] / (look[LOOK_PAIRS][0] * 1e-3), "unit": "evals/s",
                      "single_pair_kernel_ms": look[1][0], "lowres_init_us_per_picture": lowres_us},
    }

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
        cb = cpu_baseline(enc_np, ref_np, hex_jobs, full_jobs)
        line["cpu_baseline"] = cb["hex"]
        line["cpu_baseline_full_search"] = cb["full"]
        line["cpu_baseline_all_cores"] = cb["hex_mt"]
        # the GPU and CPU legs ran the same jobs: their results must agree (checker, not product)
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_ffi as F
        o = F.Oracle()
        chk, _ = o.search(enc_np, [ref_np], hex_jobs[::5], threads=8)
        line["parity_sample_ok"] = bool(np.array_equal(chk, hex_res[::5]))
        fsel = np.arange(0, len(full_jobs), 400)
        chk, _ = o.search(enc_np, [ref_np], full_jobs[fsel], threads=8)
        line["parity_sample_ok"] = line["parity_sample_ok"] and bool(np.array_equal(chk, full_res[fsel]))

    print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
