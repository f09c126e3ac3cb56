#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY (the cpu_baseline leg of bench.py): times the reference's own CPU path
(oracle/_ref, compiled from the reference's sources) on a bounded sample of the benchmark's job lists,
on the host cores of the machine it runs on.  One process = one core; bench.py starts several side by
side for the all-cores figure (one MotionEstimate object per process mirrors the reference's one per
CTU-row thread, encoder/cturow.h:56).

    cpu_baseline_worker.py <kind: 8 | 8v> <leg: full | hex> <seconds> [evaluations of one pass over the sample]
prints one JSON line {"evals": E, "seconds": S, "jobs": N, "reps": R, "pixel_sads": P, "crc": C}.
"""
import ctypes as C
import json
import os
import sys
import time
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle_ffi as F  # noqa: E402
from x265_amd import jobs as J  # noqa: E402
from x265_amd import synth  # noqa: E402

W, H, QP, MERANGE = 1920, 1080, 32, 64


def sample_jobs(leg):
    """the bounded samples: the SAME job lists as the GPU legs, strided"""
    if leg == "full":
        full = J.frame_jobs(W, H, sizes=(64, 32, 16, 8), method=J.FULL, subme=0, merange=MERANGE, qp=QP)
        return np.concatenate([full[full["w"] == s][::st] for s, st in ((64, 24), (32, 48), (16, 96), (8, 192))])
    hexj = J.frame_jobs(W, H, sizes=(64, 16), method=J.HEX, subme=5, merange=MERANGE, qp=QP)
    return np.ascontiguousarray(hexj[::2])


def main():
    kind = sys.argv[1]
    leg = sys.argv[2]
    budget = float(sys.argv[3])
    lib = F.Reference("8v" if kind == "8v" else 8)
    base = synth.base_frame(W, H)
    ref = synth.pad_plane(base)
    enc = synth.pad_plane(synth.moved_frame(base, 1))
    jobs = sample_jobs(leg)
    # evaluation count of the sample: from the restatement (the reference does not count), given by the caller
    # when it already knows it
    evals = float(sys.argv[4]) if len(sys.argv) > 4 else F.Oracle().search(enc, [ref], jobs, threads=8)[1]
    mv = np.zeros((len(jobs), 2), dtype=np.int16)
    cost = np.zeros(len(jobs), dtype=np.int32)
    fn = lib.lib.ref_me_search_batch
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_ssize_t, C.c_void_p, C.c_ssize_t, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    reps, dt = 0, 0.0
    while dt < budget and reps < 1000:
        t0 = time.perf_counter()
        fn(lib.me, F.plane_origin(enc), enc.shape[1], F.plane_origin(ref), ref.shape[1], jobs.ctypes.data, len(jobs),
           mv.ctypes.data, cost.ctypes.data)
        dt += time.perf_counter() - t0
        reps += 1
    px = float(sum(J.full_search_evals(jobs[jobs["w"] == s]) * s * s for s in (64, 32, 16, 8))) if leg == "full" else 0.0
    print(json.dumps({"evals": float(evals) * reps, "seconds": dt, "jobs": int(len(jobs)), "reps": reps, "pixel_sads": px * reps,
                      "crc": zlib.crc32(mv.tobytes() + cost.tobytes())}), flush=True)


if __name__ == "__main__":
    main()
