#!/usr/bin/env python3
"""Timing of the 4K 10-bit SA8D configuration per PU size: FULL +-16 and HEX with the SA8D cost at
integer-pel (subme 0 still runs its half-pel round), whole frame of one square PU size per line."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from x265_amd import jobs as J, synth
from x265_amd.api import Engine
W, H = 3840, 2160
eng = Engine(0)
fr = synth.make_sequence(W, H, 2, seed=5, bit_depth=10)
ids = [eng.upload_plane(f, bit_depth=10) for f in fr]
eng.set_timing(True)
for method, name, mer in ((J.FULL, "FULL+-16", 16), (J.HEX, "HEX", 64)):
    for s in (64, 32, 16, 8):
        x, y = J.square_pu_grid(W - (W % s), H - (H % s), s)
        jobs = J.make_jobs(W, H, x, y, s, s, method=method, subme=0, merange=mer, qp=27, icost=J.SA8D)
        jl = eng.make_joblist(jobs, bytes_per_px=2)
        eng.eval_count(reset=True)
        best = 1e9
        for _ in range(2):
            jl.run(ids[1], [ids[0]], fetch=False)
            a, b = eng.kernel_times()
            best = min(best, a + b)
        ev = eng.eval_count(reset=True) / 2
        jl.close()
        print("%-8s SA8D %2dx%-2d jobs %6d: %8.3f ms  %.3g evals/s  (%.1f evals/job)" % (name, s, s, len(jobs), best, ev / (best * 1e-3), ev / len(jobs)), flush=True)
