#!/usr/bin/env python3
"""Per-PU-size timing of the exhaustive kernel on a 1080p 10-bit (16-bit pixel) frame."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from x265_amd import jobs as J, synth
from x265_amd.api import Engine
W, H = 1920, 1080
eng = Engine(0)
fr = synth.make_sequence(W, H, 2, seed=5, bit_depth=10)
ids = [eng.upload_plane(f, bit_depth=10) for f in fr]
eng.set_timing(True)
peak = 256 * 64 * 2 * 2.4e9          # v_sad_u16: 2 pixels per lane-op
for (w, h) in ((64, 64), (32, 32), (16, 16), (8, 8)):
    x, y = J.square_pu_grid(W - (W % w), H - (H % h), w)
    jobs = J.make_jobs(W, H, x, y, w, h, method=J.FULL, subme=0, merange=64, qp=32)
    jl = eng.make_joblist(jobs, bytes_per_px=2)
    best = 1e9
    for _ in range(3):
        jl.run(ids[1], [ids[0]], fetch=False)
        best = min(best, eng.kernel_times()[0])
    jl.close()
    px = J.full_search_evals(jobs) * w * h
    print("FULL u16 %2dx%-2d jobs %6d  k_full %8.3f ms  %6.2f T px-SAD/s  frac %.3f of the v_sad_u16 peak" % (w, h, len(jobs), best, px / best / 1e9, px / (best * 1e-3) / peak))
jobs = J.frame_jobs(W, H, sizes=(64, 32, 16, 8), method=J.FULL, subme=0, merange=64, qp=32)
jl = eng.make_joblist(jobs, bytes_per_px=2)
best = (1e9, 0)
for _ in range(3):
    jl.run(ids[1], [ids[0]], fetch=False)
    best = min(best, eng.kernel_times())
jl.close()
px = sum(J.full_search_evals(jobs[jobs["w"] == s]) * s * s for s in (64, 32, 16, 8))
print("FULL u16 all four levels, jobs %6d  k_full %8.3f ms (tail %.3f)  %6.2f T algorithmic px-SAD/s  frac %.3f" % (len(jobs), best[0], best[1], px / best[0] / 1e9, px / (best[0] * 1e-3) / peak))
