#!/usr/bin/env python3
"""Per-PU-size kernel timing of the exhaustive and general search kernels (1080p frame)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from x265_amd import jobs as J, synth
from x265_amd.api import Engine

W, H = 1920, 1080
dev = torch.device("cuda", 0)
eng = Engine(0)
base = synth.base_frame(W, H)
ref = synth.pad_plane(base); enc = synth.pad_plane(synth.moved_frame(base, 1))
et = torch.from_numpy(enc).to(dev); rt = torch.from_numpy(ref).to(dev)
eid = eng.wrap_device_plane(et.data_ptr(), W, H, ref.shape[1], 1, 8)
rid = eng.wrap_device_plane(rt.data_ptr(), W, H, ref.shape[1], 1, 8)
eng.set_timing(True)
clk = 2.4e9; peak = 256 * 64 * 4 * clk

def run(jobs, reps=3):
    jl = eng.make_joblist(jobs)
    best = [1e9, 1e9]
    for _ in range(reps):
        jl.run(eid, [rid], fetch=False)
        a, b = eng.kernel_times()
        best = [min(best[0], a), min(best[1], b)]
    jl.close()
    return best

which = sys.argv[1] if len(sys.argv) > 1 else "all"
if which in ("all", "full"):
    for (w, h) in ((64, 64), (32, 32), (16, 16), (8, 8), (64, 32), (16, 8), (4, 4)):
        x, y = J.square_pu_grid(W - (W % max(w, h)), H - (H % max(w, h)), max(w, h))
        jobs = J.make_jobs(W, H, x, y, w, h, method=J.FULL, subme=0, merange=64, qp=32)
        a, b = run(jobs)
        px = J.full_search_evals(jobs) * w * h
        print("FULL %2dx%-2d jobs %6d  k_full %8.3f ms  %6.2f T px-SAD/s  frac %.3f   tail k_search %.3f ms" % (w, h, len(jobs), a, px / a / 1e9, px / (a * 1e-3) / peak, b))
if which in ("all", "hex"):
    for method, name in ((J.HEX, "HEX"), (J.DIA, "DIA"), (J.UMH, "UMH"), (J.STAR, "STAR")):
        for s in (64, 32, 16, 8):
            for subme in (0, 5):
                jobs = J.frame_jobs(W, H, sizes=(s,), method=method, subme=subme, merange=64, qp=32)
                eng.eval_count(reset=True)
                a, b = run(jobs, reps=2)
                ev = eng.eval_count() / 2
                print("%-4s %2dx%-2d subme %d jobs %6d  k_search %8.3f ms  %7.1f us/job-wave  evals/job %.1f" % (name, s, s, subme, len(jobs), b, b * 1e3 / len(jobs) * 256 * 8, ev / len(jobs)))
