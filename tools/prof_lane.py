#!/usr/bin/env python3
"""Kernel time of the small-PU searches (1080p frame of one PU shape): run once as is and once with
X265AMD_FLAGS=128 (the smallest PUs stay in the general kernel) to compare the lane-per-job kernel with it."""
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

def run(jobs, reps=3):
    jl = eng.make_joblist(jobs)
    best = [1e9, 1e9]
    for _ in range(reps):
        jl.run(eid, [rid], fetch=False)
        a, b = eng.kernel_times()
        best = [min(best[0], a), min(best[1], b)]
    jl.close()
    return best

print("X265AMD_FLAGS =", os.environ.get("X265AMD_FLAGS", "0"))
shapes = ((8, 8), (8, 4), (4, 8), (16, 4), (16, 8), (16, 16))
if len(sys.argv) > 1 and sys.argv[1] == 'big':
    shapes = ((8, 8), (16, 8), (16, 16), (32, 16), (32, 32))
for (w, h) in shapes:
    x, y = J.square_pu_grid(W - (W % 16), H - (H % 16), max(w, h))
    kinds = (("FULL+-16 subme0", dict(method=J.FULL, subme=0, merange=16)), ("FULL+-16 subme5", dict(method=J.FULL, subme=5, merange=16))) if len(sys.argv) > 1 else None
    for (name, kw) in kinds or (("FULL+-16 subme0", dict(method=J.FULL, subme=0, merange=16)), ("HEX subme0", dict(method=J.HEX, subme=0, merange=64)),
                       ("HEX subme5", dict(method=J.HEX, subme=5, merange=64)), ("STAR subme2", dict(method=J.STAR, subme=2, merange=32))):
        jobs = J.make_jobs(W, H, x, y, w, h, qp=32, **kw)
        eng.eval_count(reset=True)
        a, b = run(jobs, reps=2)
        ev = eng.eval_count() / 2
        print("%2dx%-2d %-16s jobs %6d  exhaustive %7.3f ms  search kernel %7.3f ms  evals/job %.1f" % (w, h, name, len(jobs), a, b, ev / len(jobs)))
