#!/usr/bin/env python3
"""Developer aid: cost of the per-batch timing events on back-to-back HEX + subme 5 frames."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from x265_amd import jobs as J, synth
from x265_amd.api import Engine
W, H = 1920, 1080
dev = torch.device("cuda", 0)
base = synth.base_frame(W, H)
ref = synth.pad_plane(base); enc = synth.pad_plane(synth.moved_frame(base, 1))
et = torch.from_numpy(enc).to(dev); rt = torch.from_numpy(ref).to(dev)
jobs = J.frame_jobs(W, H, sizes=(64, 16), method=J.HEX, subme=5, merange=64, qp=32)
e = Engine(0)
eid = e.wrap_device_plane(et.data_ptr(), W, H, ref.shape[1], 1, 8)
rid = e.wrap_device_plane(rt.data_ptr(), W, H, ref.shape[1], 1, 8)
jl = e.make_joblist(jobs)
for timing in (False, True, False, True):
    e.set_timing(timing)
    for k in range(8):
        jl.run(eid, [rid], fetch=False)
    e.sync()
    K = 100
    t0 = time.perf_counter()
    for k in range(K):
        jl.run(eid, [rid], fetch=False)
    t1 = time.perf_counter()
    e.sync()
    dt = time.perf_counter() - t0
    print("timing events %s: %.4f ms per frame (host submit %.4f ms per frame)" % (timing, dt / K * 1e3, (t1 - t0) / K * 1e3), flush=True)
