#!/usr/bin/env python3
"""Run one planned FULL-search job list of a single PU size a few times (profiling target)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from x265_amd import jobs as J, synth
from x265_amd.api import Engine
W, H = 1920, 1080
size = int(sys.argv[1]) if len(sys.argv) > 1 else 64
method = int(sys.argv[2]) if len(sys.argv) > 2 else J.FULL
subme = int(sys.argv[3]) if len(sys.argv) > 3 else 0
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
dev = torch.device("cuda", 0)
eng = Engine(0)
base = synth.base_frame(W, H)
ref = synth.pad_plane(base); enc = synth.pad_plane(synth.moved_frame(base, 1))
et = torch.from_numpy(enc).to(dev); rt = torch.from_numpy(ref).to(dev)
eid = eng.wrap_device_plane(et.data_ptr(), W, H, ref.shape[1], 1, 8)
rid = eng.wrap_device_plane(rt.data_ptr(), W, H, ref.shape[1], 1, 8)
sizes = (64, 32, 16, 8) if size == 0 else (size,)
jobs = J.frame_jobs(W, H, sizes=sizes, method=method, subme=subme, merange=64, qp=32)
jl = eng.make_joblist(jobs)
eng.set_timing(True)
for _ in range(reps):
    jl.run(eid, [rid], fetch=False)
    print(eng.kernel_times())
