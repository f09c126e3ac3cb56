#!/usr/bin/env python3
"""Developer aid: per-phase shader-clock totals of the search kernel (library built with -DXA_PROFILE)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from x265_amd import jobs as J, synth
from x265_amd.api import Engine
W, H = 1920, 1080
size = int(sys.argv[1]); method = int(sys.argv[2]); subme = int(sys.argv[3])
dev = torch.device("cuda", 0)
eng = Engine(0)
base = synth.base_frame(W, H)
ref = synth.pad_plane(base); enc = synth.pad_plane(synth.moved_frame(base, 1))
et = torch.from_numpy(enc).to(dev); rt = torch.from_numpy(ref).to(dev)
eid = eng.wrap_device_plane(et.data_ptr(), W, H, ref.shape[1], 1, 8)
rid = eng.wrap_device_plane(rt.data_ptr(), W, H, ref.shape[1], 1, 8)
jobs = J.frame_jobs(W, H, sizes=(size,), method=method, subme=subme, merange=64, qp=32)
jl = eng.make_joblist(jobs)
eng.set_timing(True)
jl.run(eid, [rid], fetch=False)
eng.eval_count(reset=True)
jl.run(eid, [rid], fetch=False)
print("jobs", len(jobs), "kernel ms", eng.kernel_times())
sys.stderr.flush()
print("evals", eng.eval_count())
