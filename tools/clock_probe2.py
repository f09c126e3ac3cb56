#!/usr/bin/env python3
"""Developer aid: keep the exhaustive 64x64 search running for a few seconds (for clock sampling with rocm-smi)."""
import os, sys, time
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
size = int(sys.argv[1]) if len(sys.argv) > 1 else 64
x, y = J.square_pu_grid(W - (W % size), H - (H % size), size)
jobs = J.make_jobs(W, H, x, y, size, size, method=J.FULL, subme=0, merange=64, qp=32)
jl = eng.make_joblist(jobs)
t0 = time.time(); ts = []
while time.time() - t0 < 4.0:
    jl.run(eid, [rid], fetch=False)
    ts.append(eng.kernel_times()[0])
print("runs", len(ts), "k_full first %.3f  median %.3f  min %.3f  last %.3f ms" % (ts[0], sorted(ts)[len(ts)//2], min(ts), ts[-1]))
