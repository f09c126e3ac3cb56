#!/usr/bin/env python3
"""Developer aid: queue searches without host waits while torch copies run on other streams
(checks that the engine's fork/join streams and foreign streams do not block each other)."""
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
jobs = J.frame_jobs(W, H, sizes=(64, 16), method=J.HEX, subme=5, merange=64, qp=32)
jl = eng.make_joblist(jobs)
nstreams = int(sys.argv[1]) if len(sys.argv) > 1 else 2
side = [torch.cuda.Stream(dev) for _ in range(nstreams)]
pin = torch.from_numpy(ref).pin_memory()
scratch = [torch.empty_like(rt) for _ in range(nstreams)]
host = [torch.empty(ref.shape, dtype=torch.uint8).pin_memory() for _ in range(nstreams)]
t0 = time.time()
for k in range(300):
    jl.run(eid, [rid], fetch=False)
    for i, s in enumerate(side):
        with torch.cuda.stream(s):
            scratch[i].copy_(pin, non_blocking=True)
            host[i].copy_(scratch[i], non_blocking=True)
    if k % 50 == 0:
        print("queued", k, flush=True)
eng.sync()
torch.cuda.synchronize()
print("done in %.3f s" % (time.time() - t0), flush=True)
