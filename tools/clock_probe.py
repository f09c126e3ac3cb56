#!/usr/bin/env python3
"""Developer aid: does the exhaustive-search time depend on how busy the GPU was just before?"""
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
x, y = J.square_pu_grid(W - (W % 64), H - (H % 64), 64)
jobs = J.make_jobs(W, H, x, y, 64, 64, method=J.FULL, subme=0, merange=64, qp=32)
jl = eng.make_joblist(jobs)
a = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
def busy(ms):
    t0 = time.time()
    while (time.time() - t0) * 1e3 < ms:
        (a @ a).sum().item()
for label, pre in (("cold (after 0.5 s idle)", lambda: time.sleep(0.5)), ("after 300 ms of GEMMs", lambda: busy(300)),
                   ("back to back x20", None)):
    if pre:
        pre()
        jl.run(eid, [rid], fetch=False); t = eng.kernel_times()[0]
        print("%-26s k_full %.3f ms" % (label, t))
    else:
        ts = []
        for _ in range(20):
            jl.run(eid, [rid], fetch=False, sync=False) if False else jl.run(eid, [rid], fetch=False)
            ts.append(eng.kernel_times()[0])
        print("%-26s k_full first %.3f  min %.3f  last %.3f ms" % (label, ts[0], min(ts), ts[-1]))
