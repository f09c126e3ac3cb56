"""1080p FULL +-64 frame (four square PU levels) through the raw device-array entry point and through a planned list:
time per call; run under rocprofv3 --kernel-trace --stats to see the kernels of either path."""
import sys, time
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch
from x265_amd.api import Engine
from x265_amd import jobs as J, synth
eng = Engine(0)
W, H = 1920, 1080
fr = synth.make_sequence(W, H, 2, seed=77)
ids = [eng.upload_plane(f) for f in fr]
jobs = J.frame_jobs(W, H, sizes=(64, 32, 16, 8), method=J.FULL, subme=0, merange=64, qp=32)
dev = torch.device("cuda", 0)
dj = torch.from_numpy(jobs.view(np.uint8).reshape(-1)).to(dev)
do = torch.zeros(len(jobs) * 8, dtype=torch.uint8, device=dev)
which = sys.argv[1] if len(sys.argv) > 1 else "both"
if which in ("both", "list"):
    jl = eng.make_joblist(jobs)
    t = []
    for _ in range(8):
        t0 = time.perf_counter(); jl.run(ids[0], [ids[1]], fetch=False); eng.sync(); t.append(time.perf_counter() - t0)
    print("planned list   %.3f ms" % (1e3 * min(t)))
if which in ("both", "dev"):
    t = []
    for _ in range(8):
        t0 = time.perf_counter(); eng.search_device(ids[0], [ids[1]], dj.data_ptr(), len(jobs), do.data_ptr()); eng.sync(); t.append(time.perf_counter() - t0)
    print("device array   %.3f ms" % (1e3 * min(t)))
