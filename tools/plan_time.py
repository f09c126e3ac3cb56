import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from x265_amd import jobs as J, synth
from x265_amd.api import Engine
eng = Engine(0)
W, H = 3840, 2160
for sizes in ((64, 32, 16, 8),):
    jobs = J.frame_jobs(W, H, sizes=sizes, method=J.FULL, subme=0, merange=16, qp=27, icost=J.SA8D)
    t = time.perf_counter(); jl = eng.make_joblist(jobs, bytes_per_px=2); dt = time.perf_counter() - t
    print("plan %d SA8D jobs: %.1f ms" % (len(jobs), dt * 1e3)); jl.close()
    jobs = J.frame_jobs(1920, 1080, sizes=sizes, method=J.FULL, subme=0, merange=64, qp=32)
    t = time.perf_counter(); jl = eng.make_joblist(jobs); dt = time.perf_counter() - t
    print("plan %d 1080p FULL jobs: %.1f ms" % (len(jobs), dt * 1e3)); jl.close()
    jobs = J.frame_jobs(1920, 1080, sizes=(16, 64), method=J.HEX, subme=5, merange=64, qp=32)
    t = time.perf_counter(); jl = eng.make_joblist(jobs); dt = time.perf_counter() - t
    print("plan %d 1080p HEX jobs: %.1f ms" % (len(jobs), dt * 1e3)); jl.close()
