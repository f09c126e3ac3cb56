#!/usr/bin/env python3
"""Developer aid: phase clocks of the exhaustive kernel per PU size (library built with -DXA_PROFILE,
X265AMD_LIB pointing at it).  The library prints the clocks on stderr when the counters are read."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from x265_amd import jobs as J, synth
from x265_amd.api import Engine
W, H = 1920, 1080
eng = Engine(0)
fr = synth.make_sequence(W, H, 2, seed=5)
ids = [eng.upload_plane(f) for f in fr]
eng.set_timing(True)
for (w, h) in ((64, 64), (32, 32), (16, 16), (8, 8)):
    x, y = J.square_pu_grid(W - (W % w), H - (H % h), w)
    jobs = J.make_jobs(W, H, x, y, w, h, method=J.FULL, subme=0, merange=64, qp=32)
    jl = eng.make_joblist(jobs)
    for _ in range(2):
        jl.run(ids[1], [ids[0]], fetch=False)
    eng.eval_count(reset=True)
    jl.run(ids[1], [ids[0]], fetch=False)
    print("FULL %dx%d jobs %d k_full %.3f ms" % (w, h, len(jobs), eng.kernel_times()[0]), flush=True)
    sys.stderr.flush()
    eng.eval_count(reset=True)
    jl.close()
